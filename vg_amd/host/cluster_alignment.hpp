// cluster_alignment.hpp — `vg map`'s side of SURVEY §8f N1: from a cluster of seeds to a graph the alignment kernels take.
//
//   cluster_subgraph_containing   the part of the graph any alignment of the read through the cluster's seeds can touch
//                                 (src/cluster.cpp:3832-3851 -> algorithms::extract_containing_graph)
//   align_to_graph                Mapper::align_to_graph (src/mapper.cpp:2425-2554): one strand if the graph allows it (the read turned
//                                 around if the seeds sit on the reverse strand), else both strands split apart; cycles unrolled to
//                                 read length + longest detectable gap (handlealgs::dagify); the alignment by the requested method
//                                 through the engine; bonuses removed unless kept; the path translated back to the graph's own nodes.
//
// The X-drop branch of align_to_graph that is seeded with the cluster's MEMs (:2530-2535) is Aligner::align_xdrop on the same prepared
// graph; callers that hold MEMs translate them with the returned node translation (`AlignableGraph::node_trans`) themselves.
#pragma once
#include <unordered_map>
#include <utility>
#include <vector>
#include "aligner.hpp"
#include "local_graph.hpp"

namespace vgamd {

// a seed of a cluster: the read interval it matches and the graph position of its first base
struct ClusterSeed { size_t begin = 0, end = 0; Position start; };

LocalGraph cluster_subgraph_containing(const HandleGraph& base, const Alignment& aln, const std::vector<ClusterSeed>& cluster, const GSSWAligner& aligner);

// The graph the kernels see and what its nodes stand for: node of `graph` -> (node of the caller's graph, on its reverse strand?)
struct AlignableGraph {
    LocalGraph graph;
    std::unordered_map<nid_t, std::pair<nid_t, bool>> node_trans;
    bool single_stranded = false, was_cyclic = false;
};
// src/mapper.cpp:2447-2516 (without the read: `reverse_strand` = the caller wants the graph's reverse strand and cannot turn the read around)
AlignableGraph make_alignable(const HandleGraph& graph, size_t target_length, bool reverse_strand = false);

// position.node_id / is_reverse through a translation (translate_oriented_node_ids, src/path.cpp:2423-2430)
void translate_oriented_node_ids(Path& path, const std::unordered_map<nid_t, std::pair<nid_t, bool>>& translator);

Alignment align_to_graph(const Alignment& aln, const HandleGraph& graph, const Aligner& aligner, bool do_flip, bool traceback,
                         bool pinned_alignment = false, bool pin_left = false, bool banded_global = false, bool keep_bonuses = true);

}  // namespace vgamd
