// local_graph.hpp — the graph side of giraffe's "align the bases between two anchors / beyond one anchor" step: a small mutable
// bidirected sequence graph and the graph algorithms MinimizerMapper::with_dagified_local_graph strings together
// (reference: src/minimizer_mapper_from_chains.cpp:3342-3628):
//
//   extract_connecting_graph   the part of a graph on walks of at most max_len bases between two positions, the end nodes cut at
//                              the positions (behaviour of src/algorithms/extract_connecting_graph.cpp)
//   extract_extending_graph    everything within max_dist bases of one position, in one direction, the start node cut at the position
//                              (behaviour of src/algorithms/extract_extending_graph.cpp)
//   StrandSplitView            every strand of every node a forward node of its own (src/split_strand_graph.cpp)
//   dagify_from                an acyclic graph holding every walk of up to a given length that leaves the given handles
//   find_tips                  the handles nothing leads into
//
//   dagify / split_strands / is_acyclic / is_single_stranded / extract_containing_graph   what Mapper::align_to_graph strings together
//                              (src/mapper.cpp:2425-2554, src/cluster.cpp:3832-3851; cluster_alignment.hpp)
//
// dagify, dagify_from, find_tips, split_strands (and find_shortest_paths, which the strict pruning of the connecting graph uses) live in
// libhandlegraph, an empty submodule of the reference snapshot: they are written here from their documented contracts.  What the
// reference's own unit tests hold about them IS checked (tests/test_graph_algorithms.py over tests/golden/ref_graph_algorithms.json:
// src/unittest/dagify.cpp — 6 / 8 / 6 copies for the three small loops, which fixes how far a cycle is unrolled; every listed walk
// preserved; dagify_from's tips — and the 39 extraction cases of src/unittest/vg_algorithms.cpp).  What those leave open (node
// numbering, the order edges are listed in, which edges of a cycle climb to the next copy) is this file's own choice
// [PARITY-UNPINNED]; none of it changes an alignment's score, it can only choose differently among equally good alignments.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "vg_standin/alignment.hpp"
#include "vg_standin/handle_graph.hpp"

namespace vgamd {

// A mutable bidirected graph.  Node ids ascend in iteration (std::map), edges keep insertion order: everything downstream is
// deterministic.  Handles pack (id << 1) | is_reverse like the rest of the shim.
class LocalGraph : public HandleGraph {
public:
    handle_t create_handle(const std::string& seq);                        // id = largest id so far + 1
    handle_t create_handle(const std::string& seq, nid_t id);
    void create_edge(const handle_t& from, const handle_t& to);            // any orientations; an edge that exists is not added again
    bool has_edge(const handle_t& from, const handle_t& to) const;
    void destroy_edge(const handle_t& from, const handle_t& to);
    void destroy_handle(const handle_t& h);                                // the node (either orientation names it) and its edges
    // split the node at `offset` bases into h; both pieces come back in h's orientation, in h's reading order.  The piece that holds
    // the forward strand's first base keeps the id.
    std::pair<handle_t, handle_t> divide_handle(const handle_t& h, size_t offset);
    // drop what lies left (trunc_left) or right of `offset` along h, with the edges of that side; -> the kept piece, in h's orientation
    handle_t truncate_handle(const handle_t& h, bool trunc_left, size_t offset);
    void clear() { nodes_.clear(); }
    size_t get_total_length() const;

    bool has_node(nid_t id) const override { return nodes_.count(id) != 0; }
    size_t get_length(const handle_t& h) const override { return nodes_.at(get_id(h)).seq.size(); }
    std::string get_sequence(const handle_t& h) const override;
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return nodes_.size(); }
    nid_t min_node_id() const override { return nodes_.empty() ? 0 : nodes_.begin()->first; }
    nid_t max_node_id() const override { return nodes_.empty() ? 0 : nodes_.rbegin()->first; }

private:
    struct Node {
        std::string seq;                       // forward strand
        std::vector<handle_t> right;           // x: the edge (node+ -> x) exists
        std::vector<handle_t> left;            // x: the edge (x -> node+) exists
    };
    std::map<nid_t, Node> nodes_;
    void edges_of(nid_t id, std::vector<edge_t>& out) const;               // every edge touching the node, each once, as (from, to)
};

// The graph with each strand of each node as a forward node: strand s of node n is node (n << 1) | s; its reverse orientation reads
// the other strand and is never reached by following edges from a forward handle (src/split_strand_graph.cpp:17-100).
class StrandSplitView : public HandleGraph {
public:
    explicit StrandSplitView(const HandleGraph* g) : g_(g) {}
    bool has_node(nid_t id) const override { return g_->has_node(id >> 1); }
    size_t get_length(const handle_t& h) const override { return g_->get_length(g_->get_handle(get_id(h) >> 1)); }
    std::string get_sequence(const handle_t& h) const override { return g_->get_sequence(get_underlying_handle(h)); }
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return g_->get_node_count() << 1; }
    nid_t min_node_id() const override { return g_->min_node_id() << 1; }
    nid_t max_node_id() const override { return (g_->max_node_id() << 1) | 1; }
    handle_t get_underlying_handle(const handle_t& h) const { return g_->get_handle(get_id(h) >> 1, ((get_id(h) & 1) != 0) != get_is_reverse(h)); }
    handle_t get_overlay_handle(const handle_t& underlying) const { return get_handle((g_->get_id(underlying) << 1) | (g_->get_is_reverse(underlying) ? 1 : 0), false); }
private:
    const HandleGraph* g_;
};

namespace handlealgs {

// every handle, in either orientation, that no edge leads into: a forward one is a head, a reverse one is a tail seen from outside
std::vector<handle_t> find_tips(const HandleGraph* g);

// Dijkstra from `start` (leftwards if asked): handle -> the bases between the far side of `start` and the near side of the handle
// (the start itself: 0).  [PARITY-UNPINNED: libhandlegraph's convention for the start node's own length is taken from its header
// comment, "between the outgoing side of the start and the incoming side of the target".]
std::unordered_map<handle_t, size_t, handle_hash> find_shortest_paths(const HandleGraph* g, const handle_t& start, bool traverse_leftward);

// An acyclic copy of the part of `g` that walks leaving `starts` can reach (a reverse start walks against the edges), in which every
// such walk of up to `min_preserved_path_length` bases is still a walk.  `g` must keep strands apart (a StrandSplitView does): an
// edge between a forward and a reverse handle is an error.  -> (node of `into` -> node of `g`, the starts' handles in `into`, in the
// orientation they were given in).  Nodes of `into` are numbered from 1 in the order a breadth-first search from the starts meets
// them; the extra copies that unroll a cycle follow.
struct Dagified { std::unordered_map<nid_t, nid_t> to_source; std::vector<handle_t> starts; };
Dagified dagify_from(const HandleGraph* g, const std::vector<handle_t>& starts, LocalGraph* into, size_t min_preserved_path_length);
// the whole graph (handlealgs::dagify, as Mapper::align_cluster calls it: src/mapper.cpp:2508-2515): -> node of `into` -> node of g
std::unordered_map<nid_t, nid_t> dagify(const HandleGraph* g, LocalGraph* into, size_t min_preserved_path_length);
bool is_acyclic(const HandleGraph* g);                    // no directed walk returns to the oriented node it left (handlealgs::is_acyclic / is_directed_acyclic)
bool is_single_stranded(const HandleGraph* g);            // no edge joins a forward strand to a reverse one (handlealgs::is_single_stranded)
// every strand a forward node of a new graph (handlealgs::split_strands): -> node of `into` -> (node of g, is_reverse)
std::unordered_map<nid_t, std::pair<nid_t, bool>> split_strands(const HandleGraph* g, LocalGraph* into);

}  // namespace handlealgs

inline bool is_empty(const Position& p) { return p.node_id == 0; }

// -> node of `into` -> node of `source` (empty, and `into` empty, when pos_2 cannot be reached from pos_1 within max_len bases);
// left_id / right_id: the nodes of `into` that hold what is left of pos_1's and pos_2's nodes (the same node when the positions
// face each other on one node).  What the reference recovers by scanning the translation for copies of the anchors' node and
// assuming an id order (src/minimizer_mapper_from_chains.cpp:3403-3486) is reported here instead.
struct ConnectingGraph { std::unordered_map<nid_t, nid_t> to_source; nid_t left_id = 0, right_id = 0; };
ConnectingGraph extract_connecting_graph(const HandleGraph* source, LocalGraph* into, int64_t max_len, const Position& pos_1, const Position& pos_2,
                                         bool strict_max_len);

// -> node of `into` -> node of `source`; cut_id: the node that holds what is left of pos's node
struct ExtendingGraph { std::unordered_map<nid_t, nid_t> to_source; nid_t cut_id = 0; };
ExtendingGraph extract_extending_graph(const HandleGraph* source, LocalGraph* into, int64_t max_dist, const Position& pos, bool backward,
                                       bool preserve_cycles_on_src_node);

// Everything within the given distances of any of the positions, forward and backward of each, whole nodes under their own ids
// (behaviour of src/algorithms/extract_containing_graph.cpp; Mapper::align_cluster's cluster graph: src/cluster.cpp:3832-3851).
// reversing_walk_length > 0 lets a walk turn around onto the other strand and run that much further.
void extract_containing_graph(const HandleGraph* source, LocalGraph* into, const std::vector<Position>& positions,
                              const std::vector<size_t>& forward_search_lengths, const std::vector<size_t>& backward_search_lengths,
                              size_t reversing_walk_length = 0);

}  // namespace vgamd
