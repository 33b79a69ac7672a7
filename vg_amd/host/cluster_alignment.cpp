// cluster_alignment.cpp — see cluster_alignment.hpp
#include "cluster_alignment.hpp"
#include <algorithm>
#include "chain_alignment.hpp"

namespace vgamd {

LocalGraph cluster_subgraph_containing(const HandleGraph& base, const Alignment& aln, const std::vector<ClusterSeed>& cluster, const GSSWAligner& aligner) {
    // far enough from every seed for any hit that is detectable without soft clipping (src/cluster.cpp:3839-3846)
    std::vector<Position> positions; std::vector<size_t> forward, backward;
    const size_t L = aln.sequence.size();
    for (const ClusterSeed& mem : cluster) {
        positions.push_back(mem.start);
        forward.push_back(aligner.scorer->longest_detectable_gap(L, mem.end) + (L - mem.begin));
        backward.push_back(aligner.scorer->longest_detectable_gap(L, mem.begin) + mem.begin);
    }
    LocalGraph cluster_graph;
    extract_containing_graph(&base, &cluster_graph, positions, forward, backward);
    return cluster_graph;
}

AlignableGraph make_alignable(const HandleGraph& graph, size_t target_length, bool reverse_strand) {
    AlignableGraph out;
    out.single_stranded = handlealgs::is_single_stranded(&graph);
    if (out.single_stranded && !reverse_strand) {
        // the forward strand as it is, under the graph's own ids (:2487-2499)
        graph.for_each_handle_v([&](const handle_t& h) {
            out.graph.create_handle(graph.get_sequence(h), graph.get_id(h));
            out.node_trans[graph.get_id(h)] = {graph.get_id(h), false};
        });
        graph.for_each_handle_v([&](const handle_t& h) {
            graph.follow_edges_v(h, false, [&](const handle_t& next) { out.graph.create_edge(out.graph.get_handle(graph.get_id(h), false), out.graph.get_handle(graph.get_id(next), graph.get_is_reverse(next))); });
        });
    } else if (out.single_stranded) {
        // handlealgs::reverse_complement_graph (:2476-2482): every node turned around, under its own id
        graph.for_each_handle_v([&](const handle_t& h) {
            out.graph.create_handle(graph.get_sequence(graph.flip(h)), graph.get_id(h));
            out.node_trans[graph.get_id(h)] = {graph.get_id(h), true};
        });
        graph.for_each_handle_v([&](const handle_t& h) {
            graph.follow_edges_v(h, false, [&](const handle_t& next) {       // h -> next on the forward strands = next' -> h' on the turned ones
                out.graph.create_edge(out.graph.get_handle(graph.get_id(next), graph.get_is_reverse(next)), out.graph.get_handle(graph.get_id(h), false)); });
        });
    } else {
        out.node_trans = handlealgs::split_strands(&graph, &out.graph);      // (:2501-2508)
    }
    if (!handlealgs::is_acyclic(&out.graph)) {                               // (:2510-2516)
        out.was_cyclic = true;
        LocalGraph dagified;
        const std::unordered_map<nid_t, nid_t> dagify_trans = handlealgs::dagify(&out.graph, &dagified, target_length);
        std::unordered_map<nid_t, std::pair<nid_t, bool>> overlaid;          // overlay_node_translations (src/utility.cpp:945-953)
        for (const auto& kv : dagify_trans) overlaid[kv.first] = out.node_trans.at(kv.second);
        out.graph = std::move(dagified);
        out.node_trans = std::move(overlaid);
    }
    return out;
}

void translate_oriented_node_ids(Path& path, const std::unordered_map<nid_t, std::pair<nid_t, bool>>& translator) {
    for (Mapping& m : path.mapping) {
        const std::pair<nid_t, bool>& t = translator.at(m.position.node_id);
        m.position.node_id = t.first;
        m.position.is_reverse = t.second != m.position.is_reverse;
    }
}

namespace {
bool is_softclip(const Edit& e) { return e.from_length == 0 && e.to_length > 0; }
int64_t softclip_start(const Alignment& aln) {
    if (aln.path.mapping.empty() || aln.path.mapping.front().edit.empty()) return 0;
    const Edit& e = aln.path.mapping.front().edit.front();
    return is_softclip(e) ? e.to_length : 0;
}
int64_t softclip_end(const Alignment& aln) {
    if (aln.path.mapping.empty() || aln.path.mapping.back().edit.empty()) return 0;
    const Edit& e = aln.path.mapping.back().edit.back();
    return is_softclip(e) ? e.to_length : 0;
}
}  // namespace

Alignment align_to_graph(const Alignment& aln, const HandleGraph& graph, const Aligner& aligner, bool do_flip, bool traceback,
                         bool pinned_alignment, bool pin_left, bool banded_global, bool keep_bonuses) {
    const size_t L = aln.sequence.size();
    // the longest path an alignment can take: a full read and the longest gap that still pays (:2443)
    const size_t target_length = L + aligner.scorer->longest_detectable_gap(L, L / 2);
    Alignment aligned = aln;
    // One strand suffices when no walk changes strands; the read is then turned instead of the graph when the reverse strand is
    // wanted (no MEMs here: do_flip says so, :2461-2474)
    const bool single = handlealgs::is_single_stranded(&graph);
    bool flipped_alignment = false;
    if (single && do_flip) {
        aligned.sequence = reverse_complement(aligned.sequence);
        std::reverse(aligned.quality.begin(), aligned.quality.end());
        flipped_alignment = true;
    }
    AlignableGraph ag = make_alignable(graph, target_length, false);
    if (banded_global) {
        const size_t band_padding = std::max<size_t>(L, 1);                  // permissive banding around the read's length (:2523-2527)
        aligner.align_global_banded(aligned, ag.graph, (int32_t)band_padding, false);
    } else if (pinned_alignment) aligner.align_pinned(aligned, ag.graph, pin_left);
    else aligner.align(aligned, ag.graph, traceback);
    if (traceback && !keep_bonuses && aligned.score) {                       // remove_full_length_bonuses (:4938-4944)
        if (softclip_start(aligned) == 0) aligned.score -= aligner.scorer->full_length_bonus;
        if (softclip_end(aligned) == 0) aligned.score -= aligner.scorer->full_length_bonus;
    }
    if (flipped_alignment)
        aligned = reverse_complement_alignment(aligned, [&](nid_t id) { return (int64_t)ag.graph.get_length(ag.graph.get_handle(id, false)); });
    if (!ag.node_trans.empty()) translate_oriented_node_ids(aligned.path, ag.node_trans);
    return aligned;
}

}  // namespace vgamd
