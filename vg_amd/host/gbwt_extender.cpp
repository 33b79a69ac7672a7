#include "gbwt_extender.hpp"
#include <algorithm>
#include <stdexcept>

namespace vgamd {

HaplotypeGraph::HaplotypeGraph(const HandleGraph& graph, const std::vector<std::vector<handle_t>>& threads) {
    graph.for_each_handle_v([&](const handle_t& h) { ids_.push_back(graph.get_id(h)); });
    std::sort(ids_.begin(), ids_.end());
    for (size_t i = 0; i < ids_.size(); ++i) { index_of_[ids_[i]] = i; seqs_.push_back(graph.get_sequence(graph.get_handle(ids_[i], false))); }
    for (const auto& t : threads) {
        threads_.emplace_back();
        for (const handle_t& h : t) threads_.back().push_back(2u * (uint32_t)index_of_.at(graph.get_id(h)) + (graph.get_is_reverse(h) ? 1u : 0u));
        for (size_t k = 0; k + 1 < t.size(); ++k) { edges_.insert({t[k].v, t[k + 1].v}); edges_.insert({t[k + 1].v ^ 1, t[k].v ^ 1}); }
    }
    // adjacency by oriented node, neighbours in handle order (the order the sorted edge set lists them in)
    next_.assign(2 * ids_.size(), {}); prev_.assign(2 * ids_.size(), {});
    for (const auto& e : edges_) { next_[oriented(handle_t{e.first})].push_back(e.second); prev_[oriented(handle_t{e.second})].push_back(e.first); }
    for (auto& v : prev_) std::sort(v.begin(), v.end());
}
static char complement_base(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}
std::string HaplotypeGraph::get_sequence(const handle_t& h) const {
    const std::string& s = seqs_[index_of_.at(get_id(h))];
    if (!get_is_reverse(h)) return s;
    std::string r(s.rbegin(), s.rend());
    for (char& c : r) c = complement_base(c);
    return r;
}
bool HaplotypeGraph::follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    for (int64_t x : (go_left ? prev_ : next_)[oriented(h)]) if (!it(handle_t{x})) return false;
    return true;
}
bool HaplotypeGraph::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    for (nid_t id : ids_) if (!it(get_handle(id, false))) return false;
    return true;
}

// (GaplessExtension / WFAAlignment methods and the default error model: vg_standin/gbwt_types.cpp)

// ---- GaplessExtender ---------------------------------------------------------------------------------------------
GaplessExtender::GaplessExtender(const HaplotypeGraph& g, const Aligner& a) : graph(&g), aligner(&a) {
    std::vector<uint32_t> node_len, thread_off{0}, thread_nodes; std::string seq;
    for (const std::string& s : g.sequences()) { node_len.push_back((uint32_t)s.size()); seq += s; }
    for (const auto& t : g.threads()) { thread_nodes.insert(thread_nodes.end(), t.begin(), t.end()); thread_off.push_back((uint32_t)thread_nodes.size()); }
    vgk_haplotypes d{};
    d.n_nodes = (uint32_t)node_len.size(); d.node_len = node_len.data(); d.seq = seq.data();
    d.n_threads = (uint32_t)g.threads().size(); d.thread_off = thread_off.data(); d.thread_nodes = thread_nodes.empty() ? thread_off.data() : thread_nodes.data();
    const int rc = a.engine_api().haplo_create(a.engine_context(), &d, &index);
    if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: cannot index the haplotypes: ") + a.engine_api().strerror(rc));
}
GaplessExtender::~GaplessExtender() { if (index) aligner->engine_api().haplo_destroy(index); }

std::vector<GaplessExtension> GaplessExtender::extend(const cluster_type& cluster, std::string sequence, size_t max_mismatches,
                                                      double overlap_threshold, bool trim) const {
    std::vector<GaplessExtension> result;
    if (cluster.empty() || sequence.empty()) return result;          // (:535-537)
    std::vector<vgk_seed> seeds;
    for (const seed_type& s : cluster) {
        vgk_seed v; v.node = graph->oriented(s.first); v.diff = (int32_t)s.second;
        bool dup = false; for (const vgk_seed& o : seeds) dup |= (o.node == v.node && o.diff == v.diff);
        if (!dup) seeds.push_back(v);
    }
    vgk_gapless_problem p{};
    p.read = sequence.c_str(); p.read_len = (uint32_t)sequence.size(); p.seeds = seeds.data(); p.n_seeds = (uint32_t)seeds.size();
    p.max_mismatches = (uint32_t)max_mismatches; p.flags = trim ? VGK_GAPLESS_TRIM : 0u; p.overlap_threshold = overlap_threshold;
    vgk_gapless_result res{};
    std::vector<vgk_extension> ext(seeds.size() + 1);
    std::vector<uint32_t> nodes(seeds.size() * (sequence.size() + 2) + 1), mism(seeds.size() * sequence.size() + 1);
    size_t written[3];
    const int rc = aligner->engine_api().gapless_extend(aligner->engine_context(), index, &p, 1, &res, ext.data(), ext.size(), nodes.data(), nodes.size(),
                                                        mism.data(), mism.size(), written);
    if (rc != VGK_OK || res.status != VGK_OK) throw std::runtime_error(std::string("vgamd: gapless extension failed: ") + aligner->engine_api().strerror(rc ? rc : res.status));
    for (uint32_t i = 0; i < res.n_ext; ++i) {
        const vgk_extension& x = ext[res.ext_begin + i];
        GaplessExtension e;
        for (uint32_t k = 0; k < x.path_len; ++k) e.path.push_back(graph->handle_of(nodes[x.path_begin + k]));
        e.offset = x.offset; e.read_interval = {x.read_begin, x.read_end};
        e.mismatch_positions.assign(mism.begin() + x.mism_begin, mism.begin() + x.mism_begin + x.n_mismatches);
        e.score = x.score; e.left_full = x.left_full; e.right_full = x.right_full;
        e.state.forward.node = x.state[0]; e.state.forward.range = {x.state[1], x.state[2]};
        e.state.backward.node = x.state[3]; e.state.backward.range = {x.state[4], x.state[5]};
        result.push_back(std::move(e));
    }
    return result;
}
std::vector<GaplessExtender::TailTree> GaplessExtender::get_tail_forest(const GaplessExtension& extended_seed, size_t read_length, bool left_tails,
                                                                        size_t* longest_detectable_gap) const {
    std::vector<TailTree> to_return;
    if (extended_seed.path.empty()) return to_return;
    // the position to read out of the extension on this tail, the tail's length, the search state to start from (:5756-5782)
    vgk_tail_problem p{};
    size_t tail_length;
    if (left_tails) {
        const handle_t first = graph->flip(extended_seed.path.front());          // look right from the start, then the other way
        p.offset = (uint32_t)(graph->get_length(first) - extended_seed.offset);
        p.node = (uint32_t)extended_seed.state.backward.node; p.lo = (int32_t)extended_seed.state.backward.range.first; p.hi = (int32_t)extended_seed.state.backward.range.second;
        tail_length = extended_seed.read_interval.first;
    } else {
        size_t before_last = 0;
        for (size_t k = 0; k + 1 < extended_seed.path.size(); ++k) before_last += graph->get_length(extended_seed.path[k]);
        p.offset = (uint32_t)(extended_seed.offset + extended_seed.length() - before_last);      // tail_position: behind the last matched base
        p.node = (uint32_t)extended_seed.state.forward.node; p.lo = (int32_t)extended_seed.state.forward.range.first; p.hi = (int32_t)extended_seed.state.forward.range.second;
        tail_length = read_length - extended_seed.read_interval.second;
    }
    if (tail_length == 0) return to_return;                                      // (:5784-5787)
    size_t gap_limit;
    if (!longest_detectable_gap) longest_detectable_gap = &gap_limit;
    *longest_detectable_gap = aligner->scorer->longest_detectable_gap(read_length, tail_length);   // (:5809)
    p.walk_distance = (uint32_t)(*longest_detectable_gap + tail_length);         // (:5816)
    vgk_tail_result r{};
    vgk_forest* forest = nullptr;
    const EngineApi& api = aligner->engine_api();
    const int rc = api.tail_forest(aligner->engine_context(), index, &p, 1, &r, &forest);
    if (rc != VGK_OK || r.status != VGK_OK) { if (forest) api.forest_destroy(forest); throw std::runtime_error(std::string("vgamd: tail forest failed: ") + api.strerror(rc ? rc : r.status)); }
    std::vector<int32_t> parent(r.n_nodes + 1); std::vector<uint32_t> node(r.n_nodes + 1);
    const int rc2 = r.n_nodes ? api.forest_fetch(forest, parent.data(), node.data(), nullptr) : VGK_OK;
    api.forest_destroy(forest);
    if (rc2 != VGK_OK) throw std::runtime_error(std::string("vgamd: tail forest failed: ") + api.strerror(rc2));
    size_t tree_start = 0;
    for (uint32_t v = 0; v < r.n_nodes; ++v) {
        if (parent[v] < 0) { to_return.emplace_back(); to_return.back().root_trim = r.root_trim; tree_start = v; }      // nothing open above it: a new tree (:5826-5838)
        to_return.back().tree.emplace_back(parent[v] < 0 ? -1 : (int64_t)parent[v] - (int64_t)tree_start, graph->handle_of(node[v]));
    }
    return to_return;
}
bool GaplessExtender::full_length_extensions(const std::vector<GaplessExtension>& result, size_t max_mismatches) {
    return !result.empty() && result.front().full() && result.front().mismatches() <= max_mismatches;      // src/gbwt_extender.cpp:741-743
}


static vgk_haplo* upload_index(const HaplotypeGraph& g, const Aligner& a) {
    std::vector<uint32_t> node_len, thread_off{0}, thread_nodes; std::string seq;
    for (const std::string& s : g.sequences()) { node_len.push_back((uint32_t)s.size()); seq += s; }
    for (const auto& t : g.threads()) { thread_nodes.insert(thread_nodes.end(), t.begin(), t.end()); thread_off.push_back((uint32_t)thread_nodes.size()); }
    vgk_haplotypes d{};
    d.n_nodes = (uint32_t)node_len.size(); d.node_len = node_len.data(); d.seq = seq.data();
    d.n_threads = (uint32_t)g.threads().size(); d.thread_off = thread_off.data(); d.thread_nodes = thread_nodes.empty() ? thread_off.data() : thread_nodes.data();
    vgk_haplo* index = nullptr;
    const int rc = a.engine_api().haplo_create(a.engine_context(), &d, &index);
    if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: cannot index the haplotypes: ") + a.engine_api().strerror(rc));
    return index;
}

WFAExtender::WFAExtender(const HaplotypeGraph& g, const Aligner& a, const ErrorModel& em) : graph(&g), aligner(&a), error_model(&em) {
    // the reference asserts these (src/gbwt_extender.cpp:1256-1270)
    for (const ErrorModel::Event* e : { &em.mismatches, &em.gaps, &em.gap_length })
        if (e->per_base < 0 || e->min < 0 || e->max < e->min) throw std::invalid_argument("vgamd: WFAExtender error model does not make sense");
    index = upload_index(g, a);
}
WFAExtender::~WFAExtender() { if (index) aligner->engine_api().haplo_destroy(index); }

std::vector<WFAAlignment> WFAExtender::extend(const std::vector<Problem>& problems) const {
    std::vector<WFAAlignment> out(problems.size());
    std::vector<vgk_wfa_problem> ps; std::vector<size_t> which;
    size_t bases = 0;
    for (size_t i = 0; i < problems.size(); ++i) {
        const Problem& q = problems[i];
        const Position& anchor = q.kind == Problem::PREFIX ? q.to : q.from;
        if (!graph->has_node(anchor.node_id)) continue;                       // an empty (failed) alignment (:2059-2064, :2249-2251)
        vgk_wfa_problem p{};
        p.seq = q.sequence.c_str(); p.seq_len = (uint32_t)q.sequence.size();
        p.mode = q.kind == Problem::CONNECT ? VGK_WFA_CONNECT : q.kind == Problem::SUFFIX ? VGK_WFA_SUFFIX : VGK_WFA_PREFIX;
        p.from_node = p.to_node = VGK_WFA_NO_NODE;
        if (q.kind != Problem::PREFIX) { p.from_node = graph->oriented(graph->get_handle(q.from.node_id, q.from.is_reverse)); p.from_offset = (uint32_t)q.from.offset; }
        if (q.kind != Problem::SUFFIX) {
            if (graph->has_node(q.to.node_id)) { p.to_node = graph->oriented(graph->get_handle(q.to.node_id, q.to.is_reverse)); p.to_offset = (uint32_t)q.to.offset; }
            else { p.to_node = 0x7fffffffu; p.to_offset = 0; }               // a target no haplotype reaches
        }
        ps.push_back(p); which.push_back(i); bases += q.sequence.size();
    }
    if (ps.empty()) return out;
    vgk_wfa_error_model em;
    em.mismatches = { error_model->mismatches.per_base, error_model->mismatches.min, error_model->mismatches.max };
    em.gaps = { error_model->gaps.per_base, error_model->gaps.min, error_model->gaps.max };
    em.gap_length = { error_model->gap_length.per_base, error_model->gap_length.min, error_model->gap_length.max };
    em.distance = { error_model->distance.per_base, error_model->distance.min, error_model->distance.max };
    std::vector<vgk_wfa_result> res(ps.size());
    std::vector<uint32_t> paths(4 * bases + 64 * ps.size() + 1), edits(2 * bases + 8 * ps.size() + 1);
    size_t written[2];
    const int rc = aligner->engine_api().wfa_extend(aligner->engine_context(), index, &em, ps.data(), (uint32_t)ps.size(), res.data(), paths.data(), paths.size(),
                                                    edits.data(), edits.size(), written);
    if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: WFA extension failed: ") + aligner->engine_api().strerror(rc));
    for (size_t k = 0; k < ps.size(); ++k) {
        const vgk_wfa_result& r = res[k];
        if (r.status != VGK_OK && r.status != VGK_ENOBAND) throw std::runtime_error(std::string("vgamd: WFA extension failed: ") + aligner->engine_api().strerror(r.status));
        if (!r.ok) continue;
        WFAAlignment& a = out[which[k]];
        for (uint32_t j = 0; j < r.path_len; ++j) a.path.push_back(graph->handle_of(paths[r.path_begin + j]));
        for (uint32_t j = 0; j < r.n_edits; ++j) { const uint32_t e = edits[r.edit_begin + j]; a.edits.emplace_back((WFAAlignment::Edit)(e & 3u), e >> 2); }
        a.node_offset = r.node_offset; a.seq_offset = r.seq_offset; a.length = r.length; a.score = r.score; a.ok = true;
    }
    return out;
}
WFAAlignment WFAExtender::connect(std::string sequence, Position from, Position to) const {
    return extend({ Problem{ Problem::CONNECT, std::move(sequence), from, to } })[0];
}
WFAAlignment WFAExtender::suffix(const std::string& sequence, Position from) const {
    return extend({ Problem{ Problem::SUFFIX, sequence, from, Position() } })[0];
}
WFAAlignment WFAExtender::prefix(const std::string& sequence, Position to) const {
    return extend({ Problem{ Problem::PREFIX, sequence, Position(), to } })[0];
}

}  // namespace vgamd
