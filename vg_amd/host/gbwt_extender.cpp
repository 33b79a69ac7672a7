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
    for (const auto& e : edges_) {
        if (!go_left && e.first == h.v) { if (!it(handle_t{e.second})) return false; }
        if (go_left && e.second == h.v) { if (!it(handle_t{e.first})) return false; }
    }
    return true;
}
bool HaplotypeGraph::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    for (nid_t id : ids_) if (!it(get_handle(id, false))) return false;
    return true;
}

// ---- GaplessExtension (src/gbwt_extender.cpp:17-151) -----------------------------------------------------------------
bool GaplessExtension::contains(const HandleGraph& graph, const seed_type& seed) const {
    size_t read_offset = read_interval.first, node_offset = offset;
    for (const handle_t& handle : path) {
        const size_t len = std::min(graph.get_length(handle) - node_offset, read_interval.second - read_offset);
        if (seed_type(handle, (int64_t)read_offset - (int64_t)node_offset) == seed) return true;
        read_offset += len; node_offset = 0;
    }
    return false;
}
Position GaplessExtension::starting_position(const HandleGraph& graph) const {
    Position p;
    if (empty()) return p;
    p.node_id = graph.get_id(path.front()); p.is_reverse = graph.get_is_reverse(path.front()); p.offset = (int64_t)offset;
    return p;
}
size_t GaplessExtension::tail_offset(const HandleGraph& graph) const {
    size_t result = offset + length();
    for (size_t i = 0; i + 1 < path.size(); ++i) result -= graph.get_length(path[i]);
    return result;
}
Position GaplessExtension::tail_position(const HandleGraph& graph) const {
    Position p;
    if (empty()) return p;
    p.node_id = graph.get_id(path.back()); p.is_reverse = graph.get_is_reverse(path.back()); p.offset = (int64_t)tail_offset(graph);
    return p;
}
size_t GaplessExtension::overlap(const HandleGraph& graph, const GaplessExtension& another) const {
    size_t result = 0, this_pos = read_interval.first, another_pos = another.read_interval.first;
    auto this_iter = path.begin(), another_iter = another.path.begin();
    size_t this_offset = offset, another_offset = another.offset;
    while (this_pos < read_interval.second && another_pos < another.read_interval.second) {
        if (this_pos == another_pos && *this_iter == *another_iter && this_offset == another_offset) {
            const size_t len = std::min({graph.get_length(*this_iter) - this_offset, read_interval.second - this_pos, another.read_interval.second - another_pos});
            result += len; this_pos += len; another_pos += len; ++this_iter; ++another_iter; this_offset = another_offset = 0;
        } else if (this_pos <= another_pos) { this_pos += graph.get_length(*this_iter) - this_offset; ++this_iter; this_offset = 0; }
        else { another_pos += graph.get_length(*another_iter) - another_offset; ++another_iter; another_offset = 0; }
    }
    return result;
}
Path GaplessExtension::to_path(const HandleGraph& graph, const std::string& sequence) const {
    Path result;
    auto mismatch = mismatch_positions.begin();
    size_t read_offset = read_interval.first, node_offset = offset;
    for (size_t i = 0; i < path.size(); ++i) {
        const size_t limit = std::min(read_offset + graph.get_length(path[i]) - node_offset, read_interval.second);
        result.mapping.emplace_back();
        Mapping& mapping = result.mapping.back();
        mapping.position.node_id = graph.get_id(path[i]); mapping.position.offset = (int64_t)node_offset; mapping.position.is_reverse = graph.get_is_reverse(path[i]);
        while (mismatch != mismatch_positions.end() && *mismatch < limit) {
            if (read_offset < *mismatch) { Edit e; e.from_length = e.to_length = (int32_t)(*mismatch - read_offset); mapping.edit.push_back(e); }
            Edit e; e.from_length = e.to_length = 1; e.sequence = std::string(1, sequence[*mismatch]); mapping.edit.push_back(e);
            read_offset = *mismatch + 1; ++mismatch;
        }
        if (read_offset < limit) { Edit e; e.from_length = e.to_length = (int32_t)(limit - read_offset); mapping.edit.push_back(e); read_offset = limit; }
        mapping.rank = (int64_t)i + 1;
        node_offset = 0;
    }
    return result;
}

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
bool GaplessExtender::full_length_extensions(const std::vector<GaplessExtension>& result, size_t max_mismatches) {
    return !result.empty() && result.front().full() && result.front().mismatches() <= max_mismatches;      // src/gbwt_extender.cpp:741-743
}

}  // namespace vgamd
