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

// ---- WFAAlignment (src/gbwt_extender.cpp:761-1123) --------------------------------------------------------------------
WFAAlignment WFAAlignment::from_extension(const GaplessExtension& extension) {
    WFAAlignment a;
    a.path = extension.path; a.node_offset = (uint32_t)extension.offset; a.seq_offset = (uint32_t)extension.read_interval.first;
    a.length = (uint32_t)extension.length(); a.score = extension.score; a.ok = true;
    size_t done = a.seq_offset;                       // sequence position after the last edit
    for (size_t at : extension.mismatch_positions) {
        if (!a.edits.empty() && done == at && a.edits.back().first == mismatch) ++a.edits.back().second;
        else {
            if (done < at) a.edits.emplace_back(match, (uint32_t)(at - done));
            a.edits.emplace_back(mismatch, 1u);
        }
        done = at;                                    // as the reference: the cursor stops ON the mismatch
    }
    if (done < a.seq_offset + a.length) a.edits.emplace_back(match, (uint32_t)(a.seq_offset + a.length - done));
    return a;
}
WFAAlignment WFAAlignment::make_unlocalized_insertion(size_t sequence_offset, size_t length, int score) {
    WFAAlignment a; a.edits.emplace_back(insertion, (uint32_t)length); a.seq_offset = (uint32_t)sequence_offset; a.length = (uint32_t)length; a.score = score; a.ok = true;
    return a;
}
WFAAlignment WFAAlignment::make_empty() { WFAAlignment a; a.ok = true; return a; }
bool WFAAlignment::unlocalized_insertion() const { return ok && path.empty() && edits.size() == 1 && edits.front().first == insertion; }
int64_t WFAAlignment::final_offset(const HandleGraph& graph) const {
    int64_t f = node_offset;
    for (const auto& e : edits) if (e.first != insertion) f += e.second;
    for (size_t i = 0; i + 1 < path.size(); ++i) f -= (int64_t)graph.get_length(path[i]);
    return f;
}
void WFAAlignment::flip(const HandleGraph& graph, const std::string& sequence) {
    seq_offset = (uint32_t)(sequence.length() - seq_offset - length);
    if (path.empty()) return;
    node_offset = (uint32_t)((int64_t)graph.get_length(path.back()) - final_offset(graph));
    std::reverse(path.begin(), path.end());
    for (handle_t& h : path) h = graph.flip(h);
    std::reverse(edits.begin(), edits.end());
}
void WFAAlignment::append(Edit edit, uint32_t len) {
    if (len == 0) return;
    if (edits.empty() || edits.back().first != edit) edits.emplace_back(edit, len);
    else edits.back().second += len;
}
void WFAAlignment::join(const WFAAlignment& second) {
    if (!ok) throw std::runtime_error("Cannot join onto an alignment that is not OK");
    if (!second.ok) throw std::runtime_error("Cannot join an alignment that is not OK onto another alignment");
    if (second.empty()) return;
    if (empty()) { *this = second; return; }
    if (seq_offset + length != second.seq_offset)
        throw std::runtime_error("Cannot join alignments because past-end position " + std::to_string(seq_offset + length) + " is not at start position " + std::to_string(second.seq_offset));
    if (path.empty() && !unlocalized_insertion()) throw std::runtime_error("Cannot join alignments because first alignment has no path");
    if (second.path.empty() && !second.unlocalized_insertion()) throw std::runtime_error("Cannot join alignments because second alignment has no path");
    if (edits.empty()) throw std::runtime_error("Cannot join alignments because first alignment has no edits");
    if (second.edits.empty()) throw std::runtime_error("Cannot join alignments because second alignment has no edits");
    if (!second.unlocalized_insertion()) {
        if (unlocalized_insertion()) { node_offset = second.node_offset; path.push_back(second.path.front()); }
        else if (second.node_offset == 0) path.push_back(second.path.front());
        else if (second.path.front() != path.back())
            throw std::runtime_error("Cannot join alignments because second alignment starts in the middle of a handle that first alignment doesn't end on");
        path.insert(path.end(), second.path.begin() + 1, second.path.end());
    }
    for (const auto& e : second.edits) append(e.first, e.second);
    length += second.length;
    score += second.score;
}
Path WFAAlignment::to_path(const HandleGraph& graph, const std::string& sequence) const {
    if (!ok) throw std::runtime_error("WFAAlignment is not OK and cannot become a path");
    if ((size_t)seq_offset + length > sequence.size()) throw std::runtime_error("WFAAlignment extends past end of sequence");
    Path result;
    if (unlocalized_insertion()) {
        result.mapping.emplace_back();
        vgamd::Edit e; e.to_length = (int32_t)edits.front().second; e.sequence = sequence.substr(seq_offset, edits.front().second);
        result.mapping.back().edit.push_back(e);
        return result;
    }
    if (path.empty()) return result;
    size_t seq_at = seq_offset, node_at = node_offset, step = 0;
    size_t node_end = graph.get_length(path[0]);
    if (node_offset >= node_end) throw std::runtime_error("WFAAlignment has offset to or past end of first node");
    if (edits.empty()) throw std::runtime_error("WFAAlignment has no edits");
    auto open_mapping = [&](size_t offset) {
        result.mapping.emplace_back();
        Position& p = result.mapping.back().position;
        p.node_id = graph.get_id(path[step]); p.is_reverse = graph.get_is_reverse(path[step]); p.offset = (int64_t)offset;
    };
    open_mapping(node_at);
    for (const auto& ed : edits) {
        if (ed.second == 0) throw std::runtime_error("WFAAlignment has empty edit");
        const bool uses_graph = ed.first != insertion, uses_seq = ed.first != deletion;
        size_t left = ed.second;
        while (left) {
            size_t take = left;
            if (uses_graph) {
                if (step == path.size()) throw std::runtime_error("WFAAlignment tried to go past end of path");
                if (node_at == node_end) throw std::runtime_error("WFAAlignment tried to go past end of node (" + std::to_string(node_end) + " bp)");
                take = std::min(take, node_end - node_at);
            }
            vgamd::Edit e;
            if (uses_graph) { e.from_length = (int32_t)take; node_at += take; }
            if (ed.first == mismatch || ed.first == insertion) {
                if (seq_at + take > (size_t)seq_offset + length) throw std::runtime_error("WFAAlignment uses more sequence than provided");
                e.sequence = sequence.substr(seq_at, take);
            }
            if (uses_seq) { e.to_length = (int32_t)take; seq_at += take; }
            result.mapping.back().edit.push_back(e);
            left -= take;
            if (uses_graph && node_at == node_end) {
                node_at = 0; ++step;
                if (step != path.size()) {
                    node_end = graph.get_length(path[step]);
                    if (node_end == 0) throw std::runtime_error("WFAAlignment has empty node " + std::to_string(graph.get_id(path[step])));
                    open_mapping(0);
                } else node_end = 0;
            }
        }
    }
    return result;
}

// ---- WFAExtender ------------------------------------------------------------------------------------------------------
const WFAExtender::ErrorModel WFAExtender::default_error_model { WFAExtender::ErrorModel::default_mismatches(), WFAExtender::ErrorModel::default_gaps(),
                                                                 WFAExtender::ErrorModel::default_gap_length(), WFAExtender::ErrorModel::default_distance() };

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
