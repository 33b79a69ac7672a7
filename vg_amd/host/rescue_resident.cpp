// rescue_resident.cpp — see rescue_resident.hpp.
#include "rescue_resident.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

namespace vgamd {

ResidentRescueGraph::ResidentRescueGraph(const Aligner& a, uint32_t n, const uint32_t* nl, const char* sq, const uint32_t* pred_off, const uint32_t* pred_idx)
    : aligner(&a), n_nodes(n), node_len(nl), seq(sq), seq_off((size_t)n + 1, 0) {
    for (uint32_t v = 0; v < n; ++v) seq_off[v + 1] = seq_off[v] + nl[v];
    vgk_graph g{}; g.n_nodes = n; g.node_len = nl; g.seq = sq; g.pred_off = pred_off; g.pred_idx = pred_idx;
    const EngineApi& api = a.engine_api();
    const int rc = api.graph_create(a.engine_context(), &g, &dg);
    if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: vgk_graph_create failed: ") + api.strerror(rc));
}
ResidentRescueGraph::~ResidentRescueGraph() { if (dg) aligner->engine_api().graph_destroy(dg); }

namespace {

// body(lo, hi) over [0, n) in chunks on up to `threads` host threads; the first exception is rethrown on the caller after the join
template <class F> void for_chunks(size_t n, unsigned threads, F body) {
    if (!threads) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    const size_t CH = 256;
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>((n + CH - 1) / CH, 1));
    if (threads < 2 || n < 2 * CH) { for (size_t i = 0; i < n; i += CH) body(i, std::min(n, i + CH)); return; }      // (always chunk by chunk: callers keep per-chunk buffers)
    std::atomic<size_t> next{0}; std::atomic<bool> failed{false};
    std::mutex first_mutex; std::exception_ptr first;
    auto work = [&]() {
        try { for (size_t i; !failed.load(std::memory_order_relaxed) && (i = next.fetch_add(CH)) < n;) body(i, std::min(n, i + CH)); }
        catch (...) { std::lock_guard<std::mutex> hold(first_mutex); if (!first) first = std::current_exception(); failed.store(true, std::memory_order_relaxed); }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < threads; ++t) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
    if (first) std::rethrow_exception(first);
}

// Room for a round's op lists: ops_per_problem entries per problem is what the engine may write, a tenth of that is what it does write (the lists come
// packed behind each other) — so the room is never filled in (a std::vector would write 20 MB of zeros per batch, touching every page), only claimed.
struct OpBuffer {
    std::unique_ptr<vgk_op[]> p; size_t cap = 0;
    void need(size_t n) { if (n > cap) { p.reset(new vgk_op[n]); cap = n; } }
    vgk_op* data() const { return p.get(); }
};

// the stretch of `len` bases of a node against the read: pushes the match runs and single-base substitutions (eight bases per comparison while they agree)
inline uint64_t load8(const char* q) { uint64_t w; std::memcpy(&w, q, 8); return w; }

// an alignment as the fix-ups see it: mappings of edits (what Alignment / Path / Mapping / Edit hold, without the strings)
enum : uint8_t { E_MATCH = 0, E_SUB = 1, E_DEL = 2, E_INS = 3 };
struct FEdit { uint8_t kind; uint32_t len; };
struct FMapping { uint32_t node; int64_t offset; uint32_t first_edit, n_edits; };
struct FlatAlignment {
    std::vector<FMapping> maps; std::vector<FEdit> edits; int32_t score = 0;
    void clear() { maps.clear(); edits.clear(); score = 0; }
    void open(uint32_t node, int64_t offset) { maps.push_back(FMapping{node, offset, (uint32_t)edits.size(), 0}); }
    void push(uint8_t kind, uint32_t len) { edits.push_back(FEdit{kind, len}); ++maps.back().n_edits; }
};

// MatrixAlignmentScorer::score_contiguous_alignment (src/alignment_scorer.cpp; vg_standin/alignment_scorer.cpp:34-57), both bonuses allowed
int32_t score_contiguous(const FlatAlignment& a, const MatrixAlignmentScorer& sc) {
    int32_t score = 0; bool in_deletion = false;
    for (size_t i = 0; i < a.maps.size(); ++i) for (uint32_t j = 0; j < a.maps[i].n_edits; ++j) {
        const FEdit& e = a.edits[a.maps[i].first_edit + j];
        const bool at_an_end = (i == 0 && j == 0) || (i + 1 == a.maps.size() && j + 1 == a.maps[i].n_edits);
        if (e.kind == E_MATCH) { score += sc.match * (int32_t)e.len; in_deletion = false; }
        else if (e.kind == E_SUB) { score -= sc.mismatch * (int32_t)e.len; in_deletion = false; }
        else if (e.kind == E_DEL) { score -= in_deletion ? (int32_t)e.len * sc.gap_extension : sc.gap_open + ((int32_t)e.len - 1) * sc.gap_extension; in_deletion = true; }
        else if (!at_an_end) { score -= sc.gap_open + ((int32_t)e.len - 1) * sc.gap_extension; in_deletion = false; }
        else in_deletion = false;                                              // a soft clip
    }
    auto clipped = [&](bool left) {
        if (a.maps.empty()) return false;
        const FMapping& m = left ? a.maps.front() : a.maps.back();
        if (!m.n_edits) return false;
        return a.edits[left ? m.first_edit : m.first_edit + m.n_edits - 1].kind == E_INS;
    };
    if (!clipped(true)) score += sc.full_length_bonus;
    if (!clipped(false)) score += sc.full_length_bonus;
    return score;
}

// MinimizerMapper::fix_dozeu_end_deletions (src/minimizer_mapper.cpp:3519-3565) as rescue_fixups.cpp states it (the evident intent where the
// reference indexes the mappings with the edit index)
void fix_end_deletions(FlatAlignment& a) {
    size_t i = 0; uint32_t j = 0;
    for (; i < a.maps.size(); ++i) {
        const FMapping& m = a.maps[i];
        for (j = 0; j < m.n_edits; ++j) if (a.edits[m.first_edit + j].kind != E_DEL) break;
        if (j != m.n_edits) break;
    }
    if (i == a.maps.size()) { a.maps.clear(); a.edits.clear(); return; }
    if (i != 0 || j != 0) {
        FMapping& from = a.maps[i];
        uint64_t removed = 0;
        for (uint32_t k = 0; k < j; ++k) removed += a.edits[from.first_edit + k].len;
        from.first_edit += j; from.n_edits -= j;
        a.maps.erase(a.maps.begin(), a.maps.begin() + (std::ptrdiff_t)i);
        a.maps[0].offset += (int64_t)removed;
    }
    while (!a.maps.empty()) {
        FMapping& m = a.maps.back();
        while (m.n_edits && a.edits[m.first_edit + m.n_edits - 1].kind == E_DEL) --m.n_edits;
        if (!m.n_edits) a.maps.pop_back(); else break;
    }
}

inline char sanitized(char c) { return (c == 'A' || c == 'T' || c == 'G' || c == 'C' || c == 'N') ? c : 'N'; }      // nonATGCNtoN (src/utility.cpp:323-332)

}  // namespace

void run_rescue_stage_resident(const Aligner& aligner, const ResidentRescueGraph& G, const char* reads, size_t reads_bytes,
                               const std::vector<RescueRequestFlat>& requests, uint64_t max_cells, unsigned host_threads,
                               std::vector<RescueResult>& results, std::vector<vgk_op>* out_ops, std::vector<uint64_t>* out_ops_begin, RescueTiming* timing) {
    const size_t n = requests.size();
    results.assign(n, RescueResult{});
    const EngineApi& api = aligner.engine_api(); vgk_ctx* ctx = aligner.engine_context();
    const MatrixAlignmentScorer& sc = *aligner.scorer;
    if (aligner.xdrop_band) throw std::invalid_argument("vgamd: the resident rescue stage runs the exact X-drop extension (GSSWAligner::xdrop_band is for the per-graph path)");
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](double RescueTiming::*field) { const auto t = std::chrono::steady_clock::now(); if (timing) timing->*field += std::chrono::duration<double, std::milli>(t - t_lap).count(); t_lap = t; };

    // per request: what it is, the head once it is known
    enum : uint8_t { R_DONE = 0, R_SEEDED = 1, R_SCAN = 2 };
    struct State { uint8_t kind = R_DONE; bool have_head = false, fallback = false, rescore_fallback = false; uint32_t head_node = 0, head_ref = 0, head_query = 0, gap = 1; int32_t slot1 = -1, slot2 = -1, slot3 = -1; };
    std::vector<State> st(n);
    // ---- 0. which requests run at all (:3352-3381), dozeu's seed or the scan (calculate_seed_position / scan_seed_position)
    for_chunks(n, host_threads, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            const RescueRequestFlat& rq = requests[k]; RescueResult& out = results[k]; State& s = st[k];
            if (rq.node_lo >= rq.node_hi || rq.node_hi > G.n_nodes || !rq.read_len || rq.read_off + rq.read_len > reads_bytes) { out.status = 2; continue; }
            const uint64_t bases = G.seq_off[rq.node_hi] - G.seq_off[rq.node_lo];
            if (bases * rq.read_len > max_cells) { out.status = 1; continue; }             // (:3372-3381: refused, the pair keeps what it has)
            s.gap = (uint32_t)std::max<size_t>(1, std::min<size_t>(sc.longest_detectable_gap(rq.read_len, rq.read_len / 2), 65535));      // (:3383; clamped to >= 1: src/aligner.cpp:638)
            if (rq.seed_node >= (int64_t)rq.node_lo && rq.seed_node < (int64_t)rq.node_hi && rq.seed_end > rq.seed_begin) {
                s.kind = R_SEEDED;
                // the "upward" extension runs from the seed's first base towards the read's end; without a read part that way the seed is the head
                s.head_node = (uint32_t)rq.seed_node; s.head_ref = (uint32_t)rq.seed_offset; s.head_query = (uint32_t)rq.seed_begin; s.have_head = true;
            } else s.kind = R_SCAN;
        }
    });
    // ---- 1. first pass: the extension from the seed that finds the head (:654-672), or the 15-base scan (:143-208)
    std::vector<vgk_extension_problem> ext1; std::vector<uint32_t> ext1_of; std::vector<vgk_window_problem> scan; std::vector<uint32_t> scan_of;
    uint32_t max_nodes = 1;
    for (size_t k = 0; k < n; ++k) {
        const State& s = st[k]; const RescueRequestFlat& rq = requests[k];
        if (s.kind == R_DONE) continue;
        max_nodes = std::max(max_nodes, rq.node_hi - rq.node_lo);
        if (s.kind == R_SEEDED) {
            if ((uint32_t)rq.seed_begin >= rq.read_len) continue;                          // (query.empty(): the job does not run)
            vgk_extension_problem p{}; p.read_off = rq.read_off; p.read_len = rq.read_len; p.flags = VGK_XDROP_PINNED; p.first_node = rq.node_lo; p.n_nodes = rq.node_hi - rq.node_lo;
            p.max_gap_length = s.gap; p.start_node = (uint32_t)rq.seed_node; p.start_offset = (uint32_t)rq.seed_offset; p.query_offset = (uint32_t)rq.seed_begin; p.leftward = 0;
            st[k].slot1 = (int32_t)ext1.size(); ext1.push_back(p); ext1_of.push_back((uint32_t)k);
        } else {
            const uint32_t scan_len = std::min<uint32_t>(rq.read_len, 15);
            vgk_window_problem w{}; w.read_off = rq.read_off + rq.read_len - scan_len; w.read_len = scan_len; w.flags = VGK_GSSW_LOCAL; w.first_node = rq.node_lo; w.n_nodes = rq.node_hi - rq.node_lo;
            st[k].slot1 = (int32_t)scan.size(); scan.push_back(w); scan_of.push_back((uint32_t)k);
        }
    }
    lap(&RescueTiming::classify_ms);
    // one round of kernels: the extension windows and the plain windows of a pass side by side (two batches: consecutive batches of a context take its two launch lanes)
    const uint32_t OPS_PER = max_nodes + 96;
    auto check = [&](int rc, const char* what) { if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: rescue stage: ") + what + " failed: " + api.strerror(rc)); };
    auto round = [&](const std::vector<vgk_extension_problem>& ep, const std::vector<vgk_window_problem>& wp, bool traced,
                     std::vector<vgk_result>& er, OpBuffer& eo, std::vector<vgk_result>& wr, OpBuffer& wo) {
        vgk_batch* be = nullptr; vgk_batch* bw = nullptr;
        struct Free { const EngineApi& api; vgk_batch*& b; ~Free() { if (b) api.batch_free(b); } } fe{api, be}, fw{api, bw};
        if (!ep.empty()) check(api.gssw_pack_extensions(ctx, G.dg, reads, reads_bytes, ep.data(), (uint32_t)ep.size(), traced ? OPS_PER : 0, &be), "vgk_gssw_pack_extensions");
        if (!wp.empty()) check(api.gssw_pack_windows(ctx, G.dg, reads, reads_bytes, wp.data(), (uint32_t)wp.size(), traced ? OPS_PER : 0, &bw), "vgk_gssw_pack_windows");
        if (be) check(api.gssw_run(be), "vgk_gssw_run");
        if (bw) check(api.gssw_run(bw), "vgk_gssw_run");
        size_t written = 0;
        er.assign(ep.size(), vgk_result{}); wr.assign(wp.size(), vgk_result{});
        if (be) { eo.need(traced ? ep.size() * (size_t)OPS_PER + 1 : 1); const int rcf = api.gssw_fetch(be, er.data(), eo.data(), traced ? eo.cap : 0, &written); if (rcf != VGK_EOPS) check(rcf, "vgk_gssw_fetch"); }      // (VGK_EOPS: single problems' tracebacks outgrew their budgets — their statuses say which)
        if (bw) { wo.need(traced ? wp.size() * (size_t)OPS_PER + 1 : 1); const int rcf = api.gssw_fetch(bw, wr.data(), wo.data(), traced ? wo.cap : 0, &written); if (rcf != VGK_EOPS) check(rcf, "vgk_gssw_fetch"); }
        if (timing) for (vgk_batch* b : {be, bw}) if (b) { timing->kernel_ms += api.batch_kernel_ms(b, -1); timing->alg_bytes += api.batch_alg_bytes(b); timing->cells += api.batch_cells(b); }
        // (a single problem the engine declined — VGK_EOPS: more op runs than a traceback's budget, VGK_ETOOBIG ... — is its request's affair, not the
        // batch's: MinimizerMapper::attempt_rescue fails per pair and leaves the other pairs what they have.  The callers below look at r.status.)
    };
    std::vector<vgk_result> er, wr; OpBuffer eo, wo;
    round(ext1, scan, false, er, eo, wr, wo);
    if (timing) { timing->first_pass += ext1.size(); timing->scans += scan.size(); }
    // the heads (xdrop_extend_finish's end position; the scan's end cell: src/dozeu_interface.cpp:188-208)
    for (size_t q = 0; q < ext1.size(); ++q) {
        const vgk_result& r = er[q]; State& s = st[ext1_of[q]]; const RescueRequestFlat& rq = requests[ext1_of[q]];
        if (r.status != VGK_OK) { s.have_head = false; s.fallback = true; s.rescore_fallback = true; continue; }      // the engine declined this extension: the full DP for this mate
        if (r.score <= 0) continue;                                                         // the seed's own position stays the head
        const uint32_t en = rq.node_lo + (uint32_t)r.end_node, used = (uint32_t)r.end_offset + 1;
        s.head_ref = (en == (uint32_t)rq.seed_node ? (uint32_t)rq.seed_offset : 0u) + used; s.head_node = en; s.head_query = (uint32_t)rq.seed_begin + (uint32_t)r.end_read + 1;
    }
    for (size_t q = 0; q < scan.size(); ++q) {
        const vgk_result& r = wr[q]; State& s = st[scan_of[q]]; const RescueRequestFlat& rq = requests[scan_of[q]];
        if (r.status != VGK_OK || r.score <= 0) { s.fallback = true; s.rescore_fallback = true; continue; }      // dozeu's seeding heuristic failed (or the engine declined the scan): gssw instead (src/aligner.cpp:848-854), then fix_dozeu_score sees that alignment
        const uint32_t scan_len = std::min<uint32_t>(rq.read_len, 15);
        s.have_head = true; s.head_node = rq.node_lo + (uint32_t)r.end_node; s.head_ref = (uint32_t)r.end_offset + 1; s.head_query = (rq.read_len - scan_len) + (uint32_t)r.end_read + 1;
    }
    lap(&RescueTiming::first_pass_ms);
    // ---- 2. second pass: the traced extension from the head the other way (align_downward, :687-722)
    std::vector<vgk_extension_problem> ext2; std::vector<uint32_t> ext2_of; std::vector<vgk_window_problem> none;
    for (size_t k = 0; k < n; ++k) {
        State& s = st[k]; const RescueRequestFlat& rq = requests[k];
        if (s.kind == R_DONE || !s.have_head || s.head_query == 0) continue;               // (nothing of the read lies that way: the job does not run)
        vgk_extension_problem p{}; p.read_off = rq.read_off; p.read_len = rq.read_len; p.flags = VGK_XDROP_PINNED | VGK_GSSW_TRACEBACK; p.first_node = rq.node_lo; p.n_nodes = rq.node_hi - rq.node_lo;
        p.max_gap_length = s.gap; p.start_node = s.head_node; p.start_offset = s.head_ref; p.query_offset = s.head_query; p.leftward = 1;
        s.slot2 = (int32_t)ext2.size(); ext2.push_back(p); ext2_of.push_back((uint32_t)k);
    }
    round(ext2, none, true, er, eo, wr, wo);
    if (timing) timing->second_pass += ext2.size();
    lap(&RescueTiming::second_pass_ms);
    // ---- 3. the alignments (xdrop_extend_finish + xdrop_finish), fix_dozeu_score's verdict; what it accepts gets its answer at once
    // (one FlatAlignment per host thread, the op runs of a chunk of mates behind each other in the chunk's own buffer: no allocation per mate)
    const bool want_ops = out_ops && out_ops_begin;
    const char* const seq_end = G.seq + G.seq_off[G.n_nodes]; const char* const reads_end = reads + reads_bytes;
    const size_t CH = 256, n_chunks = (n + CH - 1) / CH;
    std::vector<std::vector<vgk_op>> chunk_ops(want_ops ? n_chunks : 0);
    std::vector<uint32_t> op_at(want_ops ? n : 0, 0), op_count(n, 0);                      // where in its chunk's buffer a mate's runs start; how many
    // fix_dozeu_end_deletions, the answer, the op runs (match / substitution stretches merged into M runs)
    auto answer = [&](FlatAlignment& a, size_t k, std::vector<vgk_op>* sink) {
        RescueResult& out = results[k];
        fix_end_deletions(a);
        out.score = a.score; out.n_mappings = (uint32_t)a.maps.size();
        if (!a.maps.empty()) { out.first_node = a.maps.front().node; out.first_offset = a.maps.front().offset; }
        uint32_t to = 0, runs = 0;
        for (const FMapping& m : a.maps) {
            int prev = -1;
            for (uint32_t j = 0; j < m.n_edits; ++j) {
                const FEdit& e = a.edits[m.first_edit + j];
                if (e.kind == E_MATCH || e.kind == E_SUB) to += e.len;
                const int op = e.kind <= E_SUB ? VGK_OP_M : e.kind == E_DEL ? VGK_OP_D : VGK_OP_I;
                if (op == VGK_OP_M && prev == VGK_OP_M) { if (sink) sink->back().len = (uint16_t)(sink->back().len + e.len); }
                else { ++runs; if (sink) { vgk_op o{}; o.node = m.node; o.op = (uint8_t)op; o.len = (uint16_t)e.len; sink->push_back(o); } }
                prev = op;
            }
        }
        out.aligned_read_bases = to; op_count[k] = runs;
    };
    for_chunks(n, host_threads, [&](size_t lo, size_t hi) {
        std::vector<vgk_op> ops; FlatAlignment a;
        std::vector<vgk_op>* sink = want_ops ? &chunk_ops[lo / CH] : nullptr;
        for (size_t k = lo; k < hi; ++k) {
            State& s = st[k]; const RescueRequestFlat& rq = requests[k];
            if (k + 2 < hi && st[k + 2].slot2 >= 0) {                                       // what the mate after next will read: its op list, its window's bases, its read
                const RescueRequestFlat& nx = requests[k + 2]; const vgk_result& nr = er[(size_t)st[k + 2].slot2];
                __builtin_prefetch(eo.data() + nr.ops_begin);
                const char* b = G.seq + G.seq_off[std::min<uint32_t>(st[k + 2].head_node, G.n_nodes - 1)] + st[k + 2].head_ref;      // (the pass runs leftwards from the head)
                for (int line = -3; line <= 1; ++line) __builtin_prefetch(b + 64 * line);
                __builtin_prefetch(reads + nx.read_off); __builtin_prefetch(reads + nx.read_off + 64); __builtin_prefetch(reads + nx.read_off + 128);
            }
            if (s.kind == R_DONE || !s.have_head) continue;
            if (s.slot2 >= 0 && er[(size_t)s.slot2].status != VGK_OK) { s.fallback = true; s.rescore_fallback = true; continue; }      // the traced pass was declined: the full DP for this mate
            const char* read = reads + rq.read_off;
            a.clear();
            int32_t down_score = 0;
            if (s.slot2 >= 0 && er[(size_t)s.slot2].score > 0 && er[(size_t)s.slot2].n_ops > 0) {
                const vgk_result& r = er[(size_t)s.slot2];
                down_score = r.score;
                // the pass ran on reversed strings: flip it back (unreverse_graph_mapping, src/aligner.cpp:255-300, on the op list)
                ops.assign(eo.data() + r.ops_begin, eo.data() + r.ops_begin + r.n_ops);
                std::reverse(ops.begin(), ops.end());
                const uint32_t first = ops[0].node; uint32_t aligned = 0, groups = 1;
                for (size_t i = 0; i < ops.size(); ++i) {
                    if (i && ops[i].node != ops[i - 1].node) ++groups;
                    if (ops[i].node == first && groups == 1 && (ops[i].op == VGK_OP_M || ops[i].op == VGK_OP_D)) aligned += ops[i].len;
                }
                const uint32_t first_abs = rq.node_lo + first;
                const uint32_t first_len = first_abs == s.head_node ? s.head_ref : G.node_len[first_abs];      // (the start node was cut at the head)
                int64_t from_pos = (int64_t)first_len - (int64_t)aligned - (groups == 1 ? (int64_t)r.first_offset : 0);
                size_t to_pos = 0; bool first_node = true;
                for (size_t i = 0; i < ops.size();) {
                    size_t j = i; while (j < ops.size() && ops[j].node == ops[i].node) ++j;
                    const uint32_t v = rq.node_lo + ops[i].node;
                    const char* node_seq = G.seq + G.seq_off[v];
                    int64_t fp = first_node ? from_pos : 0; first_node = false;
                    a.open(v, fp);
                    for (size_t q = i; q < j; ++q) {
                        const uint32_t len = ops[q].len;
                        if (ops[q].op == VGK_OP_M) {
                            uint32_t run = 0;
                            const char* x = node_seq + fp; const char* y = read + to_pos;
                            for (uint32_t t = 0; t < len;) {
                                if (t + 8 <= len && x + t + 8 <= seq_end && y + t + 8 <= reads_end) {
                                    const uint64_t d = load8(x + t) ^ load8(y + t);
                                    if (!d) { run += 8; t += 8; continue; }
                                    const uint32_t same = (uint32_t)(__builtin_ctzll(d) >> 3);
                                    run += same; t += same;
                                }
                                if (x[t] == y[t]) ++run;
                                else { if (run) { a.push(E_MATCH, run); run = 0; } a.push(E_SUB, 1); }
                                ++t;
                            }
                            if (run) a.push(E_MATCH, run);
                            fp += len; to_pos += len;
                        } else if (ops[q].op == VGK_OP_D) { a.push(E_DEL, len); fp += len; }
                        else {                                                               // I / S: merged with an insertion right before it in this mapping
                            FMapping& m = a.maps.back();
                            if (m.n_edits && a.edits.back().kind == E_INS) a.edits.back().len += len; else a.push(E_INS, len);
                            to_pos += len;
                        }
                    }
                    i = j;
                }
            }
            if (down_score <= 0 || a.maps.empty()) {
                a.clear(); a.open(s.head_node, (int64_t)s.head_ref); a.push(E_INS, rq.read_len);      // full-length insertion at the head position (:344-359)
            } else {
                a.score = down_score;
                if (s.head_query < rq.read_len) a.push(E_INS, rq.read_len - s.head_query);            // the read beyond the head was never shown to dozeu (:498-526)
            }
            // MinimizerMapper::fix_dozeu_score (:3502-3517)
            const int32_t rescored = score_contiguous(a, sc);
            if (rescored > 0) { a.score = rescored; if (sink) op_at[k] = (uint32_t)sink->size(); answer(a, k, sink); }
            else s.fallback = true;                                                                    // not worth keeping: the full DP instead (:3510-3515)
        }
    });
    lap(&RescueTiming::finish_ms);
    // ---- 4. the full DP for what asks for it: LOCAL gssw with a traceback over the whole rescue subgraph
    std::vector<vgk_window_problem> full; std::vector<uint32_t> full_of;
    for (size_t k = 0; k < n; ++k) {
        State& s = st[k]; const RescueRequestFlat& rq = requests[k];
        if (!s.fallback) continue;
        vgk_window_problem w{}; w.read_off = rq.read_off; w.read_len = rq.read_len; w.flags = VGK_GSSW_LOCAL | VGK_GSSW_TRACEBACK; w.first_node = rq.node_lo; w.n_nodes = rq.node_hi - rq.node_lo;
        s.slot3 = (int32_t)full.size(); full.push_back(w); full_of.push_back((uint32_t)k);
    }
    std::vector<vgk_extension_problem> no_ext;
    if (!full.empty()) round(no_ext, full, true, er, eo, wr, wo);
    if (timing) timing->fallbacks += full.size();
    std::vector<std::vector<vgk_op>> full_ops(want_ops ? full.size() : 0);                 // (the few mates that took the full DP keep their runs apart)
    for_chunks(full.size(), host_threads, [&](size_t lo, size_t hi) {
        FlatAlignment a;
        for (size_t q = lo; q < hi; ++q) {
            const size_t k = full_of[q]; State& s = st[k]; const RescueRequestFlat& rq = requests[k]; const vgk_result& r = wr[q];
            if (r.status != VGK_OK) { RescueResult& out = results[k]; out = RescueResult{}; out.status = 3; out.score = 0; out.first_node = r.status; op_count[k] = 0; continue; }   // this mate stays unrescued (status 3; first_node: the VGK_E* code)
            const char* read = reads + rq.read_off;
            // gssw_mapping_to_alignment (src/aligner.cpp:120-241) over the op list: matches and single-base substitutions by character (the node's
            // bases as gssw saw them: nonATGCNtoN), every insertion / soft clip an edit of its own
            a.clear(); a.score = r.score;
            int64_t from_pos = r.first_offset; size_t to_pos = 0; bool first_node = true;
            const vgk_op* ops = wo.data() + r.ops_begin;
            for (uint32_t i = 0; i < r.n_ops;) {
                uint32_t j = i; while (j < r.n_ops && ops[j].node == ops[i].node) ++j;
                const uint32_t v = rq.node_lo + ops[i].node;
                const char* node_seq = G.seq + G.seq_off[v];
                if (!first_node) from_pos = 0;
                first_node = false;
                a.open(v, from_pos);
                for (uint32_t p = i; p < j; ++p) {
                    const uint32_t len = ops[p].len;
                    if (ops[p].op == VGK_OP_M) {
                        uint32_t run = 0;
                        for (uint32_t t = 0; t < len; ++t) {
                            if (sanitized(node_seq[from_pos + t]) != read[to_pos + t]) { if (run) { a.push(E_MATCH, run); run = 0; } a.push(E_SUB, 1); }
                            else ++run;
                        }
                        if (run) a.push(E_MATCH, run);
                        from_pos += len; to_pos += len;
                    } else if (ops[p].op == VGK_OP_D) { a.push(E_DEL, len); from_pos += len; }
                    else { a.push(E_INS, len); to_pos += len; }
                }
                i = j;
            }
            if (s.rescore_fallback) {                                                        // (this alignment came from align_xdrop's own fallback: fix_dozeu_score looks at it)
                const int32_t rescored = score_contiguous(a, sc);
                if (rescored > 0) a.score = rescored;                                        // (else: cleared and aligned again — the same alignment, gssw's score)
            }
            answer(a, k, want_ops ? &full_ops[q] : nullptr);
        }
    });
    lap(&RescueTiming::fallback_ms);
    // ---- 5. the op runs, request by request
    if (want_ops) {
        std::vector<uint64_t>& begin = *out_ops_begin;
        begin.assign(n + 1, 0);
        for (size_t k = 0; k < n; ++k) begin[k + 1] = begin[k] + op_count[k];
        out_ops->resize((size_t)begin[n]);
        for_chunks(n, host_threads, [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                if (!op_count[k]) continue;
                const vgk_op* src = st[k].fallback ? full_ops[(size_t)st[k].slot3].data() : chunk_ops[k / CH].data() + op_at[k];
                std::copy(src, src + op_count[k], out_ops->data() + begin[k]);
            }
        });
    }
    lap(&RescueTiming::finish_ms);
}

}  // namespace vgamd
