// xdrop_band_api.cpp — vgk_xdrop_band_align: pinned X-drop extension WITH dozeu's band (include/vgk.h states the rules;
// reference call sites src/dozeu_interface.cpp:226, :261-283, src/xdrop_aligner.cpp:95-109).
//
// Device (gssw_matrix_device.hpp): xdrop_band_wave_lane fills a problem's columns — one of dozeu's 8-cell vectors per lane, four problems
// of up to 127 read bases to a wavefront (a DPP row of 16 lanes each), the front trimmed after every column, only the front's H and E
// vectors written (two bytes per column say where it lies), the last columns of the two nodes before a node kept in LDS — and finds the
// end cell; xdrop_band_walk_one, a kernel of its own with one lane per problem, walks the traceback over what the fill left in HBM; the
// ops are packed on the device and only results and ops come back (round 2 copied three planes and traced on host threads).  DESIGN.md
// §15 has the measurements.  VGK_XDROP_PINNED through vgk_gssw_* (every cell kept, 4-bit codes) stays the default and the fast path.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <emmintrin.h>
#include "ctx.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {

// page-locked staging, kept between calls: the inputs go up and the results come down at the full DMA rate (pageable vectors cost ~7 of 13 host
// ms per 200 000 tails).  Two sets: a call's sub-batches alternate between them (see below).
struct BandStage { PinnedBuf<uint8_t> reads, quals, graph, want; PinnedBuf<uint16_t> bucket; PinnedBuf<MProb> probs; PinnedBuf<MNode> nodes; PinnedBuf<uint32_t> preds, order; PinnedBuf<uint64_t> ops_off;
                   PinnedBuf<vgk_result> dres; PinnedBuf<vgk_op> dops; PinnedBuf<unsigned long long> stat; };
struct BandHost { BandStage set[2]; void* ev[2] = {nullptr, nullptr}; Backend* be = nullptr;
                  ~BandHost() { if (be) for (void* e : ev) if (e) be->event_destroy(e); } };
// device scratch slots of the two sets (ctx.hpp lists who owns which slot)
enum { D_PROBS, D_READS, D_QUALS, D_GRAPH, D_NODES, D_PREDS, D_MAT, D_CELLS, D_FMAX, D_STATS, D_FRONT, D_ORDER, D_RES, D_OPS, D_OPSOFF, D_WANT, D_OFFS, D_SUMS, D_PRES, D_POPS, D_COUNT };
constexpr int kSlot[2][D_COUNT] = { {40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 87, 80, 72, 73, 74, 75, 76, 77, 78, 79},
                                    {100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119} };
static_assert(119 < (int)(sizeof(vgk_ctx::scratch) / sizeof(vgk_ctx::DevBuf)), "scratch slots");

// one sub-batch of a call between its two halves: packed, uploaded and launched — then fetched and handed out
struct Sub { uint32_t i = 0, j = 0, m = 0; std::vector<uint32_t> owner; GsswMatrixParams P{}; uint64_t ops_total = 0; bool launched = false; };

}  // namespace

extern "C" {

// A call runs as a pipeline of sub-batches, two in flight: while the kernels of one run, the host packs the next and hands out the one
// before — pack, upload and launch on the main stream; the packed ops and results come back on the fetch stream behind an event.  (One
// sub-batch at a time, the host's 9 ms of checking, packing and copying per 200 000 tails stood beside 4.5 ms of kernels: 15 ms a call.)
int vgk_xdrop_band_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                         vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written, uint64_t stats[2]) try {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    if (ops_written) *ops_written = 0;
    if (stats) stats[0] = stats[1] = 0;
    for (uint32_t i = 0; i < n; ++i) if ((problems[i].flags & 15u) != VGK_XDROP_PINNED) return VGK_EINVAL;
    if (ctx->sc.gap_open < ctx->sc.gap_extend) return VGK_EUNSUPPORTED;          // the column scan needs gap_open >= gap_extend
    std::lock_guard<std::mutex> lock(ctx->mu);
    Backend* be = ctx->be.get();
    const bool qa = ctx->has_qa;
    ctx->xband_class[0] = ctx->xband_class[1] = ctx->xband_class[2] = 0;
    if (!ctx->xband_host) ctx->xband_host = std::make_shared<BandHost>();
    BandHost& Hs = *static_cast<BandHost*>(ctx->xband_host.get());
    if (!Hs.be) { Hs.be = be; for (void*& e : Hs.ev) e = be->event_create(); }      // (null events: a backend without streams — every wait is then a sync)
    uint64_t budget = be->memory_bytes() ? be->memory_bytes() / 8 : (1ull << 30);
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    const bool timing = std::getenv("VGAMD_XBAND_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; const auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[vgk_xdrop_band_align] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
    // per-problem checks; what fails is answered in its own status
    std::vector<int> status(n, VGK_OK); std::vector<uint64_t> cols(n, 0);
    std::vector<uint32_t> n_pred_of(n, 0), len_of(n, 0), nodes_of(n, 0);      // (what the serial placing loops below need, side by side: no second walk through every problem's pointers)
    // 16-bit cells in the planes (GsswMatrixParams::xb_cell16) when no reachable cell of any problem can leave int16's range on either side:
    // every step of an alignment costs or earns at most |score| + gap_open + gap_extend, and a cell is L + columns steps from the root at most
    int32_t step_max = 0, score_abs = 0; bool bytes_ok = true;      // bytes_ok: every score + score_abs (+ the bonus) is a byte — what the packed fill's profile holds
    { const int8_t* m = qa ? ctx->qmat.data() : ctx->sc.matrix; const size_t nm = qa ? 6400 : 25;
      for (size_t k = 0; k < nm; ++k) score_abs = std::max<int32_t>(score_abs, m[k] < 0 ? -m[k] : m[k]);
      int32_t bonus = ctx->sc.full_length_bonus, bonus_min = bonus;
      if (qa) for (int q = 0; q < 256; ++q) { bonus = std::max<int32_t>(bonus, ctx->qbon[q]); bonus_min = std::min<int32_t>(bonus_min, ctx->qbon[q]); }
      bytes_ok = bonus_min >= 0 && 2 * score_abs + bonus <= 255 && ctx->sc.gap_extend < 64;
      step_max = score_abs + (int32_t)ctx->sc.gap_open + (int32_t)ctx->sc.gap_extend + (bonus > 0 ? bonus : -bonus); }
    std::atomic<uint32_t> beyond16{0};
    parallel_for(n, [&](uint32_t i, unsigned) {
        if (i + 3 < n) { const vgk_gssw_problem& q = problems[i + 3]; __builtin_prefetch(q.graph.node_len); __builtin_prefetch(q.graph.pred_off); __builtin_prefetch(q.graph.pred_idx); }
        const vgk_gssw_problem& p = problems[i]; const vgk_graph& g = p.graph;
        int st = VGK_OK;
        if (!p.read_len || !p.read || !g.n_nodes || !g.node_len || !g.pred_off || !g.seq || (qa && !p.qual)) st = VGK_EINVAL;
        else if (p.read_len > 511) st = VGK_ETOOLONG;             // 64 lanes x one 8-row vector, rows 0 .. L
        else {
            uint64_t R = 0;
            for (uint32_t v = 0; v < g.n_nodes && st == VGK_OK; ++v) {
                if (!g.node_len[v] || g.pred_off[v + 1] < g.pred_off[v] || (g.pred_off[v + 1] > g.pred_off[v] && !g.pred_idx)) st = VGK_EINVAL;
                else if (g.node_len[v] > 65535u) st = VGK_ETOOBIG;
                else for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) if (g.pred_idx[k] >= v) st = VGK_EINVAL;
                R += g.node_len[v];
            }
            if (st == VGK_OK && R * (p.read_len + 1ull) > (1ull << 28)) st = VGK_ETOOBIG;
            cols[i] = R;
            if (st == VGK_OK && (uint64_t)step_max * (R + p.read_len + 10ull) >= 16000ull) beyond16.fetch_add(1, std::memory_order_relaxed);
            if (st == VGK_OK) { n_pred_of[i] = g.pred_off[g.n_nodes] - g.pred_off[0]; len_of[i] = p.read_len; nodes_of[i] = g.n_nodes; }
        }
        status[i] = st;
    });
    lap("check");
    const bool cell16 = beyond16.load() == 0 && !std::getenv("VGAMD_XBAND_CELLS32");
    const int cell_form = !cell16 ? 0 : (bytes_ok && !std::getenv("VGAMD_XBAND_ARITH32") ? 2 : 1);
    const size_t cell_bytes = cell16 ? sizeof(int16_t) : sizeof(int32_t);
    // sub-batches: what the device budget allows, and for a large call no more than a quarter of it, so that there is something to overlap
    uint32_t sub_cap = n, n_ok = 0;
    for (uint32_t i = 0; i < n; ++i) n_ok += status[i] == VGK_OK;      // (counted here: one counter bumped by every thread of the check was 9 of its 10 ms)
    if (n_ok >= 32768u && !std::getenv("VGAMD_XBAND_ONE_BATCH")) sub_cap = (n_ok + 3u) / 4u;
    size_t used = 0; int rc_all = VGK_OK; uint64_t in_band_total = 0, rect_total = 0; double ms = 0;
    const bool no_tb = std::getenv("VGAMD_XBAND_SCORES_ONLY") != nullptr;      // (a measuring aid: the fill and the end cell without the walk)

    // ---- first half of a sub-batch: the problems from `from` on that fit, packed, uploaded, launched
    auto build = [&](uint32_t from, Sub& S, int set) -> int {
        BandStage& St = Hs.set[set]; const int* slot = kSlot[set];
        uint64_t n_cells = 0, n_read = 0, n_graph = 0, n_nodes = 0, n_preds = 0;
        S.i = from; S.owner.clear(); S.launched = false; S.ops_total = 0;
        // where the sub-batch ends: the running sum of cells and the count decide it (two values per problem)
        uint32_t j = from, taken = 0;
        for (; j < n; ++j) {
            if (status[j] != VGK_OK) continue;
            const uint64_t c3 = 2ull * cols[j] * ((len_of[j] + 8ull) & ~7ull);               // H and E planes; a column is whole 8-row vectors
            if (taken && ((n_cells + c3) * cell_bytes > budget || taken >= sub_cap)) break;
            n_cells += c3; ++taken;
        }
        S.j = j;
        const uint32_t m = S.m = taken;
        if (!m) return VGK_OK;
        // the places: sums over chunks of [from, j) on the host threads, then every chunk places and packs its own problems
        struct Sums { uint64_t ok, cells, read, graph, nodes, preds, ops, rect; };
        const uint32_t span = j - from, n_chunks = chunk_count(span);
        std::vector<Sums> pre(n_chunks + 1, Sums{0, 0, 0, 0, 0, 0, 0, 0});
        auto add = [&](Sums& t, uint32_t q) {
            const uint64_t L = len_of[q], R = cols[q];
            t.ok += 1; t.cells += 2ull * R * ((L + 8ull) & ~7ull); t.read += L; t.graph += R; t.nodes += nodes_of[q]; t.preds += n_pred_of[q]; t.ops += L + R + 3ull; t.rect += R * (L + 1ull);
        };
        parallel_chunks(span, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Sums t{0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t q = from + lo; q < from + hi; ++q) if (status[q] == VGK_OK) add(t, q);
            pre[c + 1] = t;
        });
        for (uint32_t c = 0; c < n_chunks; ++c) { Sums& t = pre[c + 1]; const Sums& u = pre[c];
            t.ok += u.ok; t.cells += u.cells; t.read += u.read; t.graph += u.graph; t.nodes += u.nodes; t.preds += u.preds; t.ops += u.ops; t.rect += u.rect; }
        const Sums all = pre[n_chunks];
        n_read = all.read; n_graph = all.graph; n_nodes = all.nodes; n_preds = all.preds; rect_total += all.rect;
        S.owner.resize(m);
        std::vector<uint32_t>& owner = S.owner;
        MProb* probs = St.probs.get(be, m + 1);
        uint8_t* reads = St.reads.get(be, n_read + 1); uint8_t* quals = qa ? St.quals.get(be, n_read + 1) : nullptr; uint8_t* graph = St.graph.get(be, n_graph + 1);
        MNode* nodes = St.nodes.get(be, n_nodes + 1); uint32_t* preds = St.preds.get(be, n_preds + 1);
        uint64_t* ops_off = St.ops_off.get(be, m + 1); uint8_t* want = St.want.get(be, m + 1); uint16_t* bucket_of = St.bucket.get(be, m + 1);
        if (!probs || !reads || (qa && !quals) || !graph || !nodes || !preds || !ops_off || !want || !bucket_of) return VGK_ENOMEM;
        ops_off[m] = all.ops;
        parallel_chunks(span, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Sums at = pre[c];
            for (uint32_t q = from + lo; q < from + hi; ++q) {
                if (status[q] != VGK_OK) continue;
                // (a problem's arrays are five pointers into the caller's memory, each a miss: the ones of the problem three ahead are asked for now)
                if (q + 3 < from + hi) { const vgk_gssw_problem& nx = problems[q + 3];
                                         __builtin_prefetch(nx.graph.node_len); __builtin_prefetch(nx.graph.pred_off); __builtin_prefetch(nx.graph.pred_idx); __builtin_prefetch(nx.read);
                                         __builtin_prefetch(nx.graph.seq); __builtin_prefetch(nx.graph.seq + 64); __builtin_prefetch(nx.graph.seq + 128); }
                const uint32_t a = (uint32_t)at.ok;
                const vgk_gssw_problem& p = problems[q];
                MProb pb{}; pb.L = len_of[q]; pb.n_nodes = nodes_of[q]; pb.R = (uint32_t)cols[q];
                pb.read_off = (uint32_t)at.read; pb.graph_off = (uint32_t)at.graph; pb.node_off = (uint32_t)at.nodes; pb.mat_off = at.cells;
                owner[a] = q; ops_off[a] = at.ops;
                want[a] = (p.flags & VGK_GSSW_TRACEBACK) && !no_tb ? 1 : 0;
                { constexpr uint32_t B = 4096; const uint32_t r = pb.R < B ? pb.R : B - 1; bucket_of[a] = (uint16_t)((pb.L <= 127u ? 0u : B) + (B - 1 - r)); }      // (launch order, below)
                pb.start_bonus = qa ? ctx->qbon[p.qual[p.read_len - 1]] : ctx->sc.full_length_bonus; pb.status = VGK_OK;
                const int32_t max_gap = (int32_t)std::max<uint32_t>(p.max_gap_length, 1u);
                pb.gap_cells = (max_gap + 7) & ~7; pb.xt = ((int32_t)ctx->sc.gap_open - (int32_t)ctx->sc.gap_extend) + (int32_t)ctx->sc.gap_extend * max_gap;
                uint32_t col = 0; uint64_t a_preds = at.preds;
                for (uint32_t v = 0; v < p.graph.n_nodes; ++v) {
                    MNode nd{}; nd.col_start = col; nd.col_end = col + p.graph.node_len[v]; nd.pred_begin = (uint32_t)a_preds; nd.n_pred = p.graph.pred_off[v + 1] - p.graph.pred_off[v];
                    for (uint32_t k = p.graph.pred_off[v]; k < p.graph.pred_off[v + 1]; ++k) preds[a_preds++] = p.graph.pred_idx[k];
                    nodes[pb.node_off + v] = nd; col = nd.col_end;
                }
                probs[a] = pb;
                code_bases<true>(reads + pb.read_off, p.read, pb.L);
                if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
                code_bases<false>(graph + pb.graph_off, p.graph.seq, pb.R);
                add(at, q);
            }
        });
        lap("pack");
        GsswMatrixParams& P = S.P; P = GsswMatrixParams{};
        P.n = m; P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend;
        auto dev = [&](int what, const void* src, size_t bytes) -> void* {
            void* d = ctx->ensure_scratch(slot[what], std::max<size_t>(bytes, 16)); if (!d) return nullptr;
            if (src && bytes && be->upload_side(d, src, bytes)) return nullptr;      // (the side stream: under the kernels of the sub-batch before; the launch below waits for them on the device)
            return d;
        };
        // (set 0's slots are the scratch slots of the k-best pinned path: the two calls never overlap under the context lock)
        P.probs = (MProb*)dev(D_PROBS, probs, sizeof(MProb) * m);
        P.reads = (const uint8_t*)dev(D_READS, reads, n_read); P.quals = qa ? (const uint8_t*)dev(D_QUALS, quals, n_read) : nullptr;
        P.graph = (const uint8_t*)dev(D_GRAPH, graph, n_graph); P.nodes = (const MNode*)dev(D_NODES, nodes, sizeof(MNode) * n_nodes);
        P.preds = (const uint32_t*)dev(D_PREDS, preds, sizeof(uint32_t) * n_preds);
        P.mat = (const int8_t*)dev(D_MAT, qa ? ctx->qmat.data() : ctx->sc.matrix, qa ? 6400 : 25);
        P.cells = (int32_t*)dev(D_CELLS, nullptr, cell_bytes * n_cells + 64); P.xb_cell16 = cell_form; P.xb_sb = score_abs;
        P.node_fmax = (int32_t*)dev(D_FMAX, nullptr, sizeof(int32_t) * (n_nodes + 1));
        P.stats = (unsigned long long*)dev(D_STATS, nullptr, 64);
        P.xb_front = (uint16_t*)dev(D_FRONT, nullptr, sizeof(uint16_t) * (n_graph + 1));
        // the wavefront that fills a problem also picks its end cell, a second kernel walks the tracebacks: results and ops come back, the
        // matrices stay where they are
        if (ops_off[m] >= (1ull << 32)) return VGK_ETOOBIG;
        S.ops_total = ops_off[m];
        // launch order: tails of up to 127 bases four to a wavefront (16 lanes each), the longer ones a wavefront each; inside a class
        // by descending graph size (a counting sort), so that the problems sharing a wavefront take about equally long
        uint32_t* order = St.order.get(be, m + 1);
        if (!order) return VGK_ENOMEM;
        // (the packed fill has a third class in front: tails of at most 63 bases — 64 rows, 8 lanes of 8 — eight to a wavefront: a 40-base
        // tail uses 6 of its 16 lanes.  Built, exact (tests/test_xdrop_band.py, 6 000 problems around the class boundaries on the MI355X), and OFF
        // unless VGAMD_XBAND_EIGHTS=1: the fills + walks of 200 000 bench tails take 5.19-5.31 ms with it against 4.82-4.98 ms without
        // (profiles/r05/NOTES.md) — what differs between the groups of a wavefront (node boundaries, predecessor fronts, the root column) runs
        // group after group, twice as many of them, and the kernel's time is its stores and their latency, not its lanes)
        uint32_t n16 = 0, n8 = 0;
        { constexpr uint32_t B = 4096;
          const bool eights = cell_form == 2 && std::getenv("VGAMD_XBAND_EIGHTS") != nullptr;
          std::vector<uint32_t> count(3 * B + 1, 0);
          // (bucket_of, made by the packing threads: 0 .. B - 1 the short tails by descending graph size, B .. 2 B - 1 the long ones)
          auto bucket = [&](uint32_t a) { const uint32_t b = bucket_of[a]; return b >= B ? b + B : (eights && probs[a].L <= 63u ? b : b + B); };
          for (uint32_t a = 0; a < m; ++a) { const uint32_t b = bucket(a); ++count[b + 1]; if (b < B) ++n8; else if (b < 2 * B) ++n16; }
          for (uint32_t b = 0; b < 3 * B; ++b) count[b + 1] += count[b];
          for (uint32_t a = 0; a < m; ++a) order[count[bucket(a)]++] = a; }
        P.xb_order = (const uint32_t*)dev(D_ORDER, order, sizeof(uint32_t) * m); P.xb_n8 = n8; P.xb_n16 = n16; P.xb_n64 = m - n16 - n8;
        ctx->xband_class[0] += n8; ctx->xband_class[1] += n16; ctx->xband_class[2] += m - n16 - n8;
        P.xb_results = (vgk_result*)dev(D_RES, nullptr, sizeof(vgk_result) * m);
        P.xb_ops = (vgk_op*)dev(D_OPS, nullptr, sizeof(vgk_op) * ops_off[m]);
        P.xb_ops_off = (const uint64_t*)dev(D_OPSOFF, ops_off, sizeof(uint64_t) * m);
        P.xb_want_tb = (const uint8_t*)dev(D_WANT, want, m);
        if (!P.probs || !P.reads || (qa && !P.quals) || !P.graph || !P.nodes || !P.preds || !P.mat || !P.cells || !P.node_fmax || !P.stats ||
            !P.xb_order || !P.xb_results || !P.xb_ops || !P.xb_ops_off || !P.xb_want_tb || !P.xb_front) return VGK_ENOMEM;
        int rc;
        if ((rc = be->zero(P.stats, 64)) || (rc = be->main_after_side())) return rc;
        if ((rc = be->run_xdrop_band_async(P, set))) return rc;
        if ((rc = be->event_record(Hs.ev[set]))) return rc;
        S.launched = true;
        lap("launch");
        return VGK_OK;
    };

    // ---- second half: wait for the kernels, pack the ops on the device, bring results and ops back, hand them out in the caller's order
    auto finish = [&](Sub& S, int set) -> int {
        BandStage& St = Hs.set[set]; const int* slot = kSlot[set];
        const uint32_t m = S.m; const GsswMatrixParams& P = S.P;
        vgk_result* dres = St.dres.get(be, m + 1); vgk_op* dops = nullptr;
        if (!dres) return VGK_ENOMEM;
        if (m) {
            int rc;
            if (Hs.ev[set]) { if ((rc = be->fetch_after(Hs.ev[set]))) return rc; }
            else if ((rc = be->sync())) return rc;
            auto dev = [&](int what, size_t bytes) -> void* { return ctx->ensure_scratch(slot[what], std::max<size_t>(bytes, 16)); };
            const uint32_t blocks = (m + Backend::OPS_SCAN_BLOCK - 1) / Backend::OPS_SCAN_BLOCK;
            uint32_t* offs = (uint32_t*)dev(D_OFFS, sizeof(uint32_t) * m);
            uint32_t* sums = (uint32_t*)dev(D_SUMS, sizeof(uint32_t) * (blocks + 8));
            if (!offs || !sums) return VGK_ENOMEM;
            uint64_t total = 0;
            rc = be->ops_offsets(P.xb_results, m, offs, sums, &total);                   // (on the fetch stream, behind the event: waits for this sub-batch's kernels only)
            if (rc == VGK_OK) {
                vgk_result* pres_d = (vgk_result*)dev(D_PRES, sizeof(vgk_result) * m);
                vgk_op* pops_d = (vgk_op*)dev(D_POPS, sizeof(vgk_op) * std::max<uint64_t>(total, 1));
                if (!pres_d || !pops_d) return VGK_ENOMEM;
                if ((rc = be->ops_gather(P.xb_results, P.xb_ops, m, offs, sums, pres_d, pops_d))) return rc;
                dops = St.dops.get(be, total + 1);
                unsigned long long* stat = St.stat.get(be, 8);
                if (!dops || !stat) return VGK_ENOMEM;
                if ((rc = be->download_fetch_async(dres, pres_d, sizeof(vgk_result) * m))) return rc;
                if (total && (rc = be->download_fetch_async(dops, pops_d, sizeof(vgk_op) * total))) return rc;
                if ((rc = be->download_fetch_async(stat, P.stats, sizeof(unsigned long long)))) return rc;
                if ((rc = be->sync_fetch())) return rc;
                in_band_total += stat[0];
                // (ops_gather zeroes n_ops of failed problems and keeps their status)
            } else if (rc == VGK_EUNSUPPORTED) {                                          // a backend without the packing kernels hands the windows over
                dops = St.dops.get(be, S.ops_total + 1);
                if (!dops) return VGK_ENOMEM;
                unsigned long long in_band = 0;
                if ((rc = be->download(dres, P.xb_results, sizeof(vgk_result) * m))) return rc;
                if (S.ops_total && (rc = be->download(dops, P.xb_ops, sizeof(vgk_op) * S.ops_total))) return rc;
                if ((rc = be->download(&in_band, P.stats, sizeof in_band))) return rc;
                in_band_total += in_band;
            } else return rc;
            ms += be->xdrop_band_ms(set);
            lap("fetch");
        }
        // the caller's order.  A problem's place among the sub-batch's results is the number of accepted problems before it, its ops' place
        // the sum of theirs: both from sums over chunks on the host threads when everything fits the caller's op buffer (the usual case);
        // otherwise the running sum decides problem by problem which ones still fit
        const uint32_t i = S.i, j = S.j, cnt = j - i, n_chunks = chunk_count(cnt);
        struct Tot { uint64_t taken, ops; };
        std::vector<Tot> tot(n_chunks + 1, Tot{0, 0});
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            uint64_t t = 0;
            for (uint32_t k = lo; k < hi; ++k) t += status[i + k] == VGK_OK;
            tot[c + 1].taken = t;
        });
        for (uint32_t c = 0; c < n_chunks; ++c) tot[c + 1].taken += tot[c].taken;
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            uint64_t t = 0;
            for (uint64_t a = tot[c].taken; a < tot[c + 1].taken; ++a) if (dres[a].status == VGK_OK) t += dres[a].n_ops;
            (void)lo; (void)hi;
            tot[c + 1].ops = t;
        });
        for (uint32_t c = 0; c < n_chunks; ++c) tot[c + 1].ops += tot[c].ops;
        const bool all_fit = (ops || !tot[n_chunks].ops) && used + tot[n_chunks].ops <= ops_cap;
        std::vector<uint32_t> from(cnt, 0);
        if (!all_fit) {
            uint32_t a = 0;
            for (uint32_t q = i; q < j; ++q) {
                vgk_result& r = results[q];
                if (status[q] != VGK_OK) { std::memset(&r, 0, sizeof r); r.status = status[q]; r.ops_begin = (uint32_t)used; continue; }
                const uint32_t mine = a++;
                r = dres[mine];
                from[q - i] = r.ops_begin;
                r.ops_begin = (uint32_t)used;
                if (r.status != VGK_OK) { r.n_ops = 0; continue; }
                if (used + r.n_ops > ops_cap || (!ops && r.n_ops)) { r.status = VGK_EOPS; r.n_ops = 0; rc_all = VGK_EOPS; continue; }
                used += r.n_ops;
            }
        }
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            uint64_t mine = tot[c].taken, at = used + tot[c].ops;
            for (uint32_t k = lo; k < hi; ++k) {
                vgk_result& r = results[i + k];
                if (all_fit) {
                    if (status[i + k] != VGK_OK) { std::memset(&r, 0, sizeof r); r.status = status[i + k]; r.ops_begin = (uint32_t)at; continue; }
                    r = dres[mine++];
                    from[k] = r.ops_begin; r.ops_begin = (uint32_t)at;
                    if (r.status != VGK_OK) { r.n_ops = 0; continue; }
                    at += r.n_ops;
                }
                if (r.status == VGK_OK && r.n_ops) std::memcpy(ops + r.ops_begin, dops + from[k], sizeof(vgk_op) * r.n_ops);
            }
        });
        if (all_fit) used += tot[n_chunks].ops;
        lap("results");
        return VGK_OK;
    };

    Sub subs[2]; int set = 0; bool pending = false; int rc = VGK_OK;
    for (uint32_t i = 0; i < n && rc == VGK_OK;) {
        Sub& S = subs[set];
        rc = build(i, S, set);                                    // (its kernels are queued behind the previous sub-batch's)
        i = S.j;
        if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
        pending = rc == VGK_OK;
        set ^= 1;
    }
    if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
    if (rc != VGK_OK) { be->sync_side(); be->sync(); be->sync_fetch(); return rc; }      // (nothing of this call stays in flight behind an error)
    ctx->xband_ms = ms; ctx->xband_cells = !cell16 ? 4 : (cell_form == 2 ? 2 : 3);
    if (ops_written) *ops_written = used;
    if (stats) { stats[0] = in_band_total; stats[1] = rect_total; }
    return rc_all;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_xdrop_band_last_ms(vgk_ctx* ctx) { return ctx ? ctx->xband_ms : 0.0; }
int vgk_xdrop_band_last_cells(vgk_ctx* ctx) { return ctx ? ctx->xband_cells : 4; }
uint64_t vgk_xdrop_band_last_class(vgk_ctx* ctx, int which) { return ctx && which >= 0 && which < 3 ? ctx->xband_class[which] : 0; }

}  // extern "C"
