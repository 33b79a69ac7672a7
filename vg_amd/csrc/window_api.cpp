// window_api.cpp — vgk_graph_create / vgk_gssw_pack_windows: one graph resident in HBM, problems as windows of it, packed
// on the device (gssw_pack_device.hpp).
//
// This is the part of vg's per-read work that sits between "the seeds say the read belongs near here" and "fill the DP":
// Mapper::align_cluster / align_maybe_flip cut a subgraph around the cluster (src/mapper.cpp:2445-2518) and
// GSSWAligner::create_gssw_graph converts it (src/aligner.cpp:30-85).  With the graph resident the host's share per read is
// 32 + read_len bytes of memcpy into page-locked staging; everything else is derived by kernels from the resident tables.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>
#include "batch.hpp"
#include "dgraph.hpp"
#include "host_parallel.hpp"


namespace {

inline int ref_code(char ch) {      // after nonATGCNtoN (src/aligner.cpp:39): upper-case ACGT only
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); const bool on = std::getenv("VGAMD_TIMING") != nullptr;
    void operator()(const char* what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pack_windows] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t;
    }
};

}  // namespace

extern "C" {

int vgk_graph_create(vgk_ctx* ctx, const vgk_graph* graph, vgk_dgraph** out) try {
    if (!ctx || !graph || !out) return VGK_EINVAL;
    *out = nullptr;
    const vgk_graph& g = *graph;
    if (g.n_nodes == 0 || !g.node_len || !g.seq || !g.pred_off) return VGK_EINVAL;
    const uint32_t n = g.n_nodes;
    if (g.pred_off[n] > g.pred_off[0] && !g.pred_idx) return VGK_EINVAL;
    // columns, flags and slots of the WHOLE graph, as vgk_gssw_pack computes them per problem (node_flags in vgk_api.cpp):
    // slow[v] = the node's first column is seeded from its predecessors' saved last columns (anything but the plain chain link
    // to v - 1); store[v] = some successor seeds from v's last column
    std::vector<uint32_t> col((size_t)n + 1), slot((size_t)n + 1);
    std::vector<uint8_t> slow(n, 0), store(n, 0);
    uint64_t cols = 0;
    for (uint32_t v = 0; v < n; ++v) {
        const uint32_t pb = g.pred_off[v], pe = g.pred_off[v + 1];
        if (pe < pb || g.node_len[v] == 0) return VGK_EINVAL;
        if (g.node_len[v] > 65535u) return VGK_ETOOBIG;
        for (uint32_t k = pb; k < pe; ++k) if (g.pred_idx[k] >= v) return VGK_EINVAL;       // not topological
        const bool chain = (pe - pb == 1) && g.pred_idx[pb] + 1 == v;
        slow[v] = (v > 0 && !chain) ? 1 : 0;
        if (slow[v]) for (uint32_t k = pb; k < pe; ++k) store[g.pred_idx[k]] = 1;
        col[v] = (uint32_t)cols; cols += g.node_len[v];
        if (cols >= (1ull << 32) - 16) return VGK_ETOOBIG;
    }
    col[n] = (uint32_t)cols;
    std::vector<uint32_t> far_prefix((size_t)n + 1, 0);
    for (uint32_t v = 0; v < n; ++v) {
        bool far = false;
        for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) far |= col[v] - col[g.pred_idx[k] + 1] > TB_JUMP;
        far_prefix[v + 1] = far_prefix[v] + (far ? 1u : 0u);
    }
    uint32_t s = 0;
    for (uint32_t v = 0; v < n; ++v) { slot[v] = s; s += store[v]; }
    slot[n] = s;
    std::vector<uint8_t> info((size_t)cols + 8, (uint8_t)CI_INVALID);
    parallel_for(n, [&](uint32_t v, unsigned) {
        const uint32_t len = g.node_len[v]; const char* sq = g.seq + col[v]; uint8_t* o = info.data() + col[v];
        for (uint32_t k = 0; k < len; ++k) o[k] = (uint8_t)ref_code(sq[k]);
        o[0] |= CI_NODE_START | (slow[v] ? CI_SEED_SLOW : 0);
        if (store[v]) o[len - 1] |= CI_STORE_END;
    });
    std::unique_ptr<vgk_dgraph> dg(new (std::nothrow) vgk_dgraph());
    if (!dg) return VGK_ENOMEM;
    dg->ctx = ctx; dg->far_prefix.swap(far_prefix);
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto put = [&](const void* src, size_t bytes, const void*& dst) -> int {
        void* p = be->alloc(bytes ? bytes : 16);
        if (!p) return VGK_ENOMEM;
        dg->dev.push_back(p); dg->dev_bytes += bytes;
        dst = p;
        return bytes ? be->upload(p, src, bytes) : VGK_OK;
    };
    // successors (ascending per node: the order edges come in when a caller builds the subgraph node by node), for leftward extension windows
    std::vector<uint32_t> so((size_t)n + 1, 0), si(g.pred_off[n] - g.pred_off[0]);
    for (uint32_t v = 0; v < n; ++v) for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) ++so[g.pred_idx[k] + 1];
    for (uint32_t v = 0; v < n; ++v) so[v + 1] += so[v];
    { std::vector<uint32_t> at(so.begin(), so.end() - 1);
      for (uint32_t v = 0; v < n; ++v) for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) si[at[g.pred_idx[k]]++] = v; }
    const void *d_col = nullptr, *d_info = nullptr, *d_po = nullptr, *d_pi = nullptr, *d_slot = nullptr, *d_so = nullptr, *d_si = nullptr;
    const size_t n_edges = g.pred_off[n] - g.pred_off[0];
    std::vector<uint32_t> po((size_t)n + 1);
    for (uint32_t v = 0; v <= n; ++v) po[v] = g.pred_off[v] - g.pred_off[0];
    int rc = put(col.data(), col.size() * 4, d_col);
    if (!rc) rc = put(info.data(), info.size(), d_info);
    if (!rc) rc = put(po.data(), po.size() * 4, d_po);
    if (!rc) rc = put(n_edges ? g.pred_idx + g.pred_off[0] : nullptr, n_edges * 4, d_pi);
    if (!rc) rc = put(slot.data(), slot.size() * 4, d_slot);
    if (!rc) rc = put(so.data(), so.size() * 4, d_so);
    if (!rc) rc = put(si.empty() ? nullptr : si.data(), si.size() * 4, d_si);
    if (!rc) rc = be->sync();         // the host vectors go away
    if (rc) { for (void* p : dg->dev) be->release(p); return rc; }
    dg->g.col = (const uint32_t*)d_col; dg->g.info = (const uint8_t*)d_info; dg->g.pred_off = (const uint32_t*)d_po;
    dg->g.pred_idx = (const uint32_t*)d_pi; dg->g.slot = (const uint32_t*)d_slot; dg->g.n_nodes = n; dg->g.n_cols = (uint32_t)cols;
    dg->g.succ_off = (const uint32_t*)d_so; dg->g.succ_idx = (const uint32_t*)d_si;
    *out = dg.release();
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

void vgk_graph_destroy(vgk_dgraph* dg) {
    if (!dg) return;
    {
        std::lock_guard<std::mutex> lk(dg->ctx->mu);
        dg->ctx->be->sync(); dg->ctx->be->sync_side();
        if (dg->dev_size.size() == dg->dev.size()) for (size_t k = 0; k < dg->dev.size(); ++k) dg->ctx->dev_give(dg->dev[k], dg->dev_size[k]);
        else for (void* p : dg->dev) dg->ctx->be->release(p);
    }
    delete dg;
}

// `on_device`: reads and problems are device arrays already (vgk_tail_stage builds them there) and the caller has waited for the
// kernels that wrote them; nothing is staged.
// `extensions` (nullable, host array, never with on_device): the problems are EXTENSION windows (gssw_pack_device.hpp WinExt; vgk_gssw_pack_extensions)
int vgk_pack_windows_impl(vgk_ctx* ctx, const vgk_dgraph* dg, const char* reads, size_t reads_bytes,
                          const vgk_window_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out, bool on_device, uint32_t forced_k,
                          const vgk::WinExt* extensions) try {
    if (!ctx || !dg || dg->ctx != ctx || !out || (!problems && n) || (!reads && reads_bytes) || (extensions && on_device)) return VGK_EINVAL;
    *out = nullptr;
    if (ctx->has_qa) return VGK_EUNSUPPORTED;
    Lap lap;
    std::unique_ptr<vgk_batch> hb(new (std::nothrow) vgk_batch());
    if (!hb) return VGK_ENOMEM;
    vgk_batch* b = hb.get();
    b->ctx = ctx; b->n = n;
    Backend* be = ctx->be.get();
    GsswParams& P = b->P;
    P = GsswParams{};
    struct StagingLease {
        vgk_ctx* ctx; std::unique_ptr<vgk_ctx::Staging> s;
        ~StagingLease() { if (s) ctx->staging_release(std::move(s)); }
    } lease{ctx, ctx->staging_acquire()};
    // device memory that only the packing itself needs goes back to the pool when this function returns
    std::vector<vgk_ctx::Pooled> temp;
    struct TempGuard {
        vgk_ctx* ctx; std::vector<vgk_ctx::Pooled>& v;
        ~TempGuard() { if (v.empty()) return; ctx->be->sync_side(); std::lock_guard<std::mutex> lk(ctx->mu); for (auto& q : v) ctx->dev_give(q.p, q.bytes); }
    } temp_guard{ctx, temp};
    auto fail = [&](int code) { vgk_batch* t = hb.release(); ctx->be->sync_side(); vgk_batch_free(t); return code; };
    auto take_temp = [&](uint64_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) temp.push_back({p, got}); return p; };
    auto take_keep = [&](uint64_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) { b->dev.push_back({p, got}); b->dev_bytes += bytes; } return p; };

    // does every window keep its tracebacks near a diagonal?  (host-side: the windows are the caller's array; a window's first node has
    // lost its predecessors, the others must not have a far one)
    bool near_chain = false;
    if (!on_device && !extensions && !dg->far_prefix.empty()) {
        std::atomic<bool> far{false};
        const std::vector<uint32_t>& fp = dg->far_prefix; const uint32_t gn = dg->g.n_nodes;
        parallel_chunks(n, [&](uint32_t lo, uint32_t hi, uint32_t) {
            for (uint32_t i = lo; i < hi; ++i) {
                const vgk_window_problem& w = problems[i];
                if (w.n_nodes < 2 || w.first_node >= gn || w.n_nodes > gn - w.first_node) continue;      // (malformed windows are reported by the packer)
                if (fp[w.first_node + w.n_nodes] != fp[w.first_node + 1]) { far.store(true, std::memory_order_relaxed); break; }
            }
        });
        near_chain = !far.load();
    }
    const uint32_t n1 = n + 1;
    const uint32_t waves_cap = n / 2 + WIN_BUCKETS + 1;
    WinParams W{};
    W.g = dg->g; W.n = n; W.raw_bytes = reads_bytes; W.ops_per_problem = ops_per_problem;
    W.forced_k = forced_k;                        // (vgk_tail_stage: one rows-per-lane class for a batch too small to fill the device three times over)
    if (const char* e = std::getenv("VGAMD_ROWS_PER_LANE")) W.forced_k = (uint32_t)std::atoi(e);
    W.max_score = ctx->max_score; W.max_bonus = ctx->max_bonus; W.scale = ctx->scale; W.bonus = ctx->sc.full_length_bonus;
    W.n_waves_cap = waves_cap;
    void* tmp = nullptr; size_t tmp_bytes = be->win_tmp_bytes(n, waves_cap);
    int rc;
    rc = VGK_OK;
    uint64_t ext_nodes_total = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        W.problems = on_device ? problems : (const vgk_window_problem*)take_temp((uint64_t)n * sizeof(vgk_window_problem));
        W.raw_reads = on_device ? (const uint8_t*)reads : (const uint8_t*)take_temp(reads_bytes + 8);
        W.sizes = (uint32_t*)take_temp((uint64_t)WIN_COLS * n1 * 4); W.offs = (uint32_t*)take_temp((uint64_t)WIN_COLS * n1 * 4);
        W.key = (uint32_t*)take_temp((uint64_t)n * 4); W.idx = (uint32_t*)take_temp((uint64_t)n * 4);
        W.key_sorted = (uint32_t*)take_temp((uint64_t)n * 4); W.idx_sorted = (uint32_t*)take_temp((uint64_t)n * 4);
        W.totals = (WinTotals*)take_temp(sizeof(WinTotals));
        W.bucket_first = (uint32_t*)take_temp(sizeof(uint32_t) * WIN_BUCKETS); W.buckets = (WinBucket*)take_temp(sizeof(WinBucket) * WIN_BUCKETS);
        W.wave_tb = (unsigned long long*)take_temp(((uint64_t)waves_cap + 1) * 8);
        tmp = take_temp(tmp_bytes);
        b->lane = (std::getenv("VGAMD_ONE_STREAM") ? 0 : (int)(ctx->batch_seq++ & 1u));
        W.probs = (ProbDesc*)take_keep((uint64_t)std::max<uint32_t>(n, 1u) * sizeof(ProbDesc));
        if (extensions) {
            ext_nodes_total = n ? (uint64_t)extensions[n - 1].kept_off + problems[n - 1].n_nodes : 0;
            W.ext = (const WinExt*)take_temp((uint64_t)std::max<uint32_t>(n, 1u) * sizeof(WinExt));
            W.kept = (WinKept*)take_temp((ext_nodes_total + 1) * sizeof(WinKept)); W.ext_count = (uint32_t*)take_temp((uint64_t)std::max<uint32_t>(n, 1u) * 4);
            W.kept_node = (uint32_t*)take_temp((ext_nodes_total + 1) * 4);
            if (!W.ext || !W.kept || !W.ext_count || !W.kept_node) rc = VGK_ENOMEM;
        }
        if (!W.problems || !W.raw_reads || !W.sizes || !W.offs || !W.key || !W.idx || !W.key_sorted || !W.idx_sorted || !W.totals ||
            !W.bucket_first || !W.buckets || !W.wave_tb || !tmp || !W.probs) rc = VGK_ENOMEM;
    }
    if (rc) return fail(rc);          // (outside the block: fail() takes the context lock itself)
    lap("device buffers");
    // the two flat inputs: host threads copy them into page-locked staging, slice by slice, and each slice starts its way to HBM
    // as soon as it is there (the caller's buffers are pageable: a direct copy would be staged by the runtime on one thread)
    {
        WinTotals zero{}; zero.first_bad = ~0ull;
        WinTotals* tz = (WinTotals*)lease.s->get(2, sizeof(WinTotals));
        if (!tz) return fail(VGK_ENOMEM);
        *tz = zero;
        if ((rc = be->upload_side(W.totals, tz, sizeof(WinTotals)))) return fail(rc);
    }
    auto staged_upload = [&](int slot, void* dst, const void* src, uint64_t bytes) -> int {
        if (!bytes) return VGK_OK;
        uint8_t* st = (uint8_t*)lease.s->get(slot, bytes);
        if (!st) return VGK_ENOMEM;
        const uint64_t SLICE = 16ull << 20, PIECE = 256ull << 10;
        for (uint64_t at = 0; at < bytes; at += SLICE) {
            const uint64_t len = std::min(SLICE, bytes - at);
            parallel_tasks((uint32_t)((len + PIECE - 1) / PIECE), [&](uint32_t c) {
                const uint64_t o = at + (uint64_t)c * PIECE;
                std::memcpy(st + o, (const uint8_t*)src + o, (size_t)std::min(PIECE, at + len - o));
            });
            const int e = be->upload_side((uint8_t*)dst + at, st + at, len);
            if (e) return e;
        }
        return VGK_OK;
    };
    if (!on_device && (rc = staged_upload(1, (void*)W.problems, problems, (uint64_t)n * sizeof(vgk_window_problem)))) return fail(rc);
    if (extensions && (rc = staged_upload(3, (void*)W.ext, extensions, (uint64_t)n * sizeof(WinExt)))) return fail(rc);
    lap("problems staged");
    if ((rc = be->win_stage1(W, tmp, tmp_bytes))) return fail(rc);         // needs the problems only: runs while the reads travel
    if (!on_device && (rc = staged_upload(0, (void*)W.raw_reads, reads, reads_bytes))) return fail(rc);
    lap("reads staged");
    WinTotals T;
    if ((rc = be->download_side(&T, W.totals, sizeof T))) return fail(rc);
    lap("stage 1 (sizes)");
    if (T.first_bad != ~0ull) return fail(-(int)(T.first_bad & 0xffu));
    for (uint32_t k = 0; k < WIN_COLS; ++k) if (T.tot[k] >= (1ull << 32)) return fail(VGK_ETOOBIG);
    if (extensions && n) {
        // which window node every node of an extension problem is: results and ops come back in those terms (vgk_gssw_fetch translates)
        // (page-locked staging: 4 bytes per window node and per problem come down at the link's rate; the batch keeps them in the windows' own layout)
        uint32_t* st_nodes = (uint32_t*)lease.s->get(4, (ext_nodes_total + 1) * 4); uint32_t* st_count = (uint32_t*)lease.s->get(6, (uint64_t)n * 4);
        if (!st_nodes || !st_count) return fail(VGK_ENOMEM);
        if ((rc = be->download_side(st_count, W.ext_count, (size_t)n * 4))) return fail(rc);
        if ((rc = be->download_side(st_nodes, W.kept_node, (size_t)ext_nodes_total * 4))) return fail(rc);
        b->ext_count.assign(st_count, st_count + n); b->ext_nodes.assign(st_nodes, st_nodes + ext_nodes_total);
        b->ext_off.resize((size_t)n + 1);
        for (uint32_t i = 0; i < n; ++i) b->ext_off[i] = extensions[i].kept_off;
        b->ext_off[n] = (uint32_t)ext_nodes_total;
        lap("extension nodes");
    }
    b->want_tb = T.want_tb != 0; b->cells = T.cells; b->tb_cells = T.tb_cells; b->in_bytes = T.in_bytes;
    W.want_tb = b->want_tb ? 1 : 0;
    uint8_t* colinfo; uint8_t* rd; NodeRec* nodes; uint32_t* preds; WaveDesc* waves; uint32_t* order;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        colinfo = (uint8_t*)take_keep(T.tot[WS_COLS] + 8); rd = (uint8_t*)take_keep(T.tot[WS_READS] + 8);
        nodes = (NodeRec*)take_keep((T.tot[WS_NODES] + 1) * sizeof(NodeRec)); preds = (uint32_t*)take_keep((T.tot[WS_PREDS] + 1) * 4);
        waves = (WaveDesc*)take_keep((uint64_t)waves_cap * sizeof(WaveDesc)); order = (uint32_t*)take_keep(((uint64_t)waves_cap * 4 + 4) * 4);      // (order: the batch's pairs, and as many again for the wavefronts a speculative fill fills twice)
        rc = dev_alloc(b, (size_t)T.tot[WS_SCRATCH] + 16, P.scratch);
        if (!rc) rc = dev_alloc(b, (size_t)tb_best_entries(n), P.best);
        if (!rc) rc = dev_alloc(b, (size_t)n + 1, P.results);
        if (!rc) rc = dev_alloc(b, (size_t)T.tot[WS_OPS] + 1, P.ops);
        if (!rc && (!colinfo || !rd || !nodes || !preds || !waves || !order)) rc = VGK_ENOMEM;
    }
    if (rc) return fail(rc);
    W.colinfo = colinfo; W.reads = rd; W.nodes = nodes; W.preds = preds; W.waves = waves; W.order = order;
    if ((rc = be->fill_side(colinfo + T.tot[WS_COLS], (int)CI_INVALID, 8))) return fail(rc);      // leaders prefetch one word ahead
    if ((rc = be->win_stage2(W, tmp, tmp_bytes))) return fail(rc);
    if ((rc = be->download_side(&T, W.totals, sizeof T))) return fail(rc);
    lap("stage 2 (order, waves, arenas)");
    if (T.n_waves > waves_cap || T.n_launches > 4) return fail(VGK_EINVAL);
    // The speculative fill (GsswParams::spec_fill; vgk_api.cpp has the same rule for batches packed on the host): mostly local alignments
    // with tracebacks, one launch, one lanes-per-pair geometry, stored codes.  The traceback arena then serves only the wavefronts that
    // are filled a second time: as many as there are wavefronts now at most, each as large as the longest one.
    uint32_t local_tb = 0;
    if (!on_device) for (uint32_t i = 0; i < n; ++i) local_tb += (problems[i].flags & (15u | VGK_GSSW_TRACEBACK)) == (uint32_t)(VGK_GSSW_LOCAL | VGK_GSSW_TRACEBACK) ? 1u : 0u;
    const bool walk2 = !on_device && n >= 1024 && 2ull * local_tb >= n && !std::getenv("VGAMD_WALK_ONE_PASS");
    bool spec = walk2 && b->want_tb && T.n_launches == 1 && T.n_buckets == 1 && T.first_G && default_tb_mode(0, near_chain) == TB_CODES && 2ull * T.n_waves <= waves_cap &&
                !std::getenv("VGAMD_NO_SPEC_FILL");
    uint64_t refill_slot = 0, tb_dwords = T.tb_dwords;
    if (spec) {
        refill_slot = tb_wave_dwords(T.max_steps, T.launch_K[0]);
        if ((uint64_t)T.n_waves * refill_slot > T.tb_dwords + T.tb_dwords / 2) spec = false;      // windows of very different widths
        else tb_dwords = (uint64_t)T.n_waves * refill_slot;
    }
    P.spec_fill = 0; P.wave_limit = nullptr; P.refill_count = nullptr; P.refill_wave0 = P.refill_pair0 = P.refill_G = P.refill_K = 0; P.refill_slot = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        rc = dev_alloc(b, (size_t)tb_dwords + 4, P.tb);
        if (!rc && spec) rc = dev_alloc(b, (size_t)4, P.refill_count);
    }
    if (rc) return fail(rc);
    if (spec) { P.spec_fill = 1; P.refill_wave0 = T.n_waves; P.refill_pair0 = T.n_pairs; P.refill_G = T.first_G; P.refill_K = T.launch_K[0]; P.refill_slot = refill_slot; }
    for (uint32_t k = 0; k < T.n_launches; ++k) b->launches.push_back(FillLaunch{T.launch_K[k], T.launch_begin[k], T.launch_count[k]});
    P.probs = W.probs; P.colinfo = colinfo; P.reads = rd; P.prof = nullptr; P.nodes = nodes; P.preds = preds; P.waves = waves; P.order = order;
    P.wave_begin = 0; P.wave_count = 0; P.K = 0;
    P.n_problems = n; P.n_pairs = T.n_pairs; P.n_waves = T.n_waves;
    const uint32_t S = ctx->scale;
    for (int q = 0; q < 6; ++q) P.prof4[q] = ctx->prof4[q] * S;
    P.bias = ctx->bias * S; P.go = ctx->sc.gap_open * S; P.ge = ctx->sc.gap_extend * S; P.bonus = ctx->sc.full_length_bonus * (int32_t)S;
    P.scale = S; P.xoff = XOFF * S;
    P.want_tb = b->want_tb ? 1 : 0;
    P.fused = 0; P.dbg = std::getenv("VGAMD_TB_DBG") ? std::atoi(std::getenv("VGAMD_TB_DBG")) : 0; P.tb_mode = default_tb_mode(0, near_chain);
    P.walk_passes = walk2 ? 2 : 1;                                    // (windows that were made on the device are tails: X-drop)
    if (P.walk_passes != 2 || P.tb_mode != TB_CODES) P.spec_fill = 0;
    P.key3 = 0;
    if (P.spec_fill) {                                                  // (speculation only with the problems at hand on the host: walk2)
        uint32_t longest = 0; bool xdrop = false;
        for (uint32_t i = 0; i < n; ++i) { longest = std::max(longest, problems[i].read_len); xdrop = xdrop || (problems[i].flags & 15u) == (uint32_t)VGK_XDROP_PINNED; }
        P.key3 = gssw_key3_ok(S, ctx->has_qa, xdrop, longest, (uint32_t)std::max(0, ctx->max_score), (uint32_t)std::max(0, ctx->max_bonus)) && !std::getenv("VGAMD_NO_KEY3") ? 1u : 0u;
    }
    std::memcpy(P.matrix, ctx->sc.matrix, 25);
    b->ops_total = T.tot[WS_OPS]; b->wave_steps = T.wave_steps;
    lap("done");
    *out = hb.release();
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_gssw_pack_windows(vgk_ctx* ctx, const vgk_dgraph* dg, const char* reads, size_t reads_bytes,
                          const vgk_window_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) try {
    return vgk_pack_windows_impl(ctx, dg, reads, reads_bytes, problems, n, ops_per_problem, out, false, 0, nullptr);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_gssw_pack_extensions(vgk_ctx* ctx, const vgk_dgraph* dg, const char* reads, size_t reads_bytes,
                             const vgk_extension_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) try {
    if (!ctx || !dg || dg->ctx != ctx || !out || (!problems && n) || (!reads && reads_bytes)) return VGK_EINVAL;
    if (!dg->g.succ_off) return VGK_EUNSUPPORTED;                       // (a graph made on the device: tail forests are not extended from inside)
    // the two halves the packer reads: the window as for vgk_gssw_pack_windows, the start beside it; kept_off = a prefix sum over the windows' nodes
    std::vector<vgk_window_problem> win(n); std::vector<WinExt> ext(n);
    uint64_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_extension_problem& p = problems[i];
        vgk_window_problem& w = win[i]; WinExt& x = ext[i];
        w.read_off = p.read_off; w.read_len = p.read_len; w.flags = p.flags; w.first_node = p.first_node; w.n_nodes = p.n_nodes; w.max_gap_length = p.max_gap_length; w.reserved = 0;
        x.start_node = p.start_node; x.start_offset = p.start_offset; x.query_offset = p.query_offset; x.leftward = p.leftward ? 1u : 0u; x.kept_off = (uint32_t)at; x.pad = 0;
        at += p.n_nodes;
        if (at >= (1ull << 32) - 16) return VGK_ETOOBIG;
    }
    return vgk_pack_windows_impl(ctx, dg, reads, reads_bytes, win.data(), n, ops_per_problem, out, false, 0, ext.data());
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

}  // extern "C"
