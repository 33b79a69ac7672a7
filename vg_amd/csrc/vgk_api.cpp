// vgk_api.cpp — the C ABI of include/vgk.h: validation, host-side packing of
// (read, DAG) problems into flat HBM arenas, kernel orchestration, result fetch.
//
// Packing mirrors what GSSWAligner::create_gssw_graph does per call on the CPU
// (reference: src/aligner.cpp:30-85) — one malloc'd gssw_node per graph node and an
// unordered_map — but emits flat arrays: a per-column info byte stream (base code +
// node-boundary flags), a node table, a predecessor CSR and a per-read descriptor.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include "backend.hpp"
#include "ctx.hpp"
#include "host_parallel.hpp"

#include "batch.hpp"
#include "gssw_wide.hpp"

extern "C" {

int vgk_abi_version(void) { return VGK_ABI_VERSION; }

const char* vgk_strerror(int code) {
    switch (code) {
        case VGK_OK: return "ok";
        case VGK_EINVAL: return "invalid argument";
        case VGK_ENODEV: return "no usable HIP device";
        case VGK_ENOMEM: return "out of memory";
        case VGK_ETOOLONG: return "read too long for the engine";
        case VGK_EOVERFLOW: return "score overflow";
        case VGK_EOPS: return "cigar buffer too small";
        case VGK_ETOOBIG: return "band matrices too big";
        case VGK_ENOBAND: return "no alignment in band";
        case VGK_EUNSUPPORTED: return "scoring parameters outside the kernels' range";
        default: return "unknown error";
    }
}

static int create_ctx(int device, const vgk_scoring* scoring, const vgk_qual_adj* qa, vgk_ctx** out) {
    if (!scoring || !out) return VGK_EINVAL;
    *out = nullptr;
    int mn = 0, mx = 0, mb = scoring->full_length_bonus;
    for (int i = 0; i < 25; ++i) { mn = std::min<int>(mn, scoring->matrix[i]); mx = std::max<int>(mx, scoring->matrix[i]); }
    if (qa) {
        if (!qa->matrix || !qa->bonuses) return VGK_EINVAL;
        for (int i = 0; i < 256 * 25; ++i) { mn = std::min<int>(mn, qa->matrix[i]); mx = std::max<int>(mx, qa->matrix[i]); }
        for (int i = 0; i < 256; ++i) { if (qa->bonuses[i] < 0) return VGK_EUNSUPPORTED; mb = std::max<int>(mb, qa->bonuses[i]); }
    }
    const int bias = std::max(1, -mn);
    // profile bytes hold score + bias + (up to two) bonuses; see gssw_device.hpp
    if (scoring->full_length_bonus < 0 || mx + bias + 2 * mb > 255) return VGK_EUNSUPPORTED;
    std::string err;
    Backend* be = make_backend(device, err);
    if (!be) return VGK_ENODEV;
    vgk_ctx* c = new (std::nothrow) vgk_ctx();
    if (!c) { delete be; return VGK_ENOMEM; }
    c->sc = *scoring; c->be.reset(be); c->bias = (uint32_t)bias; c->max_score = mx; c->max_bonus = mb;
    c->scale = ((mx + bias + 2 * mb) * 8 <= 255) ? 8u : 1u;
    if (const char* e = std::getenv("VGAMD_SCORE_SCALE")) c->scale = (std::atoi(e) == 8 && c->scale == 8) ? 8u : 1u;
    if (qa) { c->has_qa = true; c->qmat.assign(qa->matrix, qa->matrix + 256 * 25); c->qbon.assign(qa->bonuses, qa->bonuses + 256); }
    for (int q = 0; q < 5; ++q) {
        uint32_t w = 0;
        for (int r = 0; r < 4; ++r) w |= (uint32_t)(scoring->matrix[5 * r + q] + bias) << (8 * r);
        c->prof4[q] = w;
    }
    if (const char* e = std::getenv("VGAMD_SPEC_POLICY")) { const int m = std::atoi(e); if (m >= 0 && m <= 2) c->spec.mode = m; }      // vgk_set_speculation
    if (const char* e = std::getenv("VGAMD_SPEC_MISS_MAX")) { const double m = std::atof(e); if (m > 0.0 && m <= 1.0) c->spec.miss_max = m; }
    if (const char* e = std::getenv("VGAMD_SPEC_PROBE_EVERY")) { const int m = std::atoi(e); if (m >= 1 && m <= 1024) c->spec.probe_every = c->spec.interval = (uint32_t)m; }
    c->prof4[5] = 0;     // X-drop row 0 ("nothing consumed"): no diagonal move can enter it
    *out = c;
    return VGK_OK;
}

int vgk_create(int device, const vgk_scoring* scoring, vgk_ctx** out) { return create_ctx(device, scoring, nullptr, out); }
int vgk_create_qual_adj(int device, const vgk_scoring* scoring, const vgk_qual_adj* qual_adj, vgk_ctx** out) try {
    if (!qual_adj) return VGK_EINVAL;
    return create_ctx(device, scoring, qual_adj, out);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

void vgk_destroy(vgk_ctx* ctx) { delete ctx; }

int vgk_device_info(vgk_ctx* ctx, char* name_out, size_t name_cap, int* cus, size_t* hbm) try {
    if (!ctx) return VGK_EINVAL;
    if (name_out && name_cap) { std::strncpy(name_out, ctx->be->name(), name_cap - 1); name_out[name_cap - 1] = 0; }
    if (cus) *cus = ctx->be->compute_units();
    if (hbm) *hbm = ctx->be->memory_bytes();
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_host_register(vgk_ctx* ctx, const void* ptr, size_t bytes) try {
    if (!ctx || !ptr || !bytes) return VGK_EINVAL;
    return ctx->be->host_register(ptr, bytes);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
int vgk_host_unregister(vgk_ctx* ctx, const void* ptr) try {
    if (!ctx || !ptr) return VGK_EINVAL;
    ctx->be->sync(); ctx->be->sync_side();                  // no copy out of the range may still be in flight
    return ctx->be->host_unregister(ptr);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

void vgk_batch_free(vgk_batch* b) {
    if (!b) return;
    // only THIS batch's work has to be over before its arenas go back to the pool: its kernels (the event behind them) and
    // whatever a fetch queued on the fetch stream — not the kernels of a later batch that may already be queued
    if (b->done) b->ctx->be->event_wait(b->done); else b->ctx->be->sync();
    b->ctx->be->sync_fetch();
    {
        std::lock_guard<std::mutex> lk(b->ctx->mu);
        for (const vgk_ctx::Pooled& q : b->dev) b->ctx->dev_give(q.p, q.bytes);
    }
    delete b;
}

int vgk_gssw_pack(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) try {
    if (!ctx || !out || (!problems && n)) return VGK_EINVAL;
    *out = nullptr;
    std::unique_ptr<vgk_batch> hb(new (std::nothrow) vgk_batch());
    if (!hb) return VGK_ENOMEM;
    vgk_batch* b = hb.get();
    b->ctx = ctx; b->n = n;

    uint32_t forced = 0;
    if (const char* e = std::getenv("VGAMD_ROWS_PER_LANE")) forced = (uint32_t)std::atoi(e);
    // lane geometry (rows per lane K, lanes per pair G) of a read: gssw_pack_device.hpp, shared with the device-side packer
    auto geometry = [&](uint32_t rows, uint32_t& K, uint32_t& G) { lane_geometry(rows, forced, K, G); };

    // Three passes over the problems, the first and the last on host threads: (1) validate and size every problem, (2) prefix sums
    // place it in the shared arenas, (3) encode it at its offsets.
    auto T0 = std::chrono::steady_clock::now(); auto lap = [&](const char* w) { if (std::getenv("VGAMD_TIMING")) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[pack] %s %.1f ms\n", w, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    // per-problem descriptors and sizes live in uninitialised storage: the packing threads initialise (and fault in) their own parts
    struct StagingLease {
        vgk_ctx* ctx; std::unique_ptr<vgk_ctx::Staging> s;
        ~StagingLease() { if (s) ctx->staging_release(std::move(s)); }
    } lease{ctx, ctx->staging_acquire()};
    b->probs = (ProbDesc*)ctx->host_take((uint64_t)std::max<uint32_t>(n, 1u) * sizeof(ProbDesc), b->probs_bytes);
    ProbDesc* probs = b->probs;
    struct Sizes { uint32_t reads = 0, prof = 0, cols = 0, nodes = 0, preds = 0, pred_at = 0; int status = VGK_OK; bool want_tb = false, malformed = false, far = false; };
    Sizes* sizes = (Sizes*)lease.s->get(6, (uint64_t)std::max<uint32_t>(n, 1u) * sizeof(Sizes));
    if (!probs || !sizes) return VGK_ENOMEM;
    std::vector<uint32_t> thread_maxL(MAX_THREADS, 1u);
    struct Flags { std::vector<uint8_t> store, slow; std::vector<uint32_t> col_end; };
    std::vector<Flags> thread_flags(MAX_THREADS);
    // store[v]: the node's last column is saved for a successor / the pinned end; slow[v]: its first column is seeded from scratch
    auto node_flags = [&](const vgk_gssw_problem& p, bool xdrop, uint32_t mode, Flags& f) -> int {
        const vgk_graph& g = p.graph;
        f.store.assign(g.n_nodes, 0); f.slow.assign(g.n_nodes, 0);
        for (uint32_t v = 0; v < g.n_nodes; ++v) {
            const uint32_t pb = g.pred_off[v], pe = g.pred_off[v + 1];
            if (pe < pb || g.node_len[v] == 0 || (pe > pb && !g.pred_idx)) return VGK_EINVAL;
            if (g.node_len[v] > 65535u) return VGK_ETOOBIG;      // vgk_op.len is 16 bits: a run inside one node must fit (vg chops nodes to <= 1024 bp)
            for (uint32_t k = pb; k < pe; ++k) if (g.pred_idx[k] >= v) return VGK_EINVAL;   // not topological
            const bool chain = (pe - pb == 1) && g.pred_idx[pb] + 1 == v;
            f.slow[v] = ((v > 0 || xdrop) && !chain) ? 1 : 0;      // X-drop: node 0 starts from the root column
            if (f.slow[v]) for (uint32_t k = pb; k < pe; ++k) f.store[g.pred_idx[k]] = 1;
            if (mode == VGK_GSSW_PINNED && p.pinning[v]) f.store[v] = 1;
        }
        return VGK_OK;
    };
    lap("setup");
    parallel_for(n, [&](uint32_t i, unsigned t) {
        const vgk_gssw_problem& p = problems[i];
        const vgk_graph& g = p.graph;
        ProbDesc& d = probs[i]; Sizes& z = sizes[i];
        d = ProbDesc{}; z = Sizes();
        if (p.read_len == 0 || !p.read || g.n_nodes == 0) { z.status = VGK_EINVAL; z.malformed = true; return; }
        const uint32_t mode = p.flags & 15u;
        thread_maxL[t] = std::max(thread_maxL[t], p.read_len + (mode == VGK_XDROP_PINNED ? 1u : 0u));
        if (mode != VGK_GSSW_LOCAL && mode != VGK_GSSW_PINNED && mode != VGK_XDROP_PINNED) { z.status = VGK_EINVAL; return; }
        const bool xdrop = mode == VGK_XDROP_PINNED;
        // offset arithmetic of the X-drop mode: every reachable gain must stay below XOFF
        if (xdrop && (int64_t)p.read_len * std::max(ctx->max_score, 0) + ctx->max_bonus >= (int64_t)XOFF) { z.status = VGK_EUNSUPPORTED; return; }
        if ((ctx->has_qa && !p.qual) || (mode == VGK_GSSW_PINNED && !p.pinning) || !g.node_len || !g.pred_off || !g.seq) { z.status = VGK_EINVAL; return; }
        z.want_tb = (p.flags & VGK_GSSW_TRACEBACK) != 0;
        d.flags = p.flags; d.L = p.read_len + (xdrop ? 1u : 0u); d.n_nodes = g.n_nodes;
        d.max_gap = xdrop ? ((std::max<uint32_t>(p.max_gap_length, 1u) + 7u) & ~7u) : 0u;
        {   // which full-length bonuses this problem grants, and their values (src/aligner.cpp:401-402, 942-952, 1164-1167)
            const uint32_t S = ctx->scale;
            const int first_b = ctx->has_qa ? ctx->qbon[p.qual[0]] : ctx->sc.full_length_bonus;
            const int last_b = ctx->has_qa ? ctx->qbon[p.qual[p.read_len - 1]] : ctx->sc.full_length_bonus;
            d.bonus_start = xdrop ? 0u : (uint32_t)first_b * S;
            d.bonus_end = (mode == VGK_GSSW_PINNED) ? 0u : (uint32_t)last_b * S;
            d.prof_off = 0xffffffffu; d.pad = 0;
        }
        Flags& f = thread_flags[t];
        if ((z.status = node_flags(p, xdrop, mode, f)) != VGK_OK) return;
        uint64_t col = 0; uint32_t slots = 0;
        f.col_end.resize(g.n_nodes);
        for (uint32_t v = 0; v < g.n_nodes; ++v) {
            // a predecessor that ends more than TB_JUMP columns before this node starts: tracebacks over that edge leave the band (batch.hpp)
            for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) z.far |= col - f.col_end[g.pred_idx[k]] > TB_JUMP;
            col += g.node_len[v]; slots += f.store[v]; f.col_end[v] = (uint32_t)col;
        }
        if (col >= (1u << 20)) { z.status = VGK_ETOOBIG; return; }
        d.R = (uint32_t)col; d.n_slots = slots;
        uint32_t pK, pG; geometry(d.L, pK, pG);
        d.geom = pK | (pG << 8); d.Lpad = pG * pK; d.wave = 0; d.lane0 = 0;
        d.ops_cap = ops_per_problem ? ops_per_problem : (p.read_len + d.R + 2);
        if (!(p.flags & VGK_GSSW_TRACEBACK)) d.ops_cap = 0;
        z.reads = d.L; z.prof = ctx->has_qa ? d.L : 0; z.cols = d.R; z.nodes = g.n_nodes; z.preds = g.pred_off[g.n_nodes] - g.pred_off[0];
    });
    lap("pass1");
    // whole-batch checks first, in the order a serial scan would report them: a malformed problem, then the longest read
    {
        uint32_t maxL = 1;
        for (uint32_t m : thread_maxL) maxL = std::max(maxL, m);
        // (malformed problems are found below, chunk by chunk, before anything else is reported)
        std::atomic<bool> malformed{false};
        parallel_chunks(n, [&](uint32_t lo, uint32_t hi, uint32_t) { for (uint32_t i = lo; i < hi; ++i) if (sizes[i].malformed) { malformed.store(true, std::memory_order_relaxed); break; } });
        if (malformed.load()) return VGK_EINVAL;
        if (maxL > 1024) return VGK_ETOOLONG;
        // best-cell keys pack score*32 + row: keep every reachable score below 2047
        if ((int64_t)maxL * std::max(ctx->max_score, 0) + 2 * (int64_t)ctx->max_bonus > 2046) return VGK_EUNSUPPORTED;
    }
    // Offsets into the shared arenas: prefix sums over the problems, in chunks — totals per chunk on the host threads, a serial
    // scan over the chunk totals, then the offsets inside every chunk on the host threads again.
    struct Totals { uint64_t scratch = 0, ops = 0, reads = 0, prof = 0, cols = 0, nodes = 0, preds = 0, cells = 0, tb_cells = 0, in_bytes = 0; int status = VGK_OK; bool want_tb = false, far = false; };
    const uint32_t n_chunks = chunk_count(n);
    std::vector<Totals> chunk_tot(n_chunks), chunk_at(n_chunks);
    parallel_chunks(n, [&](uint32_t lo, uint32_t hi, uint32_t c) {
        Totals t;
        for (uint32_t i = lo; i < hi; ++i) {
            const Sizes& z = sizes[i]; const ProbDesc& d = probs[i];
            if (z.status != VGK_OK) { t.status = z.status; break; }               // the first failing problem of the chunk
            t.want_tb |= z.want_tb; t.far |= z.far;
            t.scratch += (uint64_t)d.n_slots * d.Lpad; t.ops += d.ops_cap;
            t.nodes += z.nodes; t.reads += z.reads; t.prof += z.prof; t.cols += (z.cols + 3ull) & ~3ull; t.preds += z.preds;   // every column stream starts on a dword
            t.cells += (uint64_t)d.R * d.L;
            if (d.flags & VGK_GSSW_TRACEBACK) t.tb_cells += (uint64_t)d.R * d.L;
            t.in_bytes += (uint64_t)problems[i].read_len + d.R + 8ull * z.nodes + 4ull * z.preds;
        }
        chunk_tot[c] = t;
    });
    Totals all;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const Totals& t = chunk_tot[c];
        if (t.status != VGK_OK) return t.status;
        chunk_at[c] = all;
        all.scratch += t.scratch; all.ops += t.ops; all.nodes += t.nodes; all.reads += t.reads; all.prof += t.prof; all.cols += t.cols; all.preds += t.preds;
        all.cells += t.cells; all.tb_cells += t.tb_cells; all.in_bytes += t.in_bytes; all.want_tb |= t.want_tb; all.far |= t.far;
    }
    const uint64_t scratch_words = all.scratch, ops_total = all.ops, n_reads = all.reads, n_prof = all.prof, n_cols = all.cols, n_nodes = all.nodes, n_preds = all.preds;
    if (scratch_words >= (1ull << 32) || ops_total >= (1ull << 32) || n_cols >= (1ull << 32) || n_reads >= (1ull << 32) || n_nodes >= (1ull << 32) || n_preds >= (1ull << 32)) return VGK_ETOOBIG;
    b->want_tb = all.want_tb; b->cells = all.cells; b->tb_cells = all.tb_cells; b->in_bytes = all.in_bytes;
    parallel_chunks(n, [&](uint32_t lo, uint32_t hi, uint32_t c) {
        Totals at = chunk_at[c];
        for (uint32_t i = lo; i < hi; ++i) {
            ProbDesc& d = probs[i]; Sizes& z = sizes[i];
            d.node_off = (uint32_t)at.nodes; d.read_off = (uint32_t)at.reads;
            if (ctx->has_qa) d.prof_off = (uint32_t)at.prof;
            d.col_off = (uint32_t)at.cols;
            d.scratch_off = (uint32_t)at.scratch; d.ops_off = (uint32_t)at.ops;
            z.pred_at = (uint32_t)at.preds;                   // pred_begin is an offset into the shared predecessor arena
            at.scratch += (uint64_t)d.n_slots * d.Lpad; at.ops += d.ops_cap;
            at.nodes += z.nodes; at.reads += z.reads; at.prof += z.prof; at.cols += (z.cols + 3ull) & ~3ull; at.preds += z.preds;
        }
    });
    // the shared arenas are page-locked staging buffers kept on the context: no zero-fill, no page faults, full-rate DMA
    lap("prefix");
    uint8_t* colinfo = (uint8_t*)lease.s->get(0, n_cols + 8);            // + 8: leaders prefetch one word ahead
    uint8_t* reads = (uint8_t*)lease.s->get(1, n_reads + 8);
    uint32_t* prof = (uint32_t*)lease.s->get(2, sizeof(uint32_t) * (n_prof + 4));
    NodeRec* nodes = (NodeRec*)lease.s->get(3, sizeof(NodeRec) * (n_nodes + 1));
    uint32_t* preds = (uint32_t*)lease.s->get(4, sizeof(uint32_t) * (n_preds + 1));
    if (!colinfo || !reads || !prof || !nodes || !preds) return VGK_ENOMEM;
    std::memset(colinfo + n_cols, CI_INVALID, 8);
    lap("staging");
    parallel_for(n, [&](uint32_t i, unsigned t) {
        const vgk_gssw_problem& p = problems[i];
        const vgk_graph& g = p.graph;
        const ProbDesc& d = probs[i];
        const uint32_t mode = p.flags & 15u; const bool xdrop = mode == VGK_XDROP_PINNED;
        uint8_t* rd = reads + d.read_off;
        if (xdrop) *rd++ = 5;                                  // row 0 = no read base consumed yet
        code_bases<true>(rd, p.read, p.read_len);
        if (ctx->has_qa) {
            const uint32_t S = ctx->scale;
            uint32_t* pf = prof + d.prof_off;
            if (xdrop) *pf++ = 0;                              // row 0 = nothing consumed
            for (uint32_t r = 0; r < p.read_len; ++r) {
                const int code = nt_read(p.read[r]);
                uint32_t w = 0;
                for (int b4 = 0; b4 < 4; ++b4)
                    w |= (uint32_t)((ctx->qmat[25 * p.qual[r] + 5 * b4 + code] + (int)ctx->bias) * (int)S) << (8 * b4);
                const uint32_t row = r + (xdrop ? 1u : 0u);
                w += 0x01010101u * row_bonus(d.bonus_start, d.bonus_end, row, d.L);
                pf[r] = w;
            }
        }
        Flags& f = thread_flags[t];
        node_flags(p, xdrop, mode, f);
        uint8_t* ci_out = colinfo + d.col_off;
        NodeRec* nrs = nodes + d.node_off;
        uint32_t* pr = preds + sizes[i].pred_at;
        uint32_t col = 0, slots = 0, seq_pos = 0, np = 0;
        for (uint32_t v = 0; v < g.n_nodes; ++v) {
            NodeRec nr;
            nr.col_start = col; nr.col_end = col + g.node_len[v];
            nr.pred_begin = sizes[i].pred_at + np; nr.n_pred = g.pred_off[v + 1] - g.pred_off[v];
            for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) pr[np++] = g.pred_idx[k];
            nr.slot = f.store[v] ? (int32_t)slots++ : -1;
            nr.pinning = (mode == VGK_GSSW_PINNED && p.pinning[v]) ? 1u : 0u;
            nrs[v] = nr;
            if (const uint32_t len = g.node_len[v]) {              // the node's bases, then the marks on its first and last column
                code_bases<false>(ci_out + col, g.seq + seq_pos, len);
                ci_out[col] |= (uint8_t)(CI_NODE_START | (f.slow[v] ? CI_SEED_SLOW : 0));
                if (f.store[v]) ci_out[col + len - 1] |= (uint8_t)CI_STORE_END;
                seq_pos += len;
            }
            col = nr.col_end;
        }
        for (uint32_t c = d.col_off + col; c & 3u; ++c) colinfo[c] = (uint8_t)CI_INVALID;      // pad to the next problem's dword
    });

    // ---- length buckets: reads with the same (K, G) geometry share wavefronts; inside a bucket reads are sorted by
    //      graph size so that the pairs of a wavefront finish together.  One fill launch per K; G is per wavefront.
    lap("pass3");
    // The encoded arenas (0.9 of the 1.0 GB that goes to HBM) start their way there now, on the copy stream, and travel while the
    // host orders the reads and describes the wavefronts.  Device arenas come from the context (under its lock); the copies run
    // without it, so that a caller can pack the next batch while the previous one runs and is fetched.
    GsswParams& P = b->P;
    int rc;
    std::unique_lock<std::mutex> lk(ctx->mu);
    b->lane = (std::getenv("VGAMD_ONE_STREAM") ? 0 : (int)(ctx->batch_seq++ & 1u));
    // on failure release whatever was allocated (vgk_batch_free takes the context lock itself)
    auto fail = [&](int code) { vgk_batch* t = hb.release(); if (lk.owns_lock()) lk.unlock(); ctx->be->sync_side(); vgk_batch_free(t); return code; };   // (copies in flight must land before the arenas go back to the pool)
    auto issue_uploads = [&]() -> int {
        for (const vgk_batch::Upload& u : b->uploads) { const int e = ctx->be->upload_side(u.dst, u.src, u.bytes); if (e) return e; }
        b->uploads.clear();
        return VGK_OK;
    };
    if ((rc = to_device(b, colinfo, n_cols + 8, P.colinfo))) return fail(rc);
    if ((rc = to_device(b, reads, n_reads, P.reads, 8))) return fail(rc);
    if ((rc = to_device(b, prof, n_prof, P.prof, 4))) return fail(rc);
    if ((rc = to_device(b, nodes, n_nodes, P.nodes))) return fail(rc);
    if ((rc = to_device(b, preds, n_preds, P.preds, 1))) return fail(rc);
    lk.unlock();
    if ((rc = issue_uploads())) return fail(rc);
    lap("arenas");
    auto gkey = [&](uint32_t i) { return ((probs[i].geom & 0xffu) << 8) | ((probs[i].geom >> 8) & 0xffu); };   // (K, G)
    // stable order by (K, G) ascending, then graph size descending: two counting-sort passes over 16-bit digits (sizes beyond 65 535
    // columns share the last digit value; the order only balances wavefronts, it never changes a result)
    std::vector<uint32_t> idx(n);
    {
        // one stable counting-sort pass src -> dst by key(i) in [0, 65536): on the host threads when the keys present span few
        // values (slices of the input, one histogram per slice, offsets by (value, slice)), else serially
        auto pass = [&](const uint32_t* src, uint32_t* dst, auto key) {
            constexpr uint32_t SLICES = 64, MAX_RANGE = 8192;
            uint32_t lo_k = 0xffffffffu, hi_k = 0;
            if (n >= 65536) {
                std::vector<uint32_t> mn(SLICES, 0xffffffffu), mx(SLICES, 0);
                parallel_tasks(SLICES, [&](uint32_t sl) {
                    const uint64_t a0 = (uint64_t)n * sl / SLICES, a1 = (uint64_t)n * (sl + 1) / SLICES;
                    uint32_t a = 0xffffffffu, b2 = 0;
                    for (uint64_t k = a0; k < a1; ++k) { const uint32_t v = key(src ? src[k] : (uint32_t)k); a = std::min(a, v); b2 = std::max(b2, v); }
                    mn[sl] = a; mx[sl] = b2;
                });
                for (uint32_t sl = 0; sl < SLICES; ++sl) { lo_k = std::min(lo_k, mn[sl]); hi_k = std::max(hi_k, mx[sl]); }
            }
            if (n >= 65536 && hi_k - lo_k < MAX_RANGE) {
                const uint32_t range = hi_k - lo_k + 1;
                std::vector<uint32_t> hist((size_t)SLICES * range, 0u);
                auto each_slice = [&](auto body) { parallel_tasks(SLICES, body); };
                each_slice([&](uint32_t sl) {
                    uint32_t* h = hist.data() + (size_t)sl * range;
                    const uint64_t a0 = (uint64_t)n * sl / SLICES, a1 = (uint64_t)n * (sl + 1) / SLICES;
                    for (uint64_t k = a0; k < a1; ++k) ++h[key(src ? src[k] : (uint32_t)k) - lo_k];
                });
                uint32_t at = 0;
                for (uint32_t v = 0; v < range; ++v)
                    for (uint32_t sl = 0; sl < SLICES; ++sl) { uint32_t& c = hist[(size_t)sl * range + v]; const uint32_t cnt = c; c = at; at += cnt; }
                each_slice([&](uint32_t sl) {
                    uint32_t* h = hist.data() + (size_t)sl * range;
                    const uint64_t a0 = (uint64_t)n * sl / SLICES, a1 = (uint64_t)n * (sl + 1) / SLICES;
                    for (uint64_t k = a0; k < a1; ++k) { const uint32_t i = src ? src[k] : (uint32_t)k; dst[h[key(i) - lo_k]++] = i; }
                });
                return;
            }
            std::vector<uint32_t> count(65537, 0u);
            for (uint32_t k = 0; k < n; ++k) ++count[key(src ? src[k] : k) + 1];
            for (uint32_t k = 0; k < 65536; ++k) count[k + 1] += count[k];
            for (uint32_t k = 0; k < n; ++k) { const uint32_t i = src ? src[k] : k; dst[count[key(i)]++] = i; }
        };
        std::vector<uint32_t> tmp(n);
        pass(nullptr, tmp.data(), [&](uint32_t i) { return 0xffffu - std::min<uint32_t>(probs[i].R, 0xffffu); });
        pass(tmp.data(), idx.data(), gkey);
    }
    lap("sort");
    // buckets of equal (K, G) are runs of idx; a bucket's reads pair up in order and its pairs fill wavefronts of 64 / G pairs.
    // The bucket bounds are found serially (there are few), the wavefronts are then described on the host threads.
    struct Bucket { uint32_t s0, s1, K, G, pair0, pair1, wave0; };
    std::vector<Bucket> buckets;
    std::vector<FillLaunch>& launches = b->launches;
    uint32_t n_pairs = 0, n_waves = 0;
    for (uint32_t s0 = 0; s0 < n;) {
        const uint32_t gk = gkey(idx[s0]);
        uint32_t lo = s0, hi = n;                              // first position whose key differs (idx is sorted by key)
        while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (gkey(idx[mid]) == gk) lo = mid; else hi = mid; }
        const uint32_t s1 = hi, K = gk >> 8, G = gk & 0xffu, gpw = 64 / G;
        Bucket bk{s0, s1, K, G, n_pairs, n_pairs + (s1 - s0 + 1) / 2, n_waves};
        if (launches.empty() || launches.back().K != K) launches.push_back(FillLaunch{K, n_waves, 0});
        n_pairs = bk.pair1; n_waves += (bk.pair1 - bk.pair0 + gpw - 1) / gpw;
        launches.back().wave_count = n_waves - launches.back().wave_begin;
        buckets.push_back(bk);
        s0 = s1;
    }
    // The speculative fill (GsswParams::spec_fill): a batch of mostly local alignments with tracebacks, one fill launch, one lanes-per-pair
    // geometry, on the stored-codes traceback.  Its first fill writes no codes; the traceback arena then only serves the wavefronts of the
    // reads walk_diag_one leaves — sized for as many of them as there are wavefronts now, each as large as the widest window asks.
    uint32_t local_tb = 0;
    for (uint32_t i = 0; i < n; ++i) local_tb += (problems[i].flags & (15u | VGK_GSSW_TRACEBACK)) == (uint32_t)(VGK_GSSW_LOCAL | VGK_GSSW_TRACEBACK) ? 1u : 0u;
    const bool fused_env = std::getenv("VGAMD_FUSED_TRACEBACK") && std::atoi(std::getenv("VGAMD_FUSED_TRACEBACK"));
    const bool walk2 = !fused_env && n >= 1024 && 2ull * local_tb >= n && !std::getenv("VGAMD_WALK_ONE_PASS");
    bool spec = walk2 && b->want_tb && buckets.size() == 1 && launches.size() == 1 && default_tb_mode(fused_env ? 1 : 0, !all.far) == TB_CODES && !std::getenv("VGAMD_NO_SPEC_FILL");
    uint64_t refill_slot = 0;
    if (spec) { uint32_t rmax_all = 0; for (uint32_t i = 0; i < n; ++i) rmax_all = std::max(rmax_all, probs[i].R);
                refill_slot = tb_wave_dwords(rmax_all + buckets[0].G - 1, buckets[0].K); }
    std::vector<uint32_t> order((size_t)(spec ? 4 : 2) * n_pairs, 0xffffffffu);        // pairs (the speculative fill's second half: the pairs of the wavefronts filled again)
    std::vector<WaveDesc> waves((size_t)(spec ? 2 : 1) * n_waves);
    parallel_for(n_waves, [&](uint32_t w, unsigned) {
        size_t bi = 0;
        while (bi + 1 < buckets.size() && buckets[bi + 1].wave0 <= w) ++bi;
        const Bucket& bk = buckets[bi];
        const uint32_t gpw = 64 / bk.G, pw = bk.pair0 + (w - bk.wave0) * gpw;
        WaveDesc wd{};
        wd.first_pair = pw; wd.G = bk.G; wd.pair_end = bk.pair1;
        uint32_t rmax = 0;
        for (uint32_t q = 0; q < gpw && pw + q < bk.pair1; ++q)
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t k = bk.s0 + 2 * (pw + q - bk.pair0) + h;
                const uint32_t i = k < bk.s1 ? idx[k] : 0xffffffffu;
                order[2 * (size_t)(pw + q) + h] = i;
                if (i == 0xffffffffu) continue;
                rmax = std::max(rmax, probs[i].R);
                probs[i].wave = w; probs[i].lane0 = q * bk.G; probs[i].geom = bk.K | (bk.G << 8) | (h << 16);
            }
        wd.n_steps = rmax ? rmax + bk.G - 1 : 0;
        wd.tb_off = b->want_tb ? tb_wave_dwords(wd.n_steps, bk.K) : 0;   // size for now
        waves[w] = wd;
    });
    uint64_t tb_dwords = 0;
    for (uint32_t w = 0; w < n_waves; ++w) { WaveDesc& wd = waves[w]; const uint64_t size = wd.tb_off; wd.tb_off = tb_dwords; tb_dwords += size; b->wave_steps += wd.n_steps; }
    if (spec && (uint64_t)n_waves * refill_slot > tb_dwords + tb_dwords / 2) spec = false;      // windows of very different widths: the worst case would need half as much again
    if (spec) tb_dwords = (uint64_t)n_waves * refill_slot;

    lap("waves");
    // the descriptors (their wave / lane fields are final now), the wavefronts and the order follow; the output arenas are allocated
    lk.lock();
    if ((rc = to_device(b, (const ProbDesc*)probs, (size_t)n, P.probs))) return fail(rc);
    if ((rc = to_device(b, waves, P.waves))) return fail(rc);
    if ((rc = to_device(b, order, P.order, 2))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)scratch_words + 16, P.scratch))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)tb_dwords + 4, P.tb))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)tb_best_entries(n), P.best))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)n + 1, P.results))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)ops_total + 1, P.ops))) return fail(rc);
    P.spec_fill = 0; P.wave_limit = nullptr; P.refill_count = nullptr; P.refill_wave0 = P.refill_pair0 = P.refill_G = P.refill_K = 0; P.refill_slot = 0;
    if (spec) {
        if ((rc = dev_alloc(b, (size_t)4, P.refill_count))) return fail(rc);
        P.spec_fill = 1; P.refill_wave0 = n_waves; P.refill_pair0 = n_pairs; P.refill_G = buckets[0].G; P.refill_K = buckets[0].K; P.refill_slot = refill_slot;
    }
    lap("allocs");
    lk.unlock();
    if ((rc = issue_uploads())) return fail(rc);
    P.wave_begin = 0; P.wave_count = 0; P.K = 0;                 // set per fill launch from b->launches
    P.n_problems = n; P.n_pairs = n_pairs; P.n_waves = n_waves;
    const uint32_t S = ctx->scale;     // 8 whenever the scaled profile bytes still fit (vg's default 1/4/6/1/5 does)
    for (int q = 0; q < 6; ++q) P.prof4[q] = ctx->prof4[q] * S;      // bytes stay < 256, no carries between them
    P.bias = ctx->bias * S; P.go = ctx->sc.gap_open * S; P.ge = ctx->sc.gap_extend * S; P.bonus = ctx->sc.full_length_bonus * (int32_t)S;
    P.scale = S; P.xoff = XOFF * S;
    P.want_tb = b->want_tb ? 1 : 0;
    P.fused = 0;
    if (const char* e = std::getenv("VGAMD_FUSED_TRACEBACK")) P.fused = std::atoi(e) ? 1 : 0;
    P.dbg = std::getenv("VGAMD_TB_DBG") ? std::atoi(std::getenv("VGAMD_TB_DBG")) : 0; P.tb_mode = default_tb_mode(P.fused, !all.far);
    P.walk_passes = walk2 && !P.fused ? 2 : 1;                          // (local alignments with a traceback: what walk_diag_one serves)
    if (P.walk_passes != 2 || P.tb_mode != TB_CODES) P.spec_fill = 0;
    P.key3 = 0;
    if (P.spec_fill) {                                                  // the first fill's column key maximum by v_pk_maximum3_f16 (gssw_device.hpp, K3)
        uint32_t longest = 0; bool xdrop = false;
        for (uint32_t i = 0; i < n; ++i) { longest = std::max(longest, probs[i].L); xdrop = xdrop || (problems[i].flags & 15u) == (uint32_t)VGK_XDROP_PINNED; }
        P.key3 = gssw_key3_ok(S, ctx->has_qa, xdrop, longest, (uint32_t)std::max(0, ctx->max_score), (uint32_t)std::max(0, ctx->max_bonus)) && !std::getenv("VGAMD_NO_KEY3") ? 1u : 0u;
    }
    std::memcpy(P.matrix, ctx->sc.matrix, 25);
    b->ops_total = ops_total;
    if ((rc = ctx->be->sync_side())) return fail(rc);     // inputs are resident in HBM when pack returns (the uploads have their own stream)
    lap("uploads");
    *out = hb.release();
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

// the miss count of a speculative run that has finished -> the context's policy (callers hold ctx->mu; the run's kernels are complete)
static void observe_speculation(vgk_batch* b) {
    if (!b->ran || !b->ran_spec || b->spec_observed || !b->P.refill_count) return;
    b->spec_observed = true;
    uint32_t refilled = 0;
    if (b->ctx->be->download_fetch(&refilled, b->P.refill_count, sizeof refilled) != VGK_OK) return;
    b->ctx->spec.observe(b->P.n_waves ? (double)refilled / (double)b->P.n_waves : 0.0, b->spec_probe);
}

int vgk_gssw_run(vgk_batch* b) try {
    if (!b) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    if (!b->done) b->done = b->ctx->be->event_create();
    GsswParams P = b->P;
    if (b->P.spec_fill) {
        // (a resident batch that is run again without a fetch in between: its own last run tells as much as a fetched one)
        if (b->ran && b->ran_spec && !b->spec_observed && b->ctx->be->event_done(b->done)) observe_speculation(b);
        bool probe = false;
        const bool speculate = b->ctx->spec.decide(&probe);
        b->spec_probe = probe;
        if (!speculate) {                                                  // the plain fill with codes over the same arenas (they hold either form)
            P.spec_fill = 0; P.wave_limit = nullptr;
            P.restore_probs = b->probs_displaced ? 1 : 0;                   // (an earlier speculative run moved its missed reads' descriptors to their second wavefronts)
            b->probs_displaced = false;
        } else b->probs_displaced = true;
        b->ran_spec = speculate; b->spec_observed = !speculate;
    }
    const int rc = b->ctx->be->run_gssw_on(b->lane, P, b->launches.data(), (uint32_t)b->launches.size(), true, b->done);
    if (rc == VGK_OK) b->ran = true;
    return rc;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_batch_sync(vgk_batch* b) try {
    if (!b) return VGK_EINVAL;
    return b->done ? b->ctx->be->event_wait(b->done) : b->ctx->be->sync();        // this batch's kernels; no context lock: another thread may be packing the next batch
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

// The usual way back: the ops are packed behind each other on the device (a read uses a handful of its ops_per_problem slots),
// results and ops cross PCIe once into page-locked staging, and host threads copy them into the caller's arrays.  Returns
// VGK_EUNSUPPORTED when the host path has to do it (no packing kernels, or the ops do not fit the caller's array).
static int fetch_packed_on_device(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    vgk_ctx* ctx = b->ctx; Backend* be = ctx->be.get();
    const uint32_t n = b->n;
    if (!n) return VGK_EUNSUPPORTED;
    const uint32_t blocks = (n + Backend::OPS_SCAN_BLOCK - 1) / Backend::OPS_SCAN_BLOCK;
    std::vector<vgk_ctx::Pooled> mine;
    auto take = [&](uint64_t bytes) -> void* { std::lock_guard<std::mutex> lk(ctx->mu); uint64_t got = 0; void* p = ctx->dev_take(bytes, got); if (p) mine.push_back({p, got}); return p; };
    struct GiveBack { vgk_ctx* ctx; std::vector<vgk_ctx::Pooled>& v; ~GiveBack() { ctx->be->sync_fetch(); std::lock_guard<std::mutex> lk(ctx->mu); for (auto& q : v) ctx->dev_give(q.p, q.bytes); } } give_back{ctx, mine};
    uint32_t* offs = (uint32_t*)take((uint64_t)n * 4);
    uint32_t* sums = (uint32_t*)take(((uint64_t)blocks + 8) * 4);
    if (!offs || !sums) return VGK_ENOMEM;
    uint64_t total = 0;
    int rc = be->ops_offsets(b->P.results, n, offs, sums, &total);
    if (rc) return rc;                                                   // VGK_EUNSUPPORTED included
    if (total && (!ops || total > ops_cap || !b->want_tb)) return VGK_EUNSUPPORTED;
    vgk_result* out_res = (vgk_result*)take((uint64_t)n * sizeof(vgk_result));
    vgk_op* out_ops = (vgk_op*)take((total + 1) * sizeof(vgk_op));
    if (!out_res || !out_ops) return VGK_ENOMEM;
    if ((rc = be->ops_gather(b->P.results, b->P.ops, n, offs, sums, out_res, out_ops))) return rc;
    struct StagingLease {
        vgk_ctx* ctx; std::unique_ptr<vgk_ctx::Staging> s;
        ~StagingLease() { if (s) ctx->staging_release(std::move(s)); }
    } lease{ctx, ctx->staging_acquire()};
    const uint64_t res_bytes = (uint64_t)n * sizeof(vgk_result), ops_bytes = total * sizeof(vgk_op);
    uint8_t* stage = (uint8_t*)lease.s->get(5, res_bytes + ops_bytes);
    if (!stage) return VGK_ENOMEM;
    if ((rc = be->download_fetch(stage, out_res, res_bytes))) return rc;
    if (ops_bytes && (rc = be->download_fetch(stage + res_bytes, out_ops, ops_bytes))) return rc;
    auto copy_out = [](void* dst, const uint8_t* src, uint64_t bytes) {
        const uint64_t chunk = 16384;
        parallel_for((uint32_t)((bytes + chunk - 1) / chunk), [&](uint32_t c, unsigned) {
            const uint64_t at = (uint64_t)c * chunk;
            std::memcpy((uint8_t*)dst + at, src + at, (size_t)std::min(chunk, bytes - at));
        });
    };
    copy_out(results, stage, res_bytes);
    if (ops_bytes) copy_out(ops, stage + res_bytes, ops_bytes);
    b->alg_bytes = b->in_bytes + 16ull * n + 2ull * total + b->tb_cells;
    if (ops_written) *ops_written = (size_t)total;
    return VGK_OK;
}

// an extension batch's results and ops, from the problem's own node numbering to the window's (vgk_gssw_pack_extensions)
static void translate_extension_results(const vgk_batch* b, vgk_result* results, vgk_op* ops) {
    parallel_chunks(b->n, [&](uint32_t lo, uint32_t hi, uint32_t) {
        for (uint32_t i = lo; i < hi; ++i) {
            vgk_result& r = results[i];
            if (r.status != VGK_OK) continue;
            if (b->ext_count[i] == WIN_EXT_DUMMY) { r.score = 0; r.n_ops = 0; r.end_node = r.end_offset = r.end_read = -1; r.first_offset = 0; continue; }
            const uint32_t* nodes = b->ext_nodes.data() + b->ext_off[i]; const uint32_t cnt = b->ext_count[i];
            if (r.end_node >= 0 && (uint32_t)r.end_node < cnt) r.end_node = (int32_t)nodes[r.end_node];
            if (ops) for (uint32_t k = 0; k < r.n_ops; ++k) { vgk_op& o = ops[r.ops_begin + k]; if (o.node < cnt) o.node = nodes[o.node]; }
        }
    });
}

static int gssw_fetch_impl(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written);
int vgk_gssw_fetch(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) try {
    const int rc = gssw_fetch_impl(b, results, ops, ops_cap, ops_written);
    if ((rc == VGK_OK || rc == VGK_EOPS) && b && !b->ext_count.empty()) translate_extension_results(b, results, ops);
    return rc;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

static int gssw_fetch_impl(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!b || !results) return VGK_EINVAL;
    if (!b->ran) { int rc = vgk_gssw_run(b); if (rc) return rc; }
    // wait for THIS batch's kernels (its own event; polling, outside blocking runtime calls and outside the context lock); the
    // packing of the ops and the copies back then run on the fetch stream, beside whatever batch the caller has queued next
    { int rc = b->done ? b->ctx->be->event_wait(b->done) : b->ctx->be->sync(); if (rc) return rc; }
    int rc = b->ctx->be->fetch_after(b->done);
    if (rc) return rc;
    { std::lock_guard<std::mutex> lk(b->ctx->mu); if (b->ran_spec && !b->spec_observed) observe_speculation(b); }      // (the two flags are vgk_gssw_run's, written under the lock)
    rc = fetch_packed_on_device(b, results, ops, ops_cap, ops_written);       // takes the context lock for its arenas only
    if (rc != VGK_EUNSUPPORTED) return rc;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    // host path (a backend without the packing kernels, or a caller whose op array is too small: per-problem VGK_EOPS)
    rc = b->ctx->be->download(results, b->P.results, (size_t)b->n * sizeof(vgk_result));
    if (rc) return rc;
    // the per-problem op slots come back through a page-locked staging buffer (no zero-fill, full-rate DMA) ...
    struct StagingLease {
        vgk_ctx* ctx; std::unique_ptr<vgk_ctx::Staging> s;
        ~StagingLease() { if (s) ctx->staging_release(std::move(s)); }
    } lease{b->ctx, nullptr};
    const vgk_op* all = nullptr;
    if (b->want_tb && b->ops_total) {
        lease.s = b->ctx->staging_acquire();
        vgk_op* buf = (vgk_op*)lease.s->get(5, (uint64_t)b->ops_total * sizeof(vgk_op));
        if (!buf) return VGK_ENOMEM;
        rc = b->ctx->be->download(buf, b->P.ops, (size_t)b->ops_total * sizeof(vgk_op));
        if (rc) return rc;
        all = buf;
    }
    // ... and are packed behind each other in the caller's order: sizes, a prefix sum, then parallel copies
    size_t w = 0; uint64_t alg = 0;
    std::vector<uint32_t> src(b->n);
    std::vector<ProbDesc> fetched;          // batches packed on the device keep no host copy of the descriptors
    const ProbDesc* descs = b->probs;
    if (!descs && b->n) {
        fetched.resize(b->n);
        if ((rc = b->ctx->be->download(fetched.data(), b->P.probs, (size_t)b->n * sizeof(ProbDesc)))) return rc;
        descs = fetched.data();
    }
    for (uint32_t i = 0; i < b->n; ++i) {
        vgk_result& r = results[i];
        const ProbDesc& d = descs[i];
        src[i] = r.ops_begin;
        if (r.status == VGK_OK && r.n_ops) {
            if (!ops || w + r.n_ops > ops_cap) { r.status = VGK_EOPS; r.n_ops = 0; r.ops_begin = (uint32_t)w; continue; }
            r.ops_begin = (uint32_t)w; w += r.n_ops;
        } else { r.n_ops = 0; r.ops_begin = (uint32_t)w; }
        alg += 16 + 2ull * r.n_ops + ((d.flags & VGK_GSSW_TRACEBACK) ? (uint64_t)d.L * d.R : 0);
    }
    if (all) parallel_for(b->n, [&](uint32_t i, unsigned) {
        const vgk_result& r = results[i];
        if (r.status == VGK_OK && r.n_ops) std::memcpy(ops + r.ops_begin, all + src[i], (size_t)r.n_ops * sizeof(vgk_op));
    });
    b->alg_bytes = b->in_bytes + alg;
    if (ops_written) *ops_written = w;
    return VGK_OK;
}

// Rough HBM footprint of one problem (dominated by the 4-bit traceback codes) — used to cut oversize calls into
// sub-batches that fit the device (288 GB on MI355X holds ~8M of the 150 bp x 400 bp problems at once).
static uint64_t problem_device_bytes(const vgk_gssw_problem& p) {
    uint64_t R = 0;
    for (uint32_t v = 0; v < p.graph.n_nodes; ++v) R += p.graph.node_len[v];
    const uint64_t rows = ((uint64_t)p.read_len + 24) / 4 * 4 + 4;
    // traceback codes + inputs + the scratch of saved last columns (4 bytes per padded row for every node that may be stored: all of
    // them on a variant-dense graph) + descriptors; the arenas come from dev_take, which rounds every request up by an eighth
    const uint64_t bytes = rows * (R + 64) / 2 + 16 * (p.read_len + R) + (4 * rows + 64) * p.graph.n_nodes + 512;
    return bytes + bytes / 8;
}

// the limits vgk_gssw_pack enforces on a whole batch, per problem: what the kernels cannot take is reported in that
// problem's status and the rest of the call goes ahead (the caller keeps its CPU path for those reads)
static int problem_limit_status(const vgk_ctx* ctx, const vgk_gssw_problem& p) {
    if (p.read_len == 0 || !p.read || p.graph.n_nodes == 0) return VGK_EINVAL;
    const uint32_t mode = p.flags & 15u;
    const bool xdrop = mode == VGK_XDROP_PINNED;
    const uint32_t rows = p.read_len + (xdrop ? 1u : 0u);
    if (rows > 1024) return VGK_ETOOLONG;
    if ((int64_t)rows * std::max(ctx->max_score, 0) + 2 * (int64_t)ctx->max_bonus > 2046) return VGK_EUNSUPPORTED;
    if (xdrop && (int64_t)p.read_len * std::max(ctx->max_score, 0) + ctx->max_bonus >= (int64_t)XOFF) return VGK_EUNSUPPORTED;
    return VGK_OK;
}

int vgk_gssw_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                   vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) try {
    if (!ctx || (!problems && n) || !results) return VGK_EINVAL;
    uint64_t budget = ctx->be->memory_bytes();
    budget = budget ? budget / 2 : (8ull << 30);          // leave half of HBM to the caller / other contexts
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    // malformed problems are answered here; those outside the packed kernels' range (reads of more than 1024 rows, scores beyond 11
    // bits) take the wide route at the end (gssw_wide_api.cpp); the others run in sub-batches that fit the budget
    std::vector<uint32_t> runnable, wide; runnable.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const int st = problem_limit_status(ctx, problems[i]);
        if (st == VGK_OK) runnable.push_back(i);
        else if (st == VGK_ETOOLONG || st == VGK_EUNSUPPORTED) wide.push_back(i);
        else { std::memset(&results[i], 0, sizeof results[i]); results[i].status = st; }
    }
    const bool all = runnable.size() == n;
    std::vector<vgk_gssw_problem> sub_problems; std::vector<vgk_result> sub_results;
    size_t w_total = 0;
    size_t begin = 0;
    while (begin < runnable.size()) {
        size_t end = begin; uint64_t bytes = 0;
        while (end < runnable.size()) {
            const uint64_t pb = problem_device_bytes(problems[runnable[end]]);
            if (end > begin && bytes + pb > budget) break;
            bytes += pb; ++end;
        }
        const uint32_t m = (uint32_t)(end - begin);
        const vgk_gssw_problem* batch_problems = problems + begin;      // contiguous when nothing was filtered out
        vgk_result* batch_results = results + begin;
        if (!all) {
            sub_problems.resize(m); sub_results.resize(m);
            for (uint32_t k = 0; k < m; ++k) sub_problems[k] = problems[runnable[begin + k]];
            batch_problems = sub_problems.data(); batch_results = sub_results.data();
        }
        vgk_batch* b = nullptr;
        int rc = vgk_gssw_pack(ctx, batch_problems, m, 0, &b);
        if (rc == VGK_ENOMEM && m > 1) {                                 // the estimate was too low for these graphs: half as many at a time
            budget = std::max<uint64_t>(bytes / 2, 1);
            continue;
        }
        if (rc) return rc;
        rc = vgk_gssw_run(b);
        size_t w = 0;
        if (!rc) rc = vgk_gssw_fetch(b, batch_results, ops ? ops + w_total : nullptr, ops_cap - w_total, &w);
        vgk_batch_free(b);
        if (rc) return rc;
        for (uint32_t k = 0; k < m; ++k) {
            batch_results[k].ops_begin += (uint32_t)w_total;             // indices into the caller's whole op array
            if (!all) results[runnable[begin + k]] = batch_results[k];
        }
        w_total += w;
        begin = end;
    }
    if (!wide.empty()) {
        const int rc = wide_align(ctx, problems, wide.data(), (uint32_t)wide.size(), results, ops, ops_cap, &w_total);
        if (rc) return rc;
    }
    if (ops_written) *ops_written = w_total;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_batch_kernel_ms(vgk_batch* b, int which) {
    if (!b) return 0.0;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    if (which == 0 || which == 1 || which == 2) return b->ctx->be->last_ms_on(b->lane, which);
    if (which == 3) return b->P.spec_fill && b->ran_spec ? b->ctx->be->last_ms_on(b->lane, 12) : 0.0;
    return b->ctx->be->last_ms_on(b->lane, 0) + b->ctx->be->last_ms_on(b->lane, 1);
}
uint64_t vgk_batch_cells(vgk_batch* b) { return b ? b->cells : 0; }
uint64_t vgk_batch_alg_bytes(vgk_batch* b) {
    if (!b) return 0;
    if (b->alg_bytes) return b->alg_bytes;
    uint64_t alg = b->in_bytes;      // before fetch: everything except the 2 B / emitted op term
    if (!b->probs) return alg + 16ull * b->n + b->tb_cells;
    for (uint32_t i = 0; i < b->n; ++i) { const ProbDesc& d = b->probs[i]; alg += 16 + ((d.flags & VGK_GSSW_TRACEBACK) ? (uint64_t)d.L * d.R : 0); }
    return alg;
}
uint64_t vgk_batch_device_bytes(vgk_batch* b) { return b ? b->dev_bytes : 0; }
uint64_t vgk_batch_wave_steps(vgk_batch* b) { return b ? b->wave_steps : 0; }
int      vgk_batch_lane(vgk_batch* b) { return b ? b->lane : 0; }
int      vgk_batch_speculated(vgk_batch* b) { if (!b) return 0; std::lock_guard<std::mutex> lk(b->ctx->mu); return b->ran && b->ran_spec ? 1 : 0; }
int      vgk_set_speculation(vgk_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->spec.mode = mode;
    return VGK_OK;
}
int      vgk_speculation_state(vgk_ctx* ctx, uint64_t counters[4], double* last_miss) {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (counters) { counters[0] = ctx->spec.observed; counters[1] = ctx->spec.turned_off; counters[2] = ctx->spec.turned_on; counters[3] = ctx->spec.on ? 0 : ctx->spec.interval; }
    if (last_miss) *last_miss = ctx->spec.last_miss;
    return ctx->spec.on ? 1 : 0;
}

}  // extern "C"
