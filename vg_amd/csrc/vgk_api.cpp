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

using namespace vgk;

struct vgk_batch {
    vgk_ctx* ctx = nullptr;
    uint32_t n = 0;
    bool want_tb = false, ran = false;
    GsswParams P{};
    std::vector<vgk_ctx::Pooled> dev;   // every device allocation of this batch (back to the context's pool when the batch is freed)
    uint64_t cells = 0, in_bytes = 0, dev_bytes = 0, alg_bytes = 0;
    uint64_t ops_total = 0;
    std::vector<ProbDesc> probs;   // kept for fetch()
    std::vector<FillLaunch> launches;   // one per length bucket
};

static inline int nt_read(char ch) {   // gssw_create_nt_table: case-insensitive ACGT, else N
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static inline int nt_ref(char ch) {    // after nonATGCNtoN (src/aligner.cpp:39): upper-case ACGT only
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

template <class T>
static int to_device(vgk_batch* b, const std::vector<T>& v, const T*& out, size_t extra = 0) {
    const size_t bytes = (v.size() + extra) * sizeof(T);
    uint64_t got = 0;
    void* p = b->ctx->dev_take(bytes, got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += bytes;
    if (!v.empty()) { int rc = b->ctx->be->upload(p, v.data(), v.size() * sizeof(T)); if (rc) return rc; }
    out = (const T*)p;
    return VGK_OK;
}

template <class T>
static int to_device(vgk_batch* b, const T* v, size_t count, const T*& out, size_t extra = 0) {      // from a staging arena
    const size_t bytes = (count + extra) * sizeof(T);
    uint64_t got = 0;
    void* p = b->ctx->dev_take(bytes, got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += bytes;
    if (count) { int rc = b->ctx->be->upload(p, v, count * sizeof(T)); if (rc) return rc; }
    out = (const T*)p;
    return VGK_OK;
}
template <class T>
static int dev_alloc(vgk_batch* b, size_t count, T*& out) {
    uint64_t got = 0;
    void* p = b->ctx->dev_take(count * sizeof(T), got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += count * sizeof(T);
    out = (T*)p;
    return VGK_OK;
}

extern "C" {

int vgk_abi_version(void) { return VGK_ABI_VERSION; }

const char* vgk_strerror(int code) {
    switch (code) {
        case VGK_OK: return "ok";
        case VGK_EINVAL: return "invalid argument";
        case VGK_ENODEV: return "no usable HIP device";
        case VGK_ENOMEM: return "out of memory";
        case VGK_ETOOLONG: return "read too long for the engine (max 1024 bases)";
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
    c->prof4[5] = 0;     // X-drop row 0 ("nothing consumed"): no diagonal move can enter it
    *out = c;
    return VGK_OK;
}

int vgk_create(int device, const vgk_scoring* scoring, vgk_ctx** out) { return create_ctx(device, scoring, nullptr, out); }
int vgk_create_qual_adj(int device, const vgk_scoring* scoring, const vgk_qual_adj* qual_adj, vgk_ctx** out) {
    if (!qual_adj) return VGK_EINVAL;
    return create_ctx(device, scoring, qual_adj, out);
}

void vgk_destroy(vgk_ctx* ctx) { delete ctx; }

int vgk_device_info(vgk_ctx* ctx, char* name_out, size_t name_cap, int* cus, size_t* hbm) {
    if (!ctx) return VGK_EINVAL;
    if (name_out && name_cap) { std::strncpy(name_out, ctx->be->name(), name_cap - 1); name_out[name_cap - 1] = 0; }
    if (cus) *cus = ctx->be->compute_units();
    if (hbm) *hbm = ctx->be->memory_bytes();
    return VGK_OK;
}

void vgk_batch_free(vgk_batch* b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->ctx->mu);
        b->ctx->be->sync();
        for (const vgk_ctx::Pooled& q : b->dev) b->ctx->dev_give(q.p, q.bytes);
    }
    delete b;
}

int vgk_gssw_pack(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) {
    if (!ctx || !out || (!problems && n)) return VGK_EINVAL;
    *out = nullptr;
    std::unique_ptr<vgk_batch> hb(new (std::nothrow) vgk_batch());
    if (!hb) return VGK_ENOMEM;
    vgk_batch* b = hb.get();
    b->ctx = ctx; b->n = n;

    uint32_t maxL = 1;
    for (uint32_t i = 0; i < n; ++i) {
        if (problems[i].read_len == 0 || !problems[i].read || problems[i].graph.n_nodes == 0) return VGK_EINVAL;
        maxL = std::max(maxL, problems[i].read_len + ((problems[i].flags & 15u) == VGK_XDROP_PINNED ? 1u : 0u));
    }
    if (maxL > 1024) return VGK_ETOOLONG;
    // best-cell keys pack score*32 + row: keep every reachable score below 2047
    if ((int64_t)maxL * std::max(ctx->max_score, 0) + 2 * (int64_t)ctx->max_bonus > 2046) return VGK_EUNSUPPORTED;
    uint32_t forced = 0;
    if (const char* e = std::getenv("VGAMD_ROWS_PER_LANE")) forced = (uint32_t)std::atoi(e);
    // Lane geometry for a read of `rows` DP rows: rows per lane K (16, 19, 20, 24) and lanes per pair G = ceil(rows/K),
    // the instantiation that spends the fewest issued instructions per useful cell:
    // (K*c_row + c_step) per step buys floor(64/G)*2*rows cells.
    auto geometry = [&](uint32_t rows, uint32_t& K, uint32_t& G) {
        double best_cost = 1e30; K = 16;
        for (uint32_t k : {16u, 19u, 20u, 24u}) {
            const uint32_t g = (rows + k - 1) / k;
            if (g > 64) continue;
            // 19 rows per lane fit a 150 bp read into 8 lanes with 2 padding rows instead of 10; for short reads a fourth
            // launch bucket costs more than the rows it saves (tails workload: 48.9 vs 52.3 M alignments/s)
            if (k == 19 && rows < 128 && forced != k) continue;
            const double cost = (k * 25.0 + 60.0) / ((64 / g) * 2.0 * rows);
            if (forced == k) { K = k; break; }
            if (!forced && cost < best_cost) { best_cost = cost; K = k; }
        }
        G = (rows + K - 1) / K;
    };

    // Three passes over the problems, the first and the last on host threads: (1) validate and size every problem, (2) prefix sums
    // place it in the shared arenas, (3) encode it at its offsets.
    auto T0 = std::chrono::steady_clock::now(); auto lap = [&](const char* w) { if (std::getenv("VGAMD_TIMING")) { auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[pack] %s %.1f ms\n", w, std::chrono::duration<double, std::milli>(t - T0).count()); T0 = t; } };
    std::vector<ProbDesc>& probs = b->probs;
    probs.resize(n);
    struct Sizes { uint32_t reads = 0, prof = 0, cols = 0, nodes = 0, preds = 0; int status = VGK_OK; bool want_tb = false; };
    std::vector<Sizes> sizes(n);
    struct Flags { std::vector<uint8_t> store, slow; };
    std::vector<Flags> thread_flags(MAX_THREADS);
    // store[v]: the node's last column is saved for a successor / the pinned end; slow[v]: its first column is seeded from scratch
    auto node_flags = [&](const vgk_gssw_problem& p, bool xdrop, uint32_t mode, Flags& f) -> int {
        const vgk_graph& g = p.graph;
        f.store.assign(g.n_nodes, 0); f.slow.assign(g.n_nodes, 0);
        for (uint32_t v = 0; v < g.n_nodes; ++v) {
            const uint32_t pb = g.pred_off[v], pe = g.pred_off[v + 1];
            if (pe < pb || g.node_len[v] == 0) return VGK_EINVAL;
            for (uint32_t k = pb; k < pe; ++k) if (g.pred_idx[k] >= v) return VGK_EINVAL;   // not topological
            const bool chain = (pe - pb == 1) && g.pred_idx[pb] + 1 == v;
            f.slow[v] = ((v > 0 || xdrop) && !chain) ? 1 : 0;      // X-drop: node 0 starts from the root column
            if (f.slow[v]) for (uint32_t k = pb; k < pe; ++k) f.store[g.pred_idx[k]] = 1;
            if (mode == VGK_GSSW_PINNED && p.pinning[v]) f.store[v] = 1;
        }
        return VGK_OK;
    };
    parallel_for(n, [&](uint32_t i, unsigned t) {
        const vgk_gssw_problem& p = problems[i];
        const vgk_graph& g = p.graph;
        ProbDesc& d = probs[i]; Sizes& z = sizes[i];
        const uint32_t mode = p.flags & 15u;
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
        for (uint32_t v = 0; v < g.n_nodes; ++v) { col += g.node_len[v]; slots += f.store[v]; }
        if (col >= (1u << 20)) { z.status = VGK_ETOOBIG; return; }
        d.R = (uint32_t)col; d.n_slots = slots;
        uint32_t pK, pG; geometry(d.L, pK, pG);
        d.geom = pK | (pG << 8); d.Lpad = pG * pK; d.wave = 0; d.lane0 = 0;
        d.ops_cap = ops_per_problem ? ops_per_problem : (p.read_len + d.R + 2);
        if (!(p.flags & VGK_GSSW_TRACEBACK)) d.ops_cap = 0;
        z.reads = d.L; z.prof = ctx->has_qa ? d.L : 0; z.cols = d.R; z.nodes = g.n_nodes; z.preds = g.pred_off[g.n_nodes] - g.pred_off[0];
    });
    uint64_t scratch_words = 0, ops_total = 0, n_reads = 0, n_prof = 0, n_cols = 0, n_nodes = 0, n_preds = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (sizes[i].status != VGK_OK) return sizes[i].status;
        ProbDesc& d = probs[i]; const Sizes& z = sizes[i];
        if (z.want_tb) b->want_tb = true;
        d.node_off = (uint32_t)n_nodes; d.read_off = (uint32_t)n_reads;
        if (ctx->has_qa) d.prof_off = (uint32_t)n_prof;
        n_cols = (n_cols + 3) & ~3ull;                       // every problem's column stream starts on a dword
        d.col_off = (uint32_t)n_cols;
        d.scratch_off = (uint32_t)scratch_words; scratch_words += (uint64_t)d.n_slots * d.Lpad;
        d.ops_off = (uint32_t)ops_total; ops_total += d.ops_cap;
        n_nodes += z.nodes; n_reads += z.reads; n_prof += z.prof; n_cols += z.cols; n_preds += z.preds;
        if (scratch_words >= (1ull << 32) || ops_total >= (1ull << 32) || n_cols >= (1ull << 32) || n_reads >= (1ull << 32) || n_nodes >= (1ull << 32) || n_preds >= (1ull << 32)) return VGK_ETOOBIG;
        b->cells += (uint64_t)d.R * d.L;
        b->in_bytes += (uint64_t)problems[i].read_len + d.R + 8ull * z.nodes + 4ull * z.preds;
    }
    // the shared arenas are page-locked staging buffers kept on the context: no zero-fill, no page faults, full-rate DMA
    struct StagingLease {
        vgk_ctx* ctx; std::unique_ptr<vgk_ctx::Staging> s;
        ~StagingLease() { if (s) ctx->staging_release(std::move(s)); }
    } lease{ctx, ctx->staging_acquire()};
    uint8_t* colinfo = (uint8_t*)lease.s->get(0, n_cols + 8);            // + 8: leaders prefetch one word ahead
    uint8_t* reads = (uint8_t*)lease.s->get(1, n_reads + 8);
    uint32_t* prof = (uint32_t*)lease.s->get(2, sizeof(uint32_t) * (n_prof + 4));
    NodeRec* nodes = (NodeRec*)lease.s->get(3, sizeof(NodeRec) * (n_nodes + 1));
    uint32_t* preds = (uint32_t*)lease.s->get(4, sizeof(uint32_t) * (n_preds + 1));
    if (!colinfo || !reads || !prof || !nodes || !preds) return VGK_ENOMEM;
    std::memset(colinfo + n_cols, CI_INVALID, 8);
    {   // pred_begin is an offset into the shared predecessor arena
        uint64_t at = 0;
        for (uint32_t i = 0; i < n; ++i) { sizes[i].preds = (uint32_t)at; at += problems[i].graph.pred_off[problems[i].graph.n_nodes] - problems[i].graph.pred_off[0]; }
    }
    parallel_for(n, [&](uint32_t i, unsigned t) {
        const vgk_gssw_problem& p = problems[i];
        const vgk_graph& g = p.graph;
        const ProbDesc& d = probs[i];
        const uint32_t mode = p.flags & 15u; const bool xdrop = mode == VGK_XDROP_PINNED;
        uint8_t* rd = reads + d.read_off;
        if (xdrop) *rd++ = 5;                                  // row 0 = no read base consumed yet
        for (uint32_t r = 0; r < p.read_len; ++r) rd[r] = (uint8_t)nt_read(p.read[r]);
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
        uint32_t* pr = preds + sizes[i].preds;
        uint32_t col = 0, slots = 0, seq_pos = 0, np = 0;
        for (uint32_t v = 0; v < g.n_nodes; ++v) {
            NodeRec nr;
            nr.col_start = col; nr.col_end = col + g.node_len[v];
            nr.pred_begin = sizes[i].preds + np; nr.n_pred = g.pred_off[v + 1] - g.pred_off[v];
            for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) pr[np++] = g.pred_idx[k];
            nr.slot = f.store[v] ? (int32_t)slots++ : -1;
            nr.pinning = (mode == VGK_GSSW_PINNED && p.pinning[v]) ? 1u : 0u;
            nrs[v] = nr;
            for (uint32_t k = 0; k < g.node_len[v]; ++k, ++seq_pos) {
                uint8_t ci = (uint8_t)nt_ref(g.seq[seq_pos]);
                if (k == 0) { ci |= CI_NODE_START; if (f.slow[v]) ci |= CI_SEED_SLOW; }
                if (k + 1 == g.node_len[v] && f.store[v]) ci |= CI_STORE_END;
                ci_out[col + k] = ci;
            }
            col = nr.col_end;
        }
        for (uint32_t c = d.col_off + col; c & 3u; ++c) colinfo[c] = (uint8_t)CI_INVALID;      // pad to the next problem's dword
    });

    // ---- length buckets: reads with the same (K, G) geometry share wavefronts; inside a bucket reads are sorted by
    //      graph size so that the pairs of a wavefront finish together.  One fill launch per K; G is per wavefront.
    lap("pass3");
    auto gkey = [&](uint32_t i) { return ((probs[i].geom & 0xffu) << 8) | ((probs[i].geom >> 8) & 0xffu); };   // (K, G)
    // stable order by (K, G) ascending, then graph size descending: two counting-sort passes over 16-bit digits (sizes beyond 65 535
    // columns share the last digit value; the order only balances wavefronts, it never changes a result)
    std::vector<uint32_t> idx(n);
    {
        std::vector<uint32_t> tmp(n), count(65537);
        auto low = [&](uint32_t i) { return 0xffffu - std::min<uint32_t>(probs[i].R, 0xffffu); };
        std::fill(count.begin(), count.end(), 0u);
        for (uint32_t i = 0; i < n; ++i) ++count[low(i) + 1];
        for (uint32_t k = 0; k < 65536; ++k) count[k + 1] += count[k];
        for (uint32_t i = 0; i < n; ++i) tmp[count[low(i)]++] = i;
        std::fill(count.begin(), count.end(), 0u);
        for (uint32_t i = 0; i < n; ++i) ++count[gkey(i) + 1];
        for (uint32_t k = 0; k < 65536; ++k) count[k + 1] += count[k];
        for (uint32_t i = 0; i < n; ++i) idx[count[gkey(tmp[i])]++] = tmp[i];
    }
    lap("sort");
    std::vector<uint32_t> order;              // pairs
    std::vector<WaveDesc> waves;
    std::vector<FillLaunch>& launches = b->launches;
    uint64_t tb_dwords = 0;
    for (uint32_t s0 = 0; s0 < n;) {
        uint32_t s1 = s0;
        const uint32_t gk = gkey(idx[s0]);
        while (s1 < n && gkey(idx[s1]) == gk) ++s1;
        const uint32_t K = gk >> 8, G = gk & 0xffu, gpw = 64 / G;
        if (launches.empty() || launches.back().K != K) launches.push_back(FillLaunch{K, (uint32_t)waves.size(), 0});
        const uint32_t pair0 = (uint32_t)(order.size() / 2);
        for (uint32_t k = s0; k < s1; k += 2) {
            order.push_back(idx[k]);
            order.push_back(k + 1 < s1 ? idx[k + 1] : 0xffffffffu);
        }
        const uint32_t pair1 = (uint32_t)(order.size() / 2);
        for (uint32_t pw = pair0; pw < pair1; pw += gpw) {
            WaveDesc wd{};
            wd.first_pair = pw; wd.G = G; wd.pair_end = pair1;
            uint32_t rmax = 0;
            for (uint32_t q = 0; q < gpw && pw + q < pair1; ++q)
                for (uint32_t h = 0; h < 2; ++h) {
                    const uint32_t i = order[2 * (pw + q) + h];
                    if (i == 0xffffffffu) continue;
                    rmax = std::max(rmax, probs[i].R);
                    probs[i].wave = (uint32_t)waves.size(); probs[i].lane0 = q * G; probs[i].geom = K | (G << 8) | (h << 16);
                }
            wd.n_steps = rmax ? rmax + G - 1 : 0;
            wd.tb_off = tb_dwords;
            if (b->want_tb) tb_dwords += (uint64_t)((wd.n_steps + TB_TILE - 1) / TB_TILE * TB_TILE) * 64 * ((K + 3) / 4);
            waves.push_back(wd);
        }
        launches.back().wave_count = (uint32_t)waves.size() - launches.back().wave_begin;
        s0 = s1;
    }
    const uint32_t n_pairs = (uint32_t)(order.size() / 2), n_waves = (uint32_t)waves.size();

    lap("waves");
    std::lock_guard<std::mutex> lk(ctx->mu);
    GsswParams& P = b->P;
    int rc;
    // on failure release whatever was allocated (vgk_batch_free takes the context lock itself)
    auto fail = [&](int code) { vgk_batch* t = hb.release(); ctx->mu.unlock(); vgk_batch_free(t); ctx->mu.lock(); return code; };
    if ((rc = to_device(b, probs, P.probs))) return fail(rc);
    if ((rc = to_device(b, colinfo, n_cols + 8, P.colinfo))) return fail(rc);
    if ((rc = to_device(b, reads, n_reads, P.reads, 8))) return fail(rc);
    if ((rc = to_device(b, prof, n_prof, P.prof, 4))) return fail(rc);
    if ((rc = to_device(b, nodes, n_nodes, P.nodes))) return fail(rc);
    if ((rc = to_device(b, preds, n_preds, P.preds, 1))) return fail(rc);
    if ((rc = to_device(b, waves, P.waves))) return fail(rc);
    if ((rc = to_device(b, order, P.order, 2))) return fail(rc);
    lap("uploads");
    if ((rc = dev_alloc(b, (size_t)scratch_words + 16, P.scratch))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)tb_dwords + 4, P.tb))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)n + 1, P.best))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)n + 1, P.results))) return fail(rc);
    if ((rc = dev_alloc(b, (size_t)ops_total + 1, P.ops))) return fail(rc);
    lap("allocs");
    P.wave_begin = 0; P.wave_count = 0; P.K = 0;                 // set per fill launch from b->launches
    P.n_problems = n; P.n_pairs = n_pairs; P.n_waves = n_waves;
    const uint32_t S = ctx->scale;     // 8 whenever the scaled profile bytes still fit (vg's default 1/4/6/1/5 does)
    for (int q = 0; q < 6; ++q) P.prof4[q] = ctx->prof4[q] * S;      // bytes stay < 256, no carries between them
    P.bias = ctx->bias * S; P.go = ctx->sc.gap_open * S; P.ge = ctx->sc.gap_extend * S; P.bonus = ctx->sc.full_length_bonus * (int32_t)S;
    P.scale = S; P.xoff = XOFF * S;
    P.want_tb = b->want_tb ? 1 : 0;
    P.fused = 0;
    if (const char* e = std::getenv("VGAMD_FUSED_TRACEBACK")) P.fused = std::atoi(e) ? 1 : 0;
    std::memcpy(P.matrix, ctx->sc.matrix, 25);
    b->ops_total = ops_total;
    if ((rc = ctx->be->sync())) return fail(rc);     // inputs are resident in HBM when pack returns
    lap("sync");
    *out = hb.release();
    return VGK_OK;
}

int vgk_gssw_run(vgk_batch* b) {
    if (!b) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    int rc = b->ctx->be->zero(b->P.best, ((size_t)b->n + 1) * sizeof(unsigned long long));
    if (rc) return rc;
    rc = b->ctx->be->run_gssw(b->P, b->launches.data(), (uint32_t)b->launches.size(), true);
    if (rc == VGK_OK) b->ran = true;
    return rc;
}

int vgk_batch_sync(vgk_batch* b) {
    if (!b) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    return b->ctx->be->sync();
}

int vgk_gssw_fetch(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!b || !results) return VGK_EINVAL;
    if (!b->ran) { int rc = vgk_gssw_run(b); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    int rc = b->ctx->be->download(results, b->P.results, (size_t)b->n * sizeof(vgk_result));
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
    for (uint32_t i = 0; i < b->n; ++i) {
        vgk_result& r = results[i];
        const ProbDesc& d = b->probs[i];
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
    return rows * (R + 64) / 2 + 16 * (p.read_len + R) + 64ull * p.graph.n_nodes + 512;
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
                   vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || !results) return VGK_EINVAL;
    uint64_t budget = ctx->be->memory_bytes();
    budget = budget ? budget / 2 : (8ull << 30);          // leave half of HBM to the caller / other contexts
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    // problems outside the kernels' range are answered here; the others run in sub-batches that fit the budget
    std::vector<uint32_t> runnable; runnable.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const int st = problem_limit_status(ctx, problems[i]);
        if (st == VGK_OK) runnable.push_back(i);
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
    if (ops_written) *ops_written = w_total;
    return VGK_OK;
}

double vgk_batch_kernel_ms(vgk_batch* b, int which) {
    if (!b) return 0.0;
    std::lock_guard<std::mutex> lk(b->ctx->mu);
    if (which == 0 || which == 1 || which == 2) return b->ctx->be->last_ms(which);
    return b->ctx->be->last_ms(0) + b->ctx->be->last_ms(1);
}
uint64_t vgk_batch_cells(vgk_batch* b) { return b ? b->cells : 0; }
uint64_t vgk_batch_alg_bytes(vgk_batch* b) {
    if (!b) return 0;
    if (b->alg_bytes) return b->alg_bytes;
    uint64_t alg = b->in_bytes;      // before fetch: everything except the 2 B / emitted op term
    for (const ProbDesc& d : b->probs) alg += 16 + ((d.flags & VGK_GSSW_TRACEBACK) ? (uint64_t)d.L * d.R : 0);
    return alg;
}
uint64_t vgk_batch_device_bytes(vgk_batch* b) { return b ? b->dev_bytes : 0; }

}  // extern "C"
