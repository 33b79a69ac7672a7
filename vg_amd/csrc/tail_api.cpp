// tail_api.cpp — vgk_tail_forest: a batch of giraffe's tail forests (MinimizerMapper::get_tail_forest, src/minimizer_mapper.cpp:
// 5745-5860) walked on the device and left there as one resident graph whose windows are the trees.
//
// The host's share is the problem array up (20 B per tail) and the result array down (24 B per tail) plus two 4-byte totals that
// size the device allocations; the trees, their bases and the packer's tables never leave HBM (vgk_forest_fetch copies the
// (parent, node, length) triples out for the caller that wants to translate alignments back — TreeSubgraph::translate_down).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>
#include "ctx.hpp"
#include "dgraph.hpp"
#include "haplo.hpp"
#include "batch.hpp"

using namespace vgk;

struct vgk_forest {
    vgk_ctx* ctx = nullptr;
    uint64_t n_nodes = 0;
    int32_t* parent = nullptr; uint32_t* node = nullptr; uint32_t* len = nullptr;     // device, n_nodes each (owned: in graph->dev)
    uint32_t* owner = nullptr;         // device, per tree node the problem it belongs to (vgk_tail_stage's forests only)
    vgk_dgraph* graph = nullptr;
};

extern "C" {

// The walks and the graph construction over problems that are on the device already (d_probs, d_res: n entries each; the caller holds
// the context lock and reads d_res itself).  owner: also record, per tree node, the problem it belongs to (forest->owner).
static int tail_forest_core(vgk_ctx* ctx, const vgk_haplo* index, const vgk_tail_problem* d_probs, vgk_tail_result* d_res, uint32_t n, bool owner, vgk_forest** out) {
    *out = nullptr;
    std::unique_ptr<vgk_forest> fo(new (std::nothrow) vgk_forest());
    std::unique_ptr<vgk_dgraph> dg(new (std::nothrow) vgk_dgraph());
    if (!fo || !dg) return VGK_ENOMEM;
    fo->ctx = ctx; dg->ctx = ctx;
    Backend* be = ctx->be.get();
    // what belongs to the call goes to the context's cached scratch; what belongs to the forest is owned by its graph
    // (from the context's pool of device arenas, like a batch's: a forest lives for one batch of tails, and hipMalloc / hipFree cost
    // more than the walks)
    auto keep = [&](size_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) { dg->dev.push_back(p); dg->dev_size.push_back(got); dg->dev_bytes += bytes; } return p; };
    auto fail = [&](int rc) { be->sync(); for (size_t k = 0; k < dg->dev.size(); ++k) ctx->dev_give(dg->dev[k], dg->dev_size[k]); dg->dev.clear(); dg->dev_size.clear(); return rc; };
    TailParams P{};
    P.index = index->dev; P.n = n;
    const uint32_t per_cu = 512;
    const uint32_t threads = (uint32_t)std::min<uint64_t>(n, (uint64_t)std::max(1, be->compute_units()) * per_cu);
    uint32_t* d_counts = (uint32_t*)ctx->ensure_scratch(52, sizeof(uint32_t) * 2 * (size_t)(n + 1));
    TScratch* d_scratch = (TScratch*)ctx->ensure_scratch(53, sizeof(TScratch) * (size_t)std::max(1u, threads));
    if (!d_counts || !d_scratch) return VGK_ENOMEM;
    uint32_t* d_first = d_counts + (n + 1);
    int rc = be->zero(d_counts, sizeof(uint32_t) * 2 * (size_t)(n + 1));
    P.probs = d_probs; P.results = d_res; P.counts = d_counts; P.first = d_first; P.scratch = d_scratch;
    P.pass = 1;
    if (!rc) rc = be->run_tail(P, threads);                                   // sizes
    if (!rc) rc = be->scan_u32(d_counts, d_first, n + 1);
    uint32_t total = 0;
    if (!rc) rc = be->download(&total, d_first + n, sizeof total);            // synchronises
    if (rc) return fail(rc);
    const size_t N = total;
    int32_t* d_parent = (int32_t*)keep(sizeof(int32_t) * N); uint32_t* d_node = (uint32_t*)keep(sizeof(uint32_t) * N);
    uint32_t* d_len = (uint32_t*)keep(sizeof(uint32_t) * (N + 1));
    uint32_t* d_col = (uint32_t*)keep(sizeof(uint32_t) * (N + 1)); uint32_t* d_po = (uint32_t*)keep(sizeof(uint32_t) * (N + 1));
    uint32_t* d_slot = (uint32_t*)keep(sizeof(uint32_t) * (N + 1));
    uint32_t* d_owner = owner ? (uint32_t*)keep(sizeof(uint32_t) * (N + 1)) : nullptr;
    // per-call tables of the construction: trim, has_pred, store, slow
    uint32_t* d_tmp = (uint32_t*)ctx->ensure_scratch(54, sizeof(uint32_t) * 4 * (N + 1));
    if (!d_parent || !d_node || !d_len || !d_col || !d_po || !d_slot || !d_tmp || (owner && !d_owner)) return fail(VGK_ENOMEM);
    uint32_t* d_trim = d_tmp; uint32_t* d_hasp = d_tmp + (N + 1); uint32_t* d_store = d_tmp + 2 * (N + 1); uint32_t* d_slow = d_tmp + 3 * (N + 1);
    rc = be->zero(d_tmp, sizeof(uint32_t) * 4 * (N + 1));
    if (!rc) rc = be->zero(d_len + N, sizeof(uint32_t));
    P.parent = d_parent; P.node = d_node; P.len = d_len; P.trim = d_trim; P.owner = d_owner; P.pass = 2;
    if (!rc) rc = be->run_tail(P, threads);                                   // the forest itself
    ForestParams F{};
    F.index = index->dev; F.n_nodes = (uint32_t)N; F.parent = d_parent; F.node = d_node; F.len = d_len; F.trim = d_trim;
    F.has_pred = d_hasp; F.store = d_store; F.slow = d_slow;
    if (!rc) rc = be->forest_flags(F);
    if (!rc) rc = be->scan_u32(d_len, d_col, (uint32_t)N + 1);
    if (!rc) rc = be->scan_u32(d_hasp, d_po, (uint32_t)N + 1);
    if (!rc) rc = be->scan_u32(d_store, d_slot, (uint32_t)N + 1);
    uint32_t tot[2] = {0, 0};                                                 // columns, edges
    if (!rc) rc = be->download(&tot[0], d_col + N, sizeof(uint32_t));
    if (!rc) rc = be->download(&tot[1], d_po + N, sizeof(uint32_t));
    if (rc) return fail(rc);
    uint8_t* d_info = (uint8_t*)keep((size_t)tot[0] + 8); uint32_t* d_pi = (uint32_t*)keep(sizeof(uint32_t) * (size_t)tot[1]);
    if (!d_info || !d_pi) return fail(VGK_ENOMEM);
    rc = be->fill(d_info, CI_INVALID, (size_t)tot[0] + 8);
    F.col = d_col; F.pred_off = d_po; F.pred_idx = d_pi; F.info = d_info;
    if (!rc) rc = be->forest_emit(F);
    if (rc) return fail(rc);
    dg->g.col = d_col; dg->g.info = d_info; dg->g.pred_off = d_po; dg->g.pred_idx = d_pi; dg->g.slot = d_slot;
    dg->g.n_nodes = (uint32_t)N; dg->g.n_cols = tot[0];
    fo->n_nodes = N; fo->parent = d_parent; fo->node = d_node; fo->len = d_len; fo->owner = d_owner;
    fo->graph = dg.release();
    *out = fo.release();
    return VGK_OK;
}

int vgk_tail_forest(vgk_ctx* ctx, const vgk_haplo* index, const vgk_tail_problem* problems, uint32_t n,
                    vgk_tail_result* results, vgk_forest** out) try {
    if (!ctx || !index || !out || (n && (!problems || !results)) || !vgk_tables_usable(index->ctx, ctx)) return VGK_EINVAL;
    *out = nullptr;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    vgk_tail_problem* d_probs = (vgk_tail_problem*)ctx->ensure_scratch(50, sizeof(vgk_tail_problem) * (size_t)(n + 1));
    vgk_tail_result* d_res = (vgk_tail_result*)ctx->ensure_scratch(51, sizeof(vgk_tail_result) * (size_t)(n + 1));
    if (!d_probs || !d_res) return VGK_ENOMEM;
    int rc = VGK_OK;
    be->watch(0);
    if (n) rc = be->upload(d_probs, problems, sizeof(vgk_tail_problem) * (size_t)n);
    vgk_forest* fo = nullptr;
    if (!rc) rc = tail_forest_core(ctx, index, d_probs, d_res, n, false, &fo);
    be->watch(1);
    if (!rc && n) rc = be->download(results, d_res, sizeof(vgk_tail_result) * (size_t)n);      // synchronises
    else if (!rc) rc = be->sync();
    if (rc) { if (fo) { vgk_dgraph* g = fo->graph; fo->graph = nullptr; delete fo; if (g) { be->sync(); for (size_t k = 0; k < g->dev.size(); ++k) ctx->dev_give(g->dev[k], g->dev_size[k]); delete g; } } return rc; }
    ctx->tail_ms = be->watch_ms();
    *out = fo;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

uint64_t vgk_forest_size(const vgk_forest* f) { return f ? f->n_nodes : 0; }

int vgk_forest_fetch(const vgk_forest* f, int32_t* parent, uint32_t* node, uint32_t* length) try {
    if (!f) return VGK_EINVAL;
    if (!f->n_nodes) return VGK_OK;
    std::lock_guard<std::mutex> lk(f->ctx->mu);
    Backend* be = f->ctx->be.get();
    int rc = VGK_OK;
    if (parent) rc = be->download(parent, f->parent, sizeof(int32_t) * f->n_nodes);
    if (!rc && node) rc = be->download(node, f->node, sizeof(uint32_t) * f->n_nodes);
    if (!rc && length) rc = be->download(length, f->len, sizeof(uint32_t) * f->n_nodes);
    return rc;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

const vgk_dgraph* vgk_forest_graph(const vgk_forest* f) { return f ? f->graph : nullptr; }

void vgk_forest_destroy(vgk_forest* f) {
    if (!f) return;
    if (f->graph) vgk_graph_destroy(f->graph);
    delete f;
}

// vgk_tail_stage: what vg_amd/host/tail_stage.cpp does on host threads, on the device, over the sets the last vgk_gapless_extend(_seeded)
// call left in HBM (tail_device.hpp "the tails of a batch of extension sets").  The host sees four totals that size allocations and, at
// the end, one int per extension and one per read.
int vgk_pack_windows_impl(vgk_ctx* ctx, const vgk_dgraph* dg, const char* reads, size_t reads_bytes,
                          const vgk_window_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out, bool on_device, uint32_t forced_k,
                          const vgk::WinExt* extensions);
}  // extern "C"
// aligned / tails_cap / ops / ops_cap / written: vgk_tail_stage_aligned's outputs (all null / 0 for vgk_tail_stage)
static int tail_stage_impl(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score, uint64_t stats[4],
                           const bool want_aligned, vgk_tail_alignment* aligned, size_t tails_cap, vgk_op* ops, size_t ops_cap, size_t* written) {
    if (!ctx || !index || !vgk_tables_usable(index->ctx, ctx)) return VGK_EINVAL;
    if (written) written[0] = written[1] = 0;
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> stage(ctx->stage_mu);                     // the sets, the resident reads and the scratch slots they live in stay this call's while mu is let go below
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!ctx->sets.valid) return VGK_EINVAL;
    const uint32_t n = ctx->sets.n; const uint64_t n_ext = ctx->sets.n_ext;
    if (n_ext > ext_cap || (n_ext && !ext_total) || (n && !read_score) || n_ext > 0xfffffff0ull) return VGK_EINVAL;
    std::vector<vgk_ctx::Pooled> temp;                                     // device blocks of this call: back to the pool at the end
    auto take = [&](uint64_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) temp.push_back({p, got}); return p; };
    auto done = [&](int rc) { be->sync(); for (auto& q : temp) ctx->dev_give(q.p, q.bytes); return rc; };
    struct Wall { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); const bool on = std::getenv("VGAMD_TIMING") != nullptr;
                  void operator()(const char* what) { if (!on) return; const auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[tail_stage] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t; } } wall;
    TStageParams S{};
    S.index = index->dev; S.n_reads = n; S.n_ext = (uint32_t)n_ext;
    S.probs = (const GProb*)ctx->sets.probs; S.reads = ctx->sets.reads;
    S.res = (const vgk_gapless_result*)ctx->sets.res; S.ext = (const vgk_extension*)ctx->sets.ext; S.nodes = ctx->sets.nodes;
    S.match = ctx->sc.matrix[0]; S.gap_open = ctx->sc.gap_open; S.gap_extend = ctx->sc.gap_extend; S.bonus = ctx->sc.full_length_bonus;
    const size_t e1 = (size_t)n_ext + 1;
    uint32_t* tab = (uint32_t*)take(sizeof(uint32_t) * 4 * e1);
    int32_t* d_ext_total = (int32_t*)take(sizeof(int32_t) * e1); int32_t* d_read_score = (int32_t*)take(sizeof(int32_t) * ((size_t)n + 1));
    unsigned long long* d_failed = (unsigned long long*)take(64);
    if (!tab || !d_ext_total || !d_read_score || !d_failed) return done(VGK_ENOMEM);
    S.read_of = ctx->sets.read_of; S.cnt_r = tab; S.cnt_l = tab + e1; S.off_r = tab + 2 * e1; S.off_l = tab + 3 * e1;
    S.ext_total = d_ext_total; S.read_score = d_read_score; S.failed = d_failed;
    be->watch(0);
    int rc = S.read_of ? be->zero(tab, sizeof(uint32_t) * 4 * e1) : VGK_EINVAL;
    if (!rc) rc = be->zero(d_failed, 64);
    if (!rc) rc = be->run_tail_stage(S, TS_COUNT);
    if (!rc) rc = be->scan_u32(S.cnt_r, tab + 2 * e1, (uint32_t)e1);
    if (!rc) rc = be->scan_u32(S.cnt_l, tab + 3 * e1, (uint32_t)e1);
    uint32_t tot_rl[2] = {0, 0};
    if (!rc) rc = be->download(&tot_rl[0], S.off_r + n_ext, sizeof(uint32_t));
    if (!rc) rc = be->download(&tot_rl[1], S.off_l + n_ext, sizeof(uint32_t));
    if (rc) return done(rc);
    const uint32_t nt = tot_rl[0] + tot_rl[1];
    S.total_r = tot_rl[0]; S.n_tails = nt;
    wall("counted");
    uint64_t n_trees = 0, tree_nodes = 0;
    unsigned long long failed = 0;
    if (nt) {
        const size_t t1 = (size_t)nt + 1;
        vgk_tail_problem* d_probs = (vgk_tail_problem*)take(sizeof(vgk_tail_problem) * t1);
        vgk_tail_result* d_tres = (vgk_tail_result*)take(sizeof(vgk_tail_result) * t1);
        TMeta* d_meta = (TMeta*)take(sizeof(TMeta) * t1);
        uint32_t* d_len = (uint32_t*)take(sizeof(uint32_t) * 2 * t1); int32_t* d_tscore = (int32_t*)take(sizeof(int32_t) * t1);
        if (!d_probs || !d_tres || !d_meta || !d_len || !d_tscore) return done(VGK_ENOMEM);
        S.problems = d_probs; S.meta = d_meta; S.tail_len = d_len; S.seq_off = d_len + t1; S.tail_score = d_tscore; S.tres = d_tres;
        rc = be->zero(d_len, sizeof(uint32_t) * 2 * t1);
        uint32_t* d_ops_tab = nullptr;
        if (want_aligned) {
            if (written) written[0] = nt;
            if (nt > tails_cap || !aligned) return done(VGK_EOPS);
            S.tail_best = (unsigned long long*)take(sizeof(unsigned long long) * t1);
            d_ops_tab = (uint32_t*)take(sizeof(uint32_t) * 2 * t1);
            S.aligned = (vgk_tail_alignment*)take(sizeof(vgk_tail_alignment) * t1);
            if (!S.tail_best || !d_ops_tab || !S.aligned) return done(VGK_ENOMEM);
            S.ops_cnt = d_ops_tab; S.ops_off = d_ops_tab + t1;
            if (!rc) rc = be->zero(S.tail_best, sizeof(unsigned long long) * t1);
            if (!rc) rc = be->zero(d_ops_tab, sizeof(uint32_t) * 2 * t1);
        }
        if (!rc) rc = be->run_tail_stage(S, TS_TAILS);
        if (!rc) rc = be->scan_u32(d_len, d_len + t1, (uint32_t)t1);
        uint32_t seq_bytes = 0;
        if (!rc) rc = be->download(&seq_bytes, d_len + t1 + nt, sizeof(uint32_t));
        if (rc) return done(rc);
        char* d_seq = (char*)take((uint64_t)seq_bytes + 16);
        if (!d_seq) return done(VGK_ENOMEM);
        S.seq = d_seq;
        rc = be->run_tail_stage(S, TS_BASES);
        be->watch(1); if (!rc) rc = be->sync(); ctx->tail_stage_ms[0] = be->watch_ms(); be->watch(0);
        wall("tails + bases");
        vgk_forest* forest = nullptr;
        if (!rc) rc = tail_forest_core(ctx, index, d_probs, d_tres, nt, true, &forest);
        if (rc) return done(rc);
        wall("forest");
        auto drop_forest = [&]() { vgk_dgraph* g = forest->graph; be->sync(); for (size_t k = 0; k < g->dev.size(); ++k) ctx->dev_give(g->dev[k], g->dev_size[k]); delete g; delete forest; };
        tree_nodes = forest->n_nodes;
        S.parent = forest->parent; S.owner = forest->owner; S.n_nodes = (uint32_t)forest->n_nodes; S.forest_node = forest->node;
        const size_t v1 = (size_t)forest->n_nodes + 1;
        uint32_t* d_root = (uint32_t*)take(sizeof(uint32_t) * 3 * v1);
        if (!d_root) { drop_forest(); return done(VGK_ENOMEM); }
        S.is_root = d_root; S.root_off = d_root + v1; S.root_pos = d_root + 2 * v1;
        rc = be->zero(d_root, sizeof(uint32_t) * 3 * v1);
        if (!rc) rc = be->run_tail_stage(S, TS_ROOT_FLAG);
        if (!rc) rc = be->scan_u32(d_root, d_root + v1, (uint32_t)v1);
        uint32_t nw = 0;
        if (!rc) rc = be->download(&nw, d_root + v1 + forest->n_nodes, sizeof(uint32_t));
        be->watch(1); if (!rc) rc = be->sync(); ctx->tail_stage_ms[1] = be->watch_ms(); be->watch(0);
        if (rc) { drop_forest(); return done(rc); }
        n_trees = nw; S.n_trees = nw;
        // vgk_tail_stage_aligned: per tail the winning tree's ops, packed behind each other, nodes translated; then down
        auto winners_down = [&](const vgk_op* window_ops) -> int {
            S.wops = window_ops;
            int r2 = be->run_tail_stage(S, TS_OPS_COUNT);
            if (!r2) r2 = be->scan_u32(S.ops_cnt, d_ops_tab + t1, (uint32_t)t1);
            uint32_t n_ops = 0;
            if (!r2) r2 = be->download(&n_ops, S.ops_off + nt, sizeof(uint32_t));
            if (r2) return r2;
            if (written) written[1] = n_ops;
            if (n_ops > ops_cap || (n_ops && !ops)) return VGK_EOPS;
            S.out_ops = (vgk_op*)take(sizeof(vgk_op) * ((size_t)n_ops + 1));
            if (!S.out_ops) return VGK_ENOMEM;
            r2 = be->run_tail_stage(S, TS_OPS_COPY);
            if (!r2) r2 = be->download(aligned, S.aligned, sizeof(vgk_tail_alignment) * nt);
            if (!r2 && n_ops) r2 = be->download(ops, S.out_ops, sizeof(vgk_op) * n_ops);
            return r2;
        };
        if (nw) {
            vgk_window_problem* d_win = (vgk_window_problem*)take(sizeof(vgk_window_problem) * (size_t)nw);
            uint32_t* d_wown = (uint32_t*)take(sizeof(uint32_t) * (size_t)nw);
            if (!d_win || !d_wown) { drop_forest(); return done(VGK_ENOMEM); }
            S.windows = d_win; S.win_owner = d_wown;
            rc = be->run_tail_stage(S, TS_ROOT_POS);
            if (!rc) rc = be->run_tail_stage(S, TS_WINDOW);
            if (!rc) rc = be->sync();                                          // the packer runs on the copy stream
            vgk_batch* b = nullptr;
            // The trees of a batch of reads are a few hundred thousand small problems: split over three rows-per-lane classes they make three
            // launches that each leave the device half empty and finish at different times (measured: 4.7 ms; one class, K = 20: 3.8 ms).
            // A million and more (bench.py --workload forest) fill it per class, and the per-problem choice wins again (35.5 vs 37.7 ms).
            const uint32_t forced_k = nw < 400000u ? 20u : 0u;
            if (!rc) { lk.unlock(); rc = vgk_pack_windows_impl(ctx, forest->graph, d_seq, seq_bytes, d_win, nw, ops_per_problem, &b, true, forced_k, nullptr); lk.lock(); }
            be->watch(1); be->sync(); ctx->tail_stage_ms[2] = be->watch_ms();
            wall("windows packed");
            if (!rc) rc = ctx->start_deferred();                                // the extension sets of a VGK_GAPLESS_DEFER call travel under the fills
            if (!rc) { lk.unlock(); rc = vgk_gssw_run(b); if (!rc) rc = vgk_batch_sync(b); lk.lock(); }
            wall("fill + traceback");
            be->watch(0);
            if (!rc) { S.wres = b->P.results; rc = be->run_tail_stage(S, TS_BEST); }
            if (!rc && want_aligned) rc = winners_down(b->P.ops);
            if (!rc) rc = be->sync();
            if (b) { lk.unlock(); vgk_batch_free(b); lk.lock(); }
            if (rc) { drop_forest(); return done(rc); }
        } else if (want_aligned) {
            rc = winners_down(nullptr);
            if (rc) { drop_forest(); return done(rc); }
        }
        rc = be->run_tail_stage(S, TS_TOTAL);
        if (!rc) rc = be->sync();
        drop_forest();
        if (rc) return done(rc);
    }
    rc = be->run_tail_stage(S, TS_READ);
    if (!rc && n_ext) rc = be->download(ext_total, d_ext_total, sizeof(int32_t) * n_ext);
    if (!rc && n) rc = be->download(read_score, d_read_score, sizeof(int32_t) * n);
    if (!rc) rc = be->download(&failed, d_failed, sizeof failed);
    be->watch(1); be->sync(); ctx->tail_stage_ms[3] = be->watch_ms();
    if (stats) { stats[0] = nt; stats[1] = n_trees; stats[2] = tree_nodes; stats[3] = failed; }
    wall("totals down");
    rc = done(rc);
    wall("blocks back to the pool");
    return rc;
}
extern "C" {
int vgk_tail_stage(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score, uint64_t stats[4]) try {
    int rc = tail_stage_impl(ctx, index, ops_per_problem, ext_total, ext_cap, read_score, stats, false, nullptr, 0, nullptr, 0, nullptr);
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); const int rc2 = ctx->finish_deferred(); if (!rc) rc = rc2; }       // the sets of a VGK_GAPLESS_DEFER call came down meanwhile
    return rc;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
int vgk_tail_stage_aligned(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score,
                           vgk_tail_alignment* tails, size_t tails_cap, vgk_op* ops, size_t ops_cap, size_t written[2], uint64_t stats[4]) try {
    int rc = tail_stage_impl(ctx, index, ops_per_problem, ext_total, ext_cap, read_score, stats, true, tails, tails_cap, ops, ops_cap, written);
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); const int rc2 = ctx->finish_deferred(); if (!rc) rc = rc2; }
    return rc;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
double vgk_tail_stage_last_ms(vgk_ctx* ctx, int which) { return ctx && which >= 0 && which < 4 ? ctx->tail_stage_ms[which] : 0.0; }

double vgk_tail_last_ms(vgk_ctx* ctx) { return ctx ? ctx->tail_ms : 0.0; }

}  // extern "C"
