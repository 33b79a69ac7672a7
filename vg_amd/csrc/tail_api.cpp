// tail_api.cpp — vgk_tail_forest: a batch of giraffe's tail forests (MinimizerMapper::get_tail_forest, src/minimizer_mapper.cpp:
// 5745-5860) walked on the device and left there as one resident graph whose windows are the trees.
//
// The host's share is the problem array up (20 B per tail) and the result array down (24 B per tail) plus two 4-byte totals that
// size the device allocations; the trees, their bases and the packer's tables never leave HBM (vgk_forest_fetch copies the
// (parent, node, length) triples out for the caller that wants to translate alignments back — TreeSubgraph::translate_down).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>
#include "ctx.hpp"
#include "dgraph.hpp"
#include "haplo.hpp"

using namespace vgk;

struct vgk_forest {
    vgk_ctx* ctx = nullptr;
    uint64_t n_nodes = 0;
    int32_t* parent = nullptr; uint32_t* node = nullptr; uint32_t* len = nullptr;     // device, n_nodes each (owned: in graph->dev)
    vgk_dgraph* graph = nullptr;
};

extern "C" {

int vgk_tail_forest(vgk_ctx* ctx, const vgk_haplo* index, const vgk_tail_problem* problems, uint32_t n,
                    vgk_tail_result* results, vgk_forest** out) {
    if (!ctx || !index || !out || (n && (!problems || !results)) || index->ctx != ctx) return VGK_EINVAL;
    *out = nullptr;
    std::unique_ptr<vgk_forest> fo(new (std::nothrow) vgk_forest());
    std::unique_ptr<vgk_dgraph> dg(new (std::nothrow) vgk_dgraph());
    if (!fo || !dg) return VGK_ENOMEM;
    fo->ctx = ctx; dg->ctx = ctx;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    // what belongs to the call goes to the context's cached scratch; what belongs to the forest is owned by its graph
    // (from the context's pool of device arenas, like a batch's: a forest lives for one batch of tails, and hipMalloc / hipFree cost
    // more than the walks)
    auto keep = [&](size_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) { dg->dev.push_back(p); dg->dev_size.push_back(got); dg->dev_bytes += bytes; } return p; };
    auto fail = [&](int rc) { be->sync(); for (size_t k = 0; k < dg->dev.size(); ++k) ctx->dev_give(dg->dev[k], dg->dev_size[k]); dg->dev.clear(); dg->dev_size.clear(); return rc; };
    TailParams P{};
    P.index = index->dev; P.n = n;
    const uint32_t per_cu = 512;
    const uint32_t threads = (uint32_t)std::min<uint64_t>(n, (uint64_t)std::max(1, be->compute_units()) * per_cu);
    vgk_tail_problem* d_probs = (vgk_tail_problem*)ctx->ensure_scratch(50, sizeof(vgk_tail_problem) * (size_t)(n + 1));
    vgk_tail_result* d_res = (vgk_tail_result*)ctx->ensure_scratch(51, sizeof(vgk_tail_result) * (size_t)(n + 1));
    uint32_t* d_counts = (uint32_t*)ctx->ensure_scratch(52, sizeof(uint32_t) * 2 * (size_t)(n + 1));
    TScratch* d_scratch = (TScratch*)ctx->ensure_scratch(53, sizeof(TScratch) * (size_t)std::max(1u, threads));
    if (!d_probs || !d_res || !d_counts || !d_scratch) return VGK_ENOMEM;
    uint32_t* d_first = d_counts + (n + 1);
    int rc = VGK_OK;
    be->watch(0);
    if (n) rc = be->upload(d_probs, problems, sizeof(vgk_tail_problem) * (size_t)n);
    if (!rc) rc = be->zero(d_counts, sizeof(uint32_t) * 2 * (size_t)(n + 1));
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
    // per-call tables of the construction: trim, has_pred, store, slow
    uint32_t* d_tmp = (uint32_t*)ctx->ensure_scratch(54, sizeof(uint32_t) * 4 * (N + 1));
    if (!d_parent || !d_node || !d_len || !d_col || !d_po || !d_slot || !d_tmp) return fail(VGK_ENOMEM);
    uint32_t* d_trim = d_tmp; uint32_t* d_hasp = d_tmp + (N + 1); uint32_t* d_store = d_tmp + 2 * (N + 1); uint32_t* d_slow = d_tmp + 3 * (N + 1);
    rc = be->zero(d_tmp, sizeof(uint32_t) * 4 * (N + 1));
    if (!rc) rc = be->zero(d_len + N, sizeof(uint32_t));
    P.parent = d_parent; P.node = d_node; P.len = d_len; P.trim = d_trim; P.pass = 2;
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
    be->watch(1);
    if (!rc && n) rc = be->download(results, d_res, sizeof(vgk_tail_result) * (size_t)n);      // synchronises
    else if (!rc) rc = be->sync();
    if (rc) return fail(rc);
    ctx->tail_ms = be->watch_ms();
    dg->g.col = d_col; dg->g.info = d_info; dg->g.pred_off = d_po; dg->g.pred_idx = d_pi; dg->g.slot = d_slot;
    dg->g.n_nodes = (uint32_t)N; dg->g.n_cols = tot[0];
    fo->n_nodes = N; fo->parent = d_parent; fo->node = d_node; fo->len = d_len;
    fo->graph = dg.release();
    *out = fo.release();
    return VGK_OK;
}

uint64_t vgk_forest_size(const vgk_forest* f) { return f ? f->n_nodes : 0; }

int vgk_forest_fetch(const vgk_forest* f, int32_t* parent, uint32_t* node, uint32_t* length) {
    if (!f) return VGK_EINVAL;
    if (!f->n_nodes) return VGK_OK;
    std::lock_guard<std::mutex> lk(f->ctx->mu);
    Backend* be = f->ctx->be.get();
    int rc = VGK_OK;
    if (parent) rc = be->download(parent, f->parent, sizeof(int32_t) * f->n_nodes);
    if (!rc && node) rc = be->download(node, f->node, sizeof(uint32_t) * f->n_nodes);
    if (!rc && length) rc = be->download(length, f->len, sizeof(uint32_t) * f->n_nodes);
    return rc;
}

const vgk_dgraph* vgk_forest_graph(const vgk_forest* f) { return f ? f->graph : nullptr; }

void vgk_forest_destroy(vgk_forest* f) {
    if (!f) return;
    if (f->graph) vgk_graph_destroy(f->graph);
    delete f;
}

double vgk_tail_last_ms(vgk_ctx* ctx) { return ctx ? ctx->tail_ms : 0.0; }

}  // extern "C"
