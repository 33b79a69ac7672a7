// rescue_requests_api.cpp — vgk_rescue_requests: the rescue candidates of a batch of pairs and their requests, from the extension sets the last
// vgk_gapless_extend(_seeded) call left in HBM (rescue_requests_device.hpp; MinimizerMapper::map_paired / attempt_rescue, reference
// src/minimizer_mapper.cpp:1793-1901, :3264-3348).  The host's share: two 4-byte totals and the table down (40 B per rescued pair).
#include <mutex>
#include <new>
#include <vector>
#include "ctx.hpp"
#include "dgraph.hpp"

using namespace vgk;

extern "C" int vgk_rescue_requests(vgk_ctx* ctx, const vgk_dgraph* graph, double fragment_mean, double fragment_sd, double rescue_stdevs,
                                   vgk_rescue_request* requests, size_t cap, size_t* written) try {
    if (written) *written = 0;
    if (!ctx || !graph || !vgk_tables_usable(graph->ctx, ctx) || !(fragment_sd >= 0.0) || !(rescue_stdevs >= 0.0)) return VGK_EINVAL;
    if (!graph->g.col || !graph->g.n_nodes) return VGK_EINVAL;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> stage(ctx->stage_mu);                     // the sets stay this call's
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->sets.valid || (ctx->sets.n & 1u)) return VGK_EINVAL;        // pairs: an even number of reads
    const uint32_t n_pairs = ctx->sets.n / 2;
    if (!n_pairs) return VGK_OK;
    std::vector<vgk_ctx::Pooled> temp;
    auto take = [&](uint64_t bytes) -> void* { uint64_t got = 0; void* p = ctx->dev_take(bytes ? bytes : 16, got); if (p) temp.push_back({p, got}); return p; };
    auto done = [&](int rc) { be->sync(); for (auto& q : temp) ctx->dev_give(q.p, q.bytes); return rc; };
    RqParams P{};
    P.probs = (const GProb*)ctx->sets.probs; P.res = (const vgk_gapless_result*)ctx->sets.res; P.ext = (const vgk_extension*)ctx->sets.ext; P.nodes = ctx->sets.nodes;
    P.col = graph->g.col; P.n_nodes = graph->g.n_nodes; P.n_pairs = n_pairs;
    P.mean_plus = fragment_mean + rescue_stdevs * fragment_sd; P.mean_minus = fragment_mean - rescue_stdevs * fragment_sd;
    uint32_t* tab = (uint32_t*)take(sizeof(uint32_t) * 2 * ((size_t)n_pairs + 1));
    if (!tab) return done(VGK_ENOMEM);
    P.flag = tab; P.slot = tab + n_pairs + 1;
    int rc = be->run_rescue_requests(P, RQ_FLAG);
    if (!rc) rc = be->scan_u32(P.flag, tab + n_pairs + 1, n_pairs + 1);
    uint32_t m = 0;
    if (!rc) rc = be->download(&m, P.slot + n_pairs, sizeof m);
    if (rc) return done(rc);
    if (written) *written = m;
    if (!m) return done(VGK_OK);
    if (m > cap || !requests) return done(VGK_EOPS);
    P.out = (vgk_rescue_request*)take(sizeof(vgk_rescue_request) * (size_t)m);
    if (!P.out) return done(VGK_ENOMEM);
    rc = be->run_rescue_requests(P, RQ_EMIT);
    if (!rc) rc = be->download(requests, P.out, sizeof(vgk_rescue_request) * (size_t)m);
    return done(rc);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
