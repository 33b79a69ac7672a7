// rescue_requests_device.hpp — which mates of a batch of pairs are rescued, from where, with which seed: one lane per pair over the extension
// sets the last vgk_gapless_extend(_seeded) call left in HBM (reads 2 i and 2 i + 1 are a pair).
//
// MinimizerMapper::map_paired decides it per pair (reference src/minimizer_mapper.cpp:1793-1901: a pair with alignments for one end only is a
// rescue candidate); attempt_rescue (:3264-3440) finds the rescue nodes — subgraph_in_distance_range over the SnarlDistanceIndex, an absent
// dependency: here, as in vg_amd/host/rescue_requests.cpp (the host statement of the same table, which the tests hold this one against), the
// nodes whose columns lie at the fragment's distance from the mapped mate on a graph whose node order is topological [stand-in, DESIGN.md] —
// and takes the best gapless extension of the lost mate inside them as dozeu's seed (:3322-3348).
//
// A rescued pair costs a dozen dependent cache misses (the mapped mate's first extension, its first node, two searches of the column table, the
// lost mate's extensions and their paths) and nothing else: on a host thread that is a microsecond per pair, one after the other; here the
// pairs of a batch miss together.  Two per-pair stages around a prefix sum keep the table in pair order.
#pragma once
#include <cstdint>
#include "../../include/vgk.h"
#include "gapless_device.hpp"

namespace vgk {

struct RqParams {
    const GProb* probs; const vgk_gapless_result* res; const vgk_extension* ext; const uint32_t* nodes;      // the sets in HBM (vgk_ctx::sets)
    const uint32_t* col; uint32_t n_nodes;      // first column of node v; col[n_nodes] = all bases (WinGraph::col of the resident graph)
    uint32_t n_pairs;
    double mean_plus, mean_minus;               // mean + k sd, mean - k sd
    uint32_t* flag;                             // [n_pairs + 1] 1 = exactly one mate with a full-length extension set (flag[n_pairs] = 0)
    const uint32_t* slot;                       // [n_pairs + 1] exclusive prefix sums of flag
    vgk_rescue_request* out;
};
enum { RQ_FLAG = 0, RQ_EMIT = 1 };

VGK_HD bool rq_full(const RqParams& P, uint32_t i) { return P.res[i].status == 0 && P.res[i].full_length != 0; }
// the first node whose first column lies beyond x (numpy searchsorted(col, x, side = "right") over col[0 .. n_nodes])
VGK_HD uint32_t rq_upper(const RqParams& P, double x) {
    uint32_t lo = 0, hi = P.n_nodes + 1;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (x < (double)P.col[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}
VGK_HD int64_t rq_olen(const RqParams& P, uint32_t oriented) { return (int64_t)P.col[(oriented >> 1) + 1] - (int64_t)P.col[oriented >> 1]; }

VGK_HD void rq_flag_one(const RqParams& P, uint32_t p) {
    P.flag[p] = p < P.n_pairs && rq_full(P, 2 * p) != rq_full(P, 2 * p + 1) ? 1u : 0u;
}

VGK_HD void rq_emit_one(const RqParams& P, uint32_t p) {
    if (P.slot[p + 1] == P.slot[p]) return;
    vgk_rescue_request rq;
    const uint32_t mapped = rq_full(P, 2 * p) ? 2 * p : 2 * p + 1, lost = mapped ^ 1u;
    const int64_t L = (int64_t)P.probs[lost].read_len;
    rq.mapped = mapped; rq.lost = lost;
    const vgk_extension e0 = P.ext[P.res[mapped].ext_begin];
    const uint32_t first = P.nodes[e0.path_begin];
    const bool fwd = (first & 1u) == 0;
    // a forward-mapped mate starting at column s: its partner lies downstream on the other strand, within [s + mean - k sd - L, s + (mean + k sd) 1.1 + 40];
    // a reverse-mapped mate ending at column e: upstream on the forward strand
    const double lo_d = P.mean_minus - (double)L > 0.0 ? P.mean_minus - (double)L : 0.0, hi_d = P.mean_plus * 1.1 + 40.0;
    const double s_col = (double)((int64_t)P.col[first >> 1] + (int64_t)e0.offset), e_col = (double)((int64_t)P.col[(first >> 1) + 1] - (int64_t)e0.offset);
    const double c_lo = fwd ? s_col + lo_d : e_col - hi_d, c_hi = fwd ? s_col + hi_d : e_col - lo_d;
    const double last_col = (double)((int64_t)P.col[P.n_nodes] - 1);
    int64_t node_lo = (int64_t)rq_upper(P, c_lo > 0.0 ? c_lo : 0.0) - 1;
    if (node_lo < 0) node_lo = 0;
    if (node_lo > (int64_t)P.n_nodes - 1) node_lo = (int64_t)P.n_nodes - 1;
    int64_t node_hi = (int64_t)rq_upper(P, c_hi < last_col ? c_hi : last_col);
    if (node_hi < 1) node_hi = 1;
    if (node_hi > (int64_t)P.n_nodes) node_hi = (int64_t)P.n_nodes;
    rq.node_lo = (uint32_t)node_lo; rq.node_hi = (uint32_t)node_hi;
    rq.seed_begin = 0; rq.seed_end = 0; rq.seed_node = -1; rq.seed_offset = 0;
    // the mate reads along the FORWARD strand of that subgraph: reverse-complemented when its partner maps forward
    const bool rc = fwd;
    rq.reverse = rc ? 1u : 0u; rq.reserved = 0;
    // dozeu's seed: the best extension of the lost mate inside the subgraph on the strand it is rescued on (best score, the earlier among equals)
    const uint32_t ne = P.res[lost].status == 0 ? P.res[lost].n_ext : 0u;
    bool have = false; uint32_t best = 0; int32_t best_score = 0;
    for (uint32_t x = 0; x < ne; ++x) {
        const uint32_t at = P.res[lost].ext_begin + x;
        const uint32_t pb = P.ext[at].path_begin, pl = P.ext[at].path_len;
        if (!pl) continue;
        const uint32_t* pn = P.nodes + pb;
        if (((pn[0] & 1u) == 1u) != rc) continue;
        uint32_t pmin = pn[0] >> 1, pmax = pn[0] >> 1;
        for (uint32_t t = 1; t < pl; ++t) { const uint32_t v = pn[t] >> 1; if (v < pmin) pmin = v; if (v > pmax) pmax = v; }
        if ((int64_t)pmin < node_lo || (int64_t)pmax >= node_hi) continue;
        const int32_t sc = P.ext[at].score;
        if (!have || sc > best_score) { have = true; best = at; best_score = sc; }
    }
    if (have) {
        const vgk_extension e = P.ext[best]; const uint32_t* pn = P.nodes + e.path_begin;
        int64_t path_bases = 0;
        for (uint32_t t = 0; t < e.path_len; ++t) path_bases += rq_olen(P, pn[t]);
        const uint32_t last_o = pn[e.path_len - 1], first_o = pn[0];
        const int64_t lastlen = rq_olen(P, last_o), matched = (int64_t)e.read_end - (int64_t)e.read_begin, off = (int64_t)e.offset;
        // seen from the forward strand (a mate rescued as its reverse complement): the path backwards, the read interval mirrored, the offset
        // counted from the last node's other end
        const int64_t end_in_last = e.path_len > 1 ? matched - (path_bases - off - lastlen) : off + matched;
        rq.seed_begin = (int32_t)(rc ? L - (int64_t)e.read_end : (int64_t)e.read_begin); rq.seed_end = (int32_t)(rc ? L - (int64_t)e.read_begin : (int64_t)e.read_end);
        rq.seed_node = (int32_t)(rc ? (last_o >> 1) : (first_o >> 1)); rq.seed_offset = (int32_t)(rc ? lastlen - end_in_last : off);
    }
    P.out[P.slot[p]] = rq;
}

}  // namespace vgk
