// banded_multi_device.hpp — the k-best banded global alignments of vgk_banded_align_multi enumerated ON THE DEVICE
// (Aligner::align_global_banded_multi -> BandedGlobalAligner::traceback with an AltTracebackStack, reference
// src/banded_global_aligner.cpp:2329-2423, :2426-2790; traceback over a node / over an edge :756-1780).
//
// The fill kernel keeps every cell's M / Ic / Ir (banded_device.hpp: BandedParams::scores).  Round 2 copied them to the host — 12 B per
// cell — and walked the reference's alternate-traceback stack on host threads (banded_api.cpp: MultiTracer, which states the rules and
// stays as the walker of what this kernel declines).  Here one lane per problem does the same walk over the matrices where they lie:
// a traceback is the list of its deflections; the stack keeps at most max_alt of them in descending score order, equal scores in the
// order they were found; while a traceback runs past its last deflection every other live predecessor state is proposed.
//
// A lane declines a problem (status VGK_ETOOBIG; a host thread walks it over its own matrices) when it has a chain of empty nodes from
// source to sink (the reference interleaves "empty" alignments with the stack's: :2616-2668), when a traceback needs more than
// BM_MAX_DEFL deflections, or when the empty nodes behind an edge can be walked in two ways to the same predecessor.
#pragma once
#include <stdint.h>
#include "banded_device.hpp"

namespace vgk {

constexpr uint32_t BM_MAX_DEFL = 24;
struct BmDefl { int32_t from_node, r, j, to_node, to_mat; };
struct BmTrace { int32_t score; uint32_t n_defl, start; BmDefl d[BM_MAX_DEFL]; };      // start: which candidate end node (its sink-side prefix is written behind the ops)

struct BandedMultiParams {
    BandedParams P;                   // the fill's own parameters (probs, nodes, seeds, pool, starts, codes, scores)
    uint32_t max_alt;
    BmTrace* pool;                    // (max_alt + 1) slots per problem
    uint32_t* order;                  // (max_alt + 1) slot numbers per problem: the stack, best first
    const uint32_t* sp_off; const uint32_t* sp_len;      // per candidate end node (indexed like P.starts): its empty sink-side nodes in `prefix`, sink first
    const uint32_t* prefix;
    const uint8_t* host_only;         // per problem: 1 = has empty walks (never looked at here)
    vgk_result* results;              // max_alt per problem
    uint32_t* n_alignments;
    vgk_op* ops;                      // per problem max_alt windows of 2 * ops_cap elements from ops_off
    const uint64_t* ops_off;
    int32_t* status;                  // per problem: VGK_OK, VGK_ENOBAND, VGK_EINVAL, or VGK_ETOOBIG (the host walks this one)
};

struct BmWalker {
    const BandedMultiParams& Q; const BProb& pb; const int32_t* sc;
    BmTrace* pool; uint32_t* order; uint32_t n_stack, cur, cur_defl, max; unsigned long long free_slots;
    vgk_op* runs; uint32_t n_runs, runs_cap;                                   // the traceback being walked, back to front
    int32_t go, ge; int64_t L; bool too_big;

    VGK_HD const BNode& nd(int32_t v) const { return Q.P.nodes[pb.node_base + v]; }
    VGK_HD int32_t val(int32_t v, uint32_t state, int64_t r, int64_t j) const {
        const BNode& n = nd(v);
        return sc[3 * ((size_t)n.tb_off + (size_t)j * n.stride) + (size_t)state * n.stride + (size_t)(r - j - n.top)];
    }
    VGK_HD int32_t sub(int32_t v, int64_t r, int64_t j) const { return bsub(Q.P, pb, Q.P.graph[pb.graph_off + nd(v).seq_off + j], r); }
    VGK_HD BmTrace& at(uint32_t k) { return pool[order[k]]; }
    VGK_HD void emit(int32_t node, uint32_t op, uint32_t inc) {                // BABuilder (:44-100)
        if (n_runs && runs[n_runs - 1].node == (uint32_t)node) {
            vgk_op& c = runs[n_runs - 1];
            if (c.op == op) { c.len = (uint16_t)(c.len + inc); return; }
            if (c.len == 0 && nd(node).len == 0) { c.op = (uint8_t)op; c.len = (uint16_t)inc; return; }
        }
        if (n_runs >= runs_cap) { too_big = true; return; }
        vgk_op c{}; c.node = (uint32_t)node; c.op = (uint8_t)op; c.len = (uint16_t)inc; runs[n_runs++] = c;
    }
    static VGK_HD uint32_t op_of(uint32_t mat) { return mat == BM ? VGK_OP_M : mat == BIR ? VGK_OP_I : VGK_OP_D; }
    // insert_traceback (:2691-2740): `base`'s deflections (n_base of them) and one more
    VGK_HD void insert(const BmTrace* base, uint32_t n_base, int32_t score, const BmDefl& last, uint32_t start) {
        uint32_t pos = n_stack;
        while (pos > 0 && score > at(pos - 1).score) --pos;
        if (n_stack && pos == n_stack && n_stack >= max) return;
        if (n_base + 1 > BM_MAX_DEFL) { too_big = true; return; }
        const uint32_t s = (uint32_t)__builtin_ctzll(free_slots); free_slots &= free_slots - 1;
        BmTrace& t = pool[s];
        for (uint32_t k = 0; k < n_base; ++k) t.d[k] = base->d[k];
        t.d[n_base] = last; t.n_defl = n_base + 1; t.score = score; t.start = start;
        for (uint32_t k = n_stack; k > pos; --k) order[k] = order[k - 1];
        order[pos] = s; ++n_stack;
        if (n_stack > max) { --n_stack; free_slots |= 1ull << order[n_stack]; }
    }
    VGK_HD void propose(int32_t alt, int32_t from_node, int64_t r, int64_t j, int32_t to_node, uint32_t to_mat) {      // propose_deflection (:2671-2689)
        if (cur_defl != at(cur).n_defl) return;
        if (alt <= at(n_stack - 1).score && n_stack >= max) return;
        const BmTrace& c = at(cur);
        insert(&c, c.n_defl, alt, BmDefl{from_node, (int32_t)r, (int32_t)j, to_node, (int32_t)to_mat}, c.start);
    }
    VGK_HD bool at_deflection(int32_t node, int64_t r, int64_t j) {
        const BmTrace& c = at(cur);
        return cur_defl < c.n_defl && c.d[cur_defl].from_node == node && c.d[cur_defl].r == r && c.d[cur_defl].j == j;
    }
    // the three predecessor states in the reference's order; the first that explains `cur_val` is taken, every other live one proposed
    VGK_HD int pick(int32_t v, int64_t r, int64_t j, int32_t cur_val, int32_t dm, int32_t dc, int32_t dr, int32_t from_node, int64_t fr, int64_t fj) {
        const int32_t S = at(cur).score;
        int found = -1;
        { const int32_t src = val(v, BM, r, j), diff = cur_val - (src + dm);
          if (diff == 0) found = BM; else if (blive(src)) propose(S - diff, from_node, fr, fj, from_node, BM); }
        { const int32_t src = val(v, BIC, r, j); if (blive(src)) { const int32_t diff = cur_val - (src + dc);
          if (found < 0 && diff == 0) found = BIC; else propose(S - diff, from_node, fr, fj, from_node, BIC); } }
        { const int32_t src = val(v, BIR, r, j); if (blive(src)) { const int32_t diff = cur_val - (src + dr);
          if (found < 0 && diff == 0) found = BIR; else propose(S - diff, from_node, fr, fj, from_node, BIR); } }
        return found;
    }
    // one traceback (BAMatrix::traceback :756-1126 + traceback_over_edge :1129-1780), following the current trace's deflections
    VGK_HD int trace() {
        const int32_t S = at(cur).score;
        int32_t node = at(cur).d[0].from_node; uint32_t mat = (uint32_t)at(cur).d[0].to_mat;
        int64_t r = L - 1, j = nd(node).len - 1;
        bool lead = false;
        for (uint32_t guard = 0;; ++guard) {
            if (guard > 4u * runs_cap + 64u || too_big) return too_big ? VGK_ETOOBIG : VGK_EINVAL;
            const BNode& n = nd(node);
            while ((j > 0 || mat == BIR) && !lead && !too_big) {
                emit(node, op_of(mat), 1);
                if (at_deflection(node, r, j)) {
                    if (mat == BM) { --r; --j; } else if (mat == BIR) --r; else --j;
                    mat = (uint32_t)at(cur).d[cur_defl++].to_mat;
                    continue;
                }
                if (mat == BM) {
                    if (r == 0) { mat = BIC; --j; r = -1; lead = true; break; }
                    const int32_t ms = sub(node, r, j);
                    const int src = pick(node, r - 1, j - 1, val(node, BM, r, j), ms, ms, ms, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --r; --j;
                } else if (mat == BIR) {
                    if (r == 0) { lead = true; r = -1; break; }
                    const int src = pick(node, r - 1, j, val(node, BIR, r, j), -go, -go, -ge, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --r;
                } else {
                    const int src = pick(node, r, j - 1, val(node, BIC, r, j), -go, -ge, -go, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --j;
                }
            }
            if (too_big) return VGK_ETOOBIG;
            if (lead) { mat = BIC; while (j > 0) { emit(node, VGK_OP_D, 1); --j; } }
            const BSeed* seeds = Q.P.seeds + pb.seed_base + n.seed_off;
            const uint32_t* pool_nodes = Q.P.pool + pb.pool_base;
            if (at_deflection(node, r, 0)) {                                                   // (:1158-1222)
                emit(node, op_of(mat), 1);
                const BmDefl d = at(cur).d[cur_defl++];
                // where the deflection lands: the predecessor named, through its empty nodes — this node's flattened predecessors hold
                // exactly those walks; two walks to the same predecessor would need the reference's own search order: declined
                int hit = -1;
                for (uint32_t si = 0; si < n.n_seeds; ++si) if ((int32_t)seeds[si].node == d.to_node) { if (hit >= 0) return VGK_ETOOBIG; hit = (int)si; }
                if (hit < 0) return VGK_EINVAL;
                for (uint32_t q = 0; q < seeds[hit].path_len; ++q) emit((int32_t)pool_nodes[seeds[hit].path_off + q], op_of(mat), 0);
                if (r == 0 && mat == BM) lead = true;
                if (mat == BM) --r;
                mat = (uint32_t)d.to_mat; node = d.to_node; j = nd(node).len - 1;
                continue;
            }
            int found = -1; uint32_t fmat = BM; bool flead = lead;
            if (lead) {
                emit(node, VGK_OP_D, 1);
                for (uint32_t si = 0; si < n.n_seeds; ++si) {
                    const BNode& sd = nd((int32_t)seeds[si].node);
                    const int32_t diff = (int32_t)((int64_t)ge * (sd.cum + sd.len - n.cum));
                    if (diff == 0 && found < 0) found = (int)si;
                    else propose(S - diff, node, r, 0, (int32_t)seeds[si].node, BIC);
                }
                if (found < 0) {
                    if (!n.as_source) return VGK_EINVAL;
                    for (uint32_t q = 0; q < n.src_path_len; ++q) emit((int32_t)pool_nodes[n.src_path_off + q], VGK_OP_D, 0);
                    return too_big ? VGK_ETOOBIG : VGK_OK;
                }
            } else {
                emit(node, op_of(mat), 1);
                const int32_t cur_val = val(node, mat == BM ? BM : BIC, r, 0);
                const int32_t ms = mat == BM ? sub(node, r, 0) : 0;
                for (uint32_t si = 0; si < n.n_seeds; ++si) {
                    const int32_t seed = (int32_t)seeds[si].node;
                    const BNode& sd = nd(seed);
                    const int64_t snt = sd.top + sd.len, snb = sd.bot + sd.len, sj = sd.len - 1;
                    if (r > snb - (mat == BIC ? 1 : 0) || r < snt) continue;
                    if (mat == BM && r == 0) {
                        const int32_t diff = cur_val - (-go - (sd.cum + sd.len - 1) * ge + ms);
                        if (diff == 0 && found < 0) { found = (int)si; fmat = BIC; flead = true; }
                        else propose(S - diff, node, r, 0, seed, BIC);
                        continue;
                    }
                    const int64_t sr = mat == BM ? r - 1 : r;
                    const int32_t dm = mat == BM ? ms : -go, dc = mat == BM ? ms : -ge, dr = mat == BM ? ms : -go;
                    { const int32_t src = val(seed, BM, sr, sj), diff = cur_val - (src + dm);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BM; } else if (blive(src)) propose(S - diff, node, r, 0, seed, BM); }
                    { const int32_t src = val(seed, BIC, sr, sj); if (blive(src)) { const int32_t diff = cur_val - (src + dc);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BIC; } else propose(S - diff, node, r, 0, seed, BIC); } }
                    { const int32_t src = val(seed, BIR, sr, sj); if (blive(src)) { const int32_t diff = cur_val - (src + dr);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BIR; } else propose(S - diff, node, r, 0, seed, BIR); } }
                }
                if (found < 0) {
                    if (!n.as_source) return VGK_EINVAL;
                    int64_t ins;
                    if (mat == BM) { if (cur_val != (r > 0 ? -go - (int32_t)(r - 1) * ge : 0) + ms) return VGK_EINVAL; ins = r; }
                    else           { if (cur_val != -go - (int32_t)r * ge - go) return VGK_EINVAL; ins = r + 1; }
                    for (uint32_t q = 0; q < n.src_path_len; ++q) emit((int32_t)pool_nodes[n.src_path_off + q], VGK_OP_D, 0);
                    const int32_t end_node = n.src_path_len ? (int32_t)pool_nodes[n.src_path_off + n.src_path_len - 1] : node;
                    for (int64_t q = 0; q < ins; ++q) emit(end_node, VGK_OP_I, 1);
                    return too_big ? VGK_ETOOBIG : VGK_OK;
                }
            }
            const BSeed& sr = seeds[found];
            for (uint32_t q = 0; q < sr.path_len; ++q) emit((int32_t)pool_nodes[sr.path_off + q], op_of(mat), 0);
            if (!lead) { if (mat == BM) --r; mat = fmat; lead = flead; }
            node = (int32_t)sr.node; j = nd(node).len - 1;
        }
    }
};

// one lane: every alternate of problem a (BandedGlobalAligner::traceback :2329-2423, without the empty walks — those problems are the host's)
VGK_HD void banded_multi_one(const BandedMultiParams& Q, uint32_t a) {
    const BProb& pb = Q.P.probs[a];
    Q.n_alignments[a] = 0;
    if (Q.host_only[a]) { Q.status[a] = VGK_ETOOBIG; return; }
    const uint32_t slots = Q.max_alt + 1;
    BmWalker w{Q, pb, Q.P.scores + 3 * pb.tb_base, Q.pool + (size_t)a * slots, Q.order + (size_t)a * slots, 0u, 0u, 0u, Q.max_alt,
               slots >= 64u ? ~0ull : (1ull << slots) - 1ull, nullptr, 0u, pb.ops_cap, Q.P.go, Q.P.ge, (int64_t)pb.L, false};
    const int64_t L = w.L;
    for (uint32_t c = 0; c < pb.n_starts; ++c) {
        const int32_t u = (int32_t)Q.P.starts[pb.start_base + c].node;
        const BNode& n = w.nd(u);
        const int64_t k = (L - 1) - (n.len - 1) - n.top;
        if (k < 0 || k > n.bot - n.top) continue;
        const int32_t cand[3] = { w.val(u, BM, L - 1, n.len - 1), w.val(u, BIR, L - 1, n.len - 1), w.val(u, BIC, L - 1, n.len - 1) };
        const uint32_t cmat[3] = { BM, BIR, BIC };
        for (int q = 0; q < 3; ++q) if (blive(cand[q])) w.insert(nullptr, 0, cand[q], BmDefl{u, (int32_t)(L - 1), n.len - 1, u, (int32_t)cmat[q]}, c);
    }
    if (!w.n_stack) { Q.status[a] = VGK_ENOBAND; return; }
    vgk_result* results = Q.results + (size_t)a * Q.max_alt;
    vgk_op* window = Q.ops + Q.ops_off[a];
    uint32_t n_out = 0;
    while (w.cur < w.n_stack && n_out < Q.max_alt) {
        vgk_op* out = window + (size_t)n_out * 2u * pb.ops_cap;
        // an alternate's window is two op lists long: the walk builds its runs back to front in the second, they are turned around into the first
        w.runs = out + pb.ops_cap; w.runs_cap = pb.ops_cap; w.n_runs = 0; w.cur_defl = 1;
        const int rc = w.trace();
        if (rc != VGK_OK || w.too_big) { Q.status[a] = w.too_big ? VGK_ETOOBIG : rc; return; }
        const BmTrace& t = w.at(w.cur);
        const uint32_t pre_len = Q.sp_len[pb.start_base + t.start];
        if (w.n_runs + pre_len > pb.ops_cap) { Q.status[a] = VGK_ETOOBIG; return; }
        uint32_t k = 0;
        for (uint32_t q = w.n_runs; q-- > 0;) { vgk_op o = w.runs[q]; if (o.len == 0) o.op = VGK_OP_M; out[k++] = o; }
        const uint32_t* pre = Q.prefix + Q.sp_off[pb.start_base + t.start];
        for (uint32_t e = pre_len; e-- > 0;) { vgk_op o{}; o.node = pre[e]; o.op = VGK_OP_M; o.len = 0; out[k++] = o; }
        vgk_result res{};
        res.score = t.score; res.status = VGK_OK; res.n_ops = k; res.ops_begin = (uint32_t)(Q.ops_off[a] + (uint64_t)n_out * 2u * pb.ops_cap);
        results[n_out++] = res;
        ++w.cur;
    }
    Q.status[a] = VGK_OK; Q.n_alignments[a] = n_out;
}

}  // namespace vgk
