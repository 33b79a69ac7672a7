// gssw_multi_device.hpp — the k-best pinned tracebacks of vgk_gssw_align_multi enumerated ON THE DEVICE (Aligner::align_pinned_multi,
// reference src/aligner.cpp:423-435, :455-480; gssw_graph_trace_back_pinned_multi behind it).
//
// The fill (gssw_matrix_device.hpp) leaves every cell's H / E / F in HBM.  Round 2 copied those matrices to the host (12 B per cell) and
// walked the alternates on host threads; here one lane per problem walks them where they lie and only the alignments come back.  The
// rules are the ones gssw_multi_api.cpp states at its top (the oracle restates them in oracle/vgo_multi.c): a traceback is the walk of
// the single traceback's state machine over H / E / F; the sources of a state come in a fixed order, each with a loss; an alternate
// names the sources it takes instead of the lossless first one (its deflections); alternates are walked best first, earlier proposals
// first among equals, and while an alternate runs past its last deflection every other source worth more than 0 becomes a proposal.
//
// What a lane keeps per problem (HBM, `pool` / `order`): the queue of proposals — at most max_alt_alns of them, each a score, a start
// and up to MT_MAX_DEFL deflections of 12 bytes — as slots of a small pool plus the order of the slots.  A problem whose nodes have more
// than MT_MAX_PRED predecessors, or whose alternates need more deflections than a slot holds, is answered VGK_ETOOBIG by the kernel and
// the caller walks that problem on a host thread over its own matrices (the round-2 path: same rules, same results).
#pragma once
#include <stdint.h>
#include "gssw_matrix_device.hpp"

namespace vgk {

constexpr uint32_t MT_MAX_DEFL = 24, MT_MAX_PRED = 15, MT_MAX_SRC = 2 * MT_MAX_PRED + 2;
enum { MT_H = 0, MT_E = 1, MT_F = 2 };

struct MtDefl { int32_t r, c; uint32_t st_take; };                   // st | take << 8
struct MtAlt { int32_t score; uint32_t start, n_defl; MtDefl d[MT_MAX_DEFL]; };
struct MtSource { int32_t value, st, r, c, v; };                     // st < 0: the walk ends after this step; v = node of column c

struct GsswMultiParams {
    GsswMatrixParams M;               // the fill's own parameters: probs (with the fill's status), arenas, cells
    const uint8_t* pinning;           // per node, indexed like M.nodes
    uint32_t max_alt;
    MtAlt* pool;                      // (max_alt + 2) slots per problem
    uint32_t* order;                  // (max_alt + 2) slot numbers per problem: the queue, best first
    vgk_result* results;              // max_alt per problem; unused ones stay zero
    uint32_t* n_alignments;
    vgk_op* ops;                      // per problem a window of max_alt * (L + R + 2) elements from ops_off[i]
    const uint64_t* ops_off;
    int32_t* status;                  // per problem: VGK_OK, the fill's failure, or VGK_ETOOBIG (the host walks this one)
};

struct MtWalker {
    const GsswMultiParams& P; const MProb& pb;
    const int32_t *H, *E, *F; const uint8_t *rd, *ql, *gr; const MNode* nodes;
    int32_t L, go, ge;
    MtAlt* pool; uint32_t* order; uint32_t qn;                        // the queue: order[0 .. qn) are slots of `pool`
    uint32_t free_mask_lo, free_mask_hi;                               // free slots (max_alt + 2 <= 64)
    bool too_big;

    VGK_HD int32_t h(int32_t c, int32_t r) const { return H[(uint64_t)c * L + r]; }
    VGK_HD int32_t e(int32_t c, int32_t r) const { return E[(uint64_t)c * L + r]; }
    VGK_HD int32_t f(int32_t c, int32_t r) const { return F[(uint64_t)c * L + r]; }
    VGK_HD int32_t score(int32_t r, int32_t c) const {
        const uint32_t ref = gr[c], q = rd[r];
        return (int32_t)(ql ? P.M.mat[25u * ql[r] + 5u * ref + q] : P.M.mat[5u * ref + q]) + (r == 0 ? pb.start_bonus : 0);
    }
    VGK_HD uint32_t n_pred_cols(int32_t c, int32_t v) const { return (uint32_t)c != nodes[v].col_start ? 1u : nodes[v].n_pred; }
    VGK_HD void pred_col(int32_t c, int32_t v, uint32_t k, int32_t& qc, int32_t& qv) const {
        if ((uint32_t)c != nodes[v].col_start) { qc = c - 1; qv = v; return; }
        qv = (int32_t)P.M.preds[nodes[v].pred_begin + k]; qc = (int32_t)nodes[qv].col_end - 1;
    }
    // the sources of a state in their fixed order; n_diag = how many of them are diagonal steps (H states)
    VGK_HD uint32_t sources(int32_t st, int32_t r, int32_t c, int32_t v, bool no_e, bool no_f, MtSource* out, uint32_t& n_diag) const {
        const uint32_t np = n_pred_cols(c, v);
        uint32_t n = 0; n_diag = 0;
        if (st == MT_H) {
            const int32_t s = score(r, c);
            if (r == 0 || np == 0) out[n++] = MtSource{s, -1, r - 1, c, v};
            else {
                bool zero_seen = false;                              // predecessors whose cell is worth 0 all mean "the alignment starts here": one source
                for (uint32_t k = 0; k < np; ++k) {
                    int32_t qc, qv; pred_col(c, v, k, qc, qv);
                    const int32_t d = h(qc, r - 1);
                    if (d == 0) { if (zero_seen) continue; zero_seen = true; }
                    out[n++] = MtSource{d + s, MT_H, r - 1, qc, qv};
                }
            }
            n_diag = n;
            out[n++] = MtSource{no_e ? 0 : e(c, r), MT_E, r, c, v};
            out[n++] = MtSource{no_f ? 0 : f(c, r), MT_F, r, c, v};
        } else if (st == MT_E) {
            for (uint32_t k = 0; k < np; ++k) {
                int32_t qc, qv; pred_col(c, v, k, qc, qv);
                out[n++] = MtSource{h(qc, r) - go, MT_H, r, qc, qv};
                out[n++] = MtSource{e(qc, r) - ge, MT_E, r, qc, qv};
            }
        } else if (r > 0) {
            out[n++] = MtSource{h(c, r - 1) - go, MT_H, r - 1, c, v};
            out[n++] = MtSource{f(c, r - 1) - ge, MT_F, r - 1, c, v};
        }
        return n;
    }
    VGK_HD bool explains(int32_t r, int32_t c, int32_t v, bool no_e, bool no_f) const {     // can H(r, c) be left without entering E (F)?
        const int32_t value = h(c, r);
        if (value == 0) return true;
        MtSource src[MT_MAX_SRC]; uint32_t nd; const uint32_t n = sources(MT_H, r, c, v, no_e, no_f, src, nd);
        for (uint32_t k = 0; k < n; ++k) if (src[k].value == value) return true;
        return false;
    }

    VGK_HD uint32_t slot_take() {
        if (free_mask_lo) { const uint32_t b = (uint32_t)__builtin_ctz(free_mask_lo); free_mask_lo &= free_mask_lo - 1; return b; }
        const uint32_t b = (uint32_t)__builtin_ctz(free_mask_hi); free_mask_hi &= free_mask_hi - 1; return 32u + b;
    }
    VGK_HD void slot_give(uint32_t s) { if (s < 32u) free_mask_lo |= 1u << s; else free_mask_hi |= 1u << (s - 32u); }

    // a proposal = `base`'s deflections (n_base of them) and one more; kept only if it ranks among the first `room` waiting ones
    VGK_HD void offer(int32_t score2, uint32_t start, const MtAlt* base, uint32_t n_base, bool extra, MtDefl more, uint32_t room) {
        if (!room) return;
        uint32_t at = qn;
        while (at > 0 && pool[order[at - 1]].score < score2) --at;
        if (at >= room) return;
        if (n_base + (extra ? 1u : 0u) > MT_MAX_DEFL) { too_big = true; return; }
        const uint32_t s = slot_take();
        MtAlt& a = pool[s];
        a.score = score2; a.start = start; a.n_defl = n_base + (extra ? 1u : 0u);
        for (uint32_t k = 0; k < n_base; ++k) a.d[k] = base->d[k];
        if (extra) a.d[n_base] = more;
        for (uint32_t k = qn; k > at; --k) order[k] = order[k - 1];
        order[at] = s; ++qn;
        if (qn > room) { --qn; slot_give(order[qn]); }
    }
};

// one lane: every alternate of problem i
VGK_HD void gssw_multi_one(const GsswMultiParams& P, uint32_t i) {
    const MProb& pb = P.M.probs[i];
    P.n_alignments[i] = 0;
    if (pb.status != VGK_OK) { P.status[i] = pb.status; return; }
    const uint64_t plane = (uint64_t)pb.R * pb.L;
    const uint32_t slots = P.max_alt + 2;
    MtWalker w{P, pb, P.M.cells + pb.mat_off, P.M.cells + pb.mat_off + plane, P.M.cells + pb.mat_off + 2 * plane,
               P.M.reads + pb.read_off, P.M.quals ? P.M.quals + pb.read_off : nullptr, P.M.graph + pb.graph_off, P.M.nodes + pb.node_off,
               (int32_t)pb.L, P.M.go, P.M.ge, P.pool + (uint64_t)i * slots, P.order + (uint64_t)i * slots, 0u,
               slots >= 32u ? 0xffffffffu : (1u << slots) - 1u, slots > 32u ? (slots >= 64u ? 0xffffffffu : (1u << (slots - 32u)) - 1u) : 0u, false};
    const uint8_t* pin = P.pinning + pb.node_off;
    const int32_t L = w.L;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) if (w.nodes[v].n_pred > MT_MAX_PRED) { P.status[i] = VGK_ETOOBIG; return; }

    // the starts: the last column of every pinning node, numbered in node order
    uint32_t n_start = 0;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        if (!pin[v]) continue;
        const int32_t val = w.h((int32_t)w.nodes[v].col_end - 1, L - 1);
        if (val > 0) w.offer(val, n_start, nullptr, 0, false, MtDefl{0, 0, 0}, P.max_alt);
        ++n_start;
    }
    vgk_result* results = P.results + (uint64_t)i * P.max_alt;
    vgk_op* ops = P.ops + P.ops_off[i];
    uint32_t n_out = 0, cursor = 0;
    while (w.qn && n_out < P.max_alt && !w.too_big) {
        const uint32_t cur = w.order[0];
        for (uint32_t k = 1; k < w.qn; ++k) w.order[k - 1] = w.order[k];
        --w.qn;
        const MtAlt& alt = w.pool[cur];
        const uint32_t room = P.max_alt - n_out - 1;
        // the start column of this alternate
        int32_t v = 0, c = 0;
        { uint32_t k = 0; for (uint32_t u = 0; u < pb.n_nodes; ++u) { if (!pin[u]) continue; if (k == alt.start) { v = (int32_t)u; c = (int32_t)w.nodes[u].col_end - 1; break; } ++k; } }
        int32_t st = MT_H, r = L - 1, first_c = c, first_v = v;
        bool no_e = false, no_f = false;
        uint32_t next = 0; int32_t lost = 0;
        const int32_t start_value = w.h(c, r);
        vgk_result res{};
        res.score = alt.score; res.status = VGK_OK; res.end_node = v; res.end_offset = c - (int32_t)w.nodes[v].col_start; res.end_read = r;
        const uint32_t begin = cursor;
#define MT_PUSH(NODE, OP, LEN) do { \
        if (cursor > begin && ops[cursor - 1].node == (uint32_t)(NODE) && ops[cursor - 1].op == (uint8_t)(OP)) ops[cursor - 1].len = (uint16_t)(ops[cursor - 1].len + (LEN)); \
        else { vgk_op x{}; x.node = (uint32_t)(NODE); x.op = (uint8_t)(OP); x.len = (uint16_t)(LEN); ops[cursor++] = x; } } while (0)
        MtSource src[MT_MAX_SRC]; uint32_t n_diag = 0;
        for (;;) {
            const int32_t value = st == MT_H ? w.h(c, r) : st == MT_E ? w.e(c, r) : w.f(c, r);
            if (st == MT_H && value == 0) break;
            uint32_t n_src = w.sources(st, r, c, v, no_e, no_f, src, n_diag);
            uint32_t take = n_src;
            const bool here = next < alt.n_defl && (int32_t)(alt.d[next].st_take & 0xffu) == st && alt.d[next].r == r && alt.d[next].c == c;
            if (here) take = alt.d[next++].st_take >> 8;
            else {
                for (uint32_t k = 0; k < n_src; ++k) if (src[k].value == value) { take = k; break; }
                if (take >= n_src && (no_e || no_f)) {           // gap_open == gap_extend: like the single traceback, re-open the gap
                    no_e = no_f = false;
                    n_src = w.sources(st, r, c, v, false, false, src, n_diag);
                    for (uint32_t k = 0; k < n_src; ++k) if (src[k].value == value) { take = k; break; }
                }
            }
            if (take >= n_src) break;
            if (next == alt.n_defl && !here) {
                for (uint32_t k = 0; k < n_src; ++k) {
                    if (k == take || src[k].value <= 0) continue;
                    if (st != MT_H && src[k].st == MT_H && !w.explains(src[k].r, src[k].c, src[k].v, st == MT_E, st == MT_F)) continue;
                    const int32_t score2 = start_value - lost - (value - src[k].value);
                    if (score2 <= 0) continue;
                    w.offer(score2, alt.start, &alt, alt.n_defl, true, MtDefl{r, c, (uint32_t)st | (k << 8)}, room);
                }
            }
            lost += value - src[take].value;
            if (st == MT_H && take < n_diag) {
                MT_PUSH(v, VGK_OP_M, 1); first_c = c; first_v = v; no_e = no_f = false;
                if (src[take].st < 0) { r -= 1; break; }
                r = src[take].r; c = src[take].c; v = src[take].v;
            } else if (st == MT_H) st = src[take].st;
            else if (st == MT_E) { MT_PUSH(v, VGK_OP_D, 1); first_c = c; first_v = v; no_e = src[take].st == MT_H; no_f = false; st = src[take].st; c = src[take].c; v = src[take].v; }
            else { MT_PUSH(v, VGK_OP_I, 1); no_f = src[take].st == MT_H; no_e = false; st = src[take].st; r = src[take].r; }
        }
        if (r >= 0) MT_PUSH(first_v, VGK_OP_S, (uint32_t)r + 1);
#undef MT_PUSH
        for (uint32_t a = begin, b = cursor; a + 1 < b; ++a, --b) { const vgk_op t = ops[a]; ops[a] = ops[b - 1]; ops[b - 1] = t; }     // found back to front
        res.ops_begin = (uint32_t)(P.ops_off[i] + begin); res.n_ops = cursor - begin;
        res.first_offset = first_c - (int32_t)w.nodes[first_v].col_start;
        results[n_out++] = res;
        w.slot_give(cur);
    }
    if (w.too_big) { P.status[i] = VGK_ETOOBIG; return; }
    P.status[i] = VGK_OK; P.n_alignments[i] = n_out;
}

}  // namespace vgk
