/*
 * vgo_xdrop.c — CPU ORACLE (test infrastructure, NOT product code) for the pinned
 * X-drop extension that vg runs through the third-party library dozeu
 * (DozeuInterface::align_pinned -> align_downward -> do_poa -> dz_extend / dz_trace;
 * reference: src/dozeu_interface.cpp:210-307, 687-766; src/xdrop_aligner.cpp:71-114).
 *
 * PARITY STATUS.  dozeu itself (`vgteam/dozeu`, deps/dozeu, .gitmodules:61-63) is an
 * UN-VENDORED submodule, empty in the reference snapshot, no recoverable revision.
 * This file restates its published algorithm (Suzuki's X-drop DP: semi-global
 * affine-gap extension from a root column, int16 cells, gap parameterised as
 * (open - extend, extend), full-length bonus on consuming the last packed query
 * base) and is anchored on the reference's call sites and on the known-answer tests
 * of src/unittest/xdrop_aligner.cpp (tests/golden/ref_xdrop_aligner.json).
 *
 * What is restated exactly:
 *   - root column from dz_align_init(dz, max_gap): H(0 query bases) = 0,
 *     H(i) = -(go + (i-1) ge) for i <= max_gap rounded up to dozeu's 8-cell vector,
 *     nothing beyond (src/dozeu_interface.cpp:226; max_gap clamped to >= 1, src/aligner.cpp:638)
 *   - every tip in the pin direction is a seed with the same query offset (:738-755)
 *   - node loop in the caller's topological order, incoming fronts merged by
 *     element-wise max in follow_edges order (:261-283)
 *   - best front = first node (in order) with strictly greater max (:286-290)
 *   - score 0  => full-length insertion at the head (:344-359), done by the host shim
 *   - E/F/H affine recurrence without a zero floor; bonus added on the diagonal
 *     move that consumes the last query base
 * DELIBERATE DIFFERENCE [PARITY-UNPINNED]: dozeu drops band-end vectors whose cells are
 * more than xt = go - ge + ge*max_gap below the running maximum; the band evolution
 * rules live only in the missing source.  This oracle (and the HIP engine) keep every
 * cell, i.e. they return the exact semi-global optimum, which is what dozeu returns
 * whenever its band contains the optimal path (the case in every reference unit test,
 * incl. "...would be x-dropped if not for the full length bonus", xdrop_aligner.cpp:819-837).
 * Also unpinned: N scores 0 against everything (as in vg's 5x5 matrix); best cell inside a
 * node = first column, then smallest query position; traceback preference
 * diagonal > deletion > insertion, gap open before extend, first explaining predecessor.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"

#define NEG (-(1 << 28))

static inline int nt_read(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static inline int nt_ref(char ch) {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

#define IDX(c, i) ((size_t)(c) * (size_t)(L + 1) + (size_t)(i))

/* Left-pinned extension; rows i = 0..L count consumed query bases. */
int vgo_xdrop_pinned_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                             vgk_result* res, vgk_op* ops, uint32_t ops_cap);
int vgo_xdrop_pinned_align(const vgk_scoring* sc, const vgk_gssw_problem* p,
                           vgk_result* res, vgk_op* ops, uint32_t ops_cap) { return vgo_xdrop_pinned_align_q(sc, NULL, p, res, ops, ops_cap); }

/* qa != NULL: QualAdjXdropAligner (src/qual_adj_xdrop_aligner.cpp:74-135); the single bonus is the quality-adjusted
 * one of the far-end base (src/aligner.cpp:1164-1167). */
static int xdrop_core(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                      vgk_result* res, vgk_op* ops, uint32_t ops_cap, int band, uint64_t* stats);
int vgo_xdrop_pinned_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                             vgk_result* res, vgk_op* ops, uint32_t ops_cap) { return xdrop_core(sc, qa, p, res, ops, ops_cap, 0, NULL); }
/* The same with dozeu's band [PARITY-UNPINNED, see include/vgk.h vgk_xdrop_band_align]: after every column the front shrinks to the
 * 8-row vectors between the first and the last one holding a cell >= (best so far on the way here) - xt; cells outside become
 * unreachable.  stats[0] += cells inside the bands, stats[1] += (L + 1) * R. */
int vgo_xdrop_band_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                           vgk_result* res, vgk_op* ops, uint32_t ops_cap, uint64_t* stats) { return xdrop_core(sc, qa, p, res, ops, ops_cap, 1, stats); }

static int xdrop_core(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                      vgk_result* res, vgk_op* ops, uint32_t ops_cap, int band, uint64_t* stats)
{
    const int L = (int)p->read_len;
    const vgk_graph* g = &p->graph;
    const int nV = (int)g->n_nodes;
    if (qa && !p->qual) { memset(res, 0, sizeof *res); res->status = VGK_EINVAL; return VGK_EINVAL; }
    const int go = sc->gap_open, ge = sc->gap_extend;
    const int bonus = qa ? (L > 0 ? qa->bonuses[p->qual[L - 1]] : 0) : sc->full_length_bonus;
    int max_gap = (int)p->max_gap_length; if (max_gap < 1) max_gap = 1;
    const int gap_cells = (max_gap + 7) & ~7;
    const int want_tb = (p->flags & VGK_GSSW_TRACEBACK) != 0;

    memset(res, 0, sizeof *res);
    res->end_node = -1; res->end_offset = -1; res->end_read = -1;
    if (L <= 0 || nV <= 0) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    int* col0 = (int*)malloc(sizeof(int) * (size_t)(nV + 1));
    col0[0] = 0;
    for (int n = 0; n < nV; ++n) {
        if (g->node_len[n] == 0) { free(col0); res->status = VGK_EINVAL; return VGK_EINVAL; }
        if (g->node_len[n] > 65535u) { free(col0); res->status = VGK_ETOOBIG; return VGK_ETOOBIG; }   /* vgk_op.len is 16 bits */
        col0[n + 1] = col0[n] + (int)g->node_len[n];
        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k)
            if ((int)g->pred_idx[k] >= n) { free(col0); res->status = VGK_EINVAL; return VGK_EINVAL; }
    }
    const int R = col0[nV];
    int* node_of = (int*)malloc(sizeof(int) * (size_t)R);
    for (int n = 0; n < nV; ++n) for (int c = col0[n]; c < col0[n + 1]; ++c) node_of[c] = n;
    int8_t* rd = (int8_t*)malloc((size_t)L);
    for (int r = 0; r < L; ++r) rd[r] = (int8_t)nt_read(p->read[r]);
    int8_t* rf = (int8_t*)malloc((size_t)R);
    for (int c = 0; c < R; ++c) rf[c] = (int8_t)nt_ref(g->seq[c]);

    const size_t cells = (size_t)R * (size_t)(L + 1);
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * cells);
    int32_t* E = (int32_t*)malloc(sizeof(int32_t) * cells);
    int32_t* F = (int32_t*)malloc(sizeof(int32_t) * cells);
    int32_t* En = (int32_t*)malloc(sizeof(int32_t) * cells);
    int32_t* rootH = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    int32_t* rootE = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    int32_t* seedH = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    int32_t* seedE = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L + 1));
    for (int i = 0; i <= L; ++i) {
        rootH[i] = i == 0 ? 0 : (i <= gap_cells ? -(go + (i - 1) * ge) : NEG);
        rootE[i] = rootH[i] > NEG ? rootH[i] - go : NEG;      /* E of the column after the root */
    }
#define SCORE(i, c) ((int)(qa ? qa->matrix[25 * p->qual[(i) - 1] + 5 * rf[c] + rd[(i) - 1]] : sc->matrix[5 * rf[c] + rd[(i) - 1]]) + ((i) == L ? bonus : 0))

    int32_t best = 0; int best_c = -1, best_i = 0;
    const int32_t xt = (go - ge) + ge * max_gap;                 /* dozeu's x-drop threshold (dz_align_init) */
    int32_t* node_fmax = (int32_t*)malloc(sizeof(int32_t) * (size_t)nV);      /* band mode: best score on the way to the end of each node */
    uint64_t in_band = 0;
    for (int n = 0; n < nV; ++n) {
        const int npred = (int)(g->pred_off[n + 1] - g->pred_off[n]);
        int32_t fmax = 0;                                        /* the root's best: nothing consumed, score 0 */
        int front_live = 0;
        if (npred == 0) { memcpy(seedH, rootH, sizeof(int32_t) * (size_t)(L + 1)); memcpy(seedE, rootE, sizeof(int32_t) * (size_t)(L + 1)); }
        else {
            fmax = NEG;
            for (int i = 0; i <= L; ++i) { seedH[i] = NEG; seedE[i] = NEG; }
            for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                const int pc = col0[g->pred_idx[k] + 1] - 1;
                for (int i = 0; i <= L; ++i) { seedH[i] = imax(seedH[i], H[IDX(pc, i)]); seedE[i] = imax(seedE[i], En[IDX(pc, i)]); }
                fmax = imax(fmax, node_fmax[g->pred_idx[k]]);
            }
        }
        for (int c = col0[n]; c < col0[n + 1]; ++c) {
            const int first = (c == col0[n]);
            int32_t colmax = NEG; int colmax_i = -1;
            for (int i = 0; i <= L; ++i) {
                const int32_t e = first ? seedE[i] : En[IDX(c - 1, i)];
                int32_t f = NEG, d = NEG;
                if (i > 0) {
                    f = imax(H[IDX(c, i - 1)] - go, F[IDX(c, i - 1)] - ge);
                    d = first ? seedH[i - 1] : H[IDX(c - 1, i - 1)];
                    if (d > NEG / 2) d += SCORE(i, c); else d = NEG;
                }
                int32_t h = imax(imax(d, e), f);
                if (h < NEG) h = NEG;
                H[IDX(c, i)] = h; E[IDX(c, i)] = e; F[IDX(c, i)] = f;
                int32_t en = imax(h - go, e - ge); if (en < NEG) en = NEG;
                En[IDX(c, i)] = en;
                if (!band && h > colmax) { colmax = h; colmax_i = i; }
            }
            if (band) {
                /* the front of this column: 8-row vectors from the first to the last one with a cell >= fmax - xt */
                const int nb = (L + 8) / 8; int sb = nb, eb = 0;
                for (int b = 0; b < nb; ++b) {
                    int alive = 0;
                    for (int i = 8 * b; i < 8 * b + 8 && i <= L; ++i) if (H[IDX(c, i)] > NEG / 2 && H[IDX(c, i)] >= fmax - xt) alive = 1;
                    if (alive) { if (b < sb) sb = b; eb = b + 1; }
                }
                for (int i = 0; i <= L; ++i) {
                    const int b = i / 8;
                    if (b < sb || b >= eb) { H[IDX(c, i)] = E[IDX(c, i)] = F[IDX(c, i)] = En[IDX(c, i)] = NEG; continue; }
                    ++in_band;
                    if (H[IDX(c, i)] > colmax) { colmax = H[IDX(c, i)]; colmax_i = i; }
                }
                fmax = imax(fmax, colmax);
                front_live = eb > sb;
            }
            if (colmax > best) { best = colmax; best_c = c; best_i = colmax_i; }
        }
        node_fmax[n] = (!band || front_live) ? fmax : NEG;        /* an empty front is not merged into its successors (src/dozeu_interface.cpp:261-269) */
    }
    free(node_fmax);
    if (stats) { stats[0] += in_band; stats[1] += (uint64_t)(L + 1) * (uint64_t)R; }

    int rc = VGK_OK;
    if (best >= 32767) { rc = VGK_EOVERFLOW; goto done; }
    res->score = best;
    if (best <= 0 || best_c < 0) { res->score = 0; goto done; }       /* the root wins: full-length insertion by the shim */
    res->end_node = node_of[best_c]; res->end_offset = best_c - col0[node_of[best_c]]; res->end_read = best_i - 1;
    if (!want_tb) goto done;
    {
        uint32_t nops = 0;
#define PUSH(NODE, OP, LEN) do { if ((LEN) > 0) { \
        if (nops > 0 && ops[nops - 1].node == (uint32_t)(NODE) && ops[nops - 1].op == (OP)) ops[nops - 1].len += (LEN); \
        else { if (nops >= ops_cap) { rc = VGK_EOPS; goto done; } \
               ops[nops].node = (uint32_t)(NODE); ops[nops].op = (uint8_t)(OP); ops[nops].len = (uint16_t)(LEN); ops[nops].pad = 0; ++nops; } } } while (0)
        int c = best_c, i = best_i; int32_t cur = best;
        PUSH(node_of[c], VGK_OP_S, L - i);
        enum { ST_H, ST_E, ST_F } st = ST_H;
        int done_walk = 0;
        while (!done_walk) {
            const int n = node_of[c];
            const int first = (c == col0[n]);
            const int npred = (int)(g->pred_off[n + 1] - g->pred_off[n]);
            if (st == ST_H) {
                int moved = 0;
                if (i > 0) {
                    int32_t d = first ? (npred == 0 ? rootH[i - 1] : NEG) : H[IDX(c - 1, i - 1)];
                    if (first && npred > 0) for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                        const int pc = col0[g->pred_idx[k] + 1] - 1; d = imax(d, H[IDX(pc, i - 1)]); }
                    if (d > NEG / 2 && cur == d + SCORE(i, c)) {
                        PUSH(n, VGK_OP_M, 1);
                        cur = d; i -= 1; moved = 1;
                        if (!first) c -= 1;
                        else if (npred == 0) { PUSH(n, VGK_OP_I, i); done_walk = 1; }   /* reached the root: leading insertion */
                        else {
                            int found = -1;
                            for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                                const int pc = col0[g->pred_idx[k] + 1] - 1; if (H[IDX(pc, i)] == cur) { found = pc; break; } }
                            if (found < 0) { rc = VGK_EINVAL; goto done; }
                            c = found;
                        }
                    }
                }
                if (!moved) {
                    if (cur == E[IDX(c, i)]) st = ST_E;
                    else if (i > 0 && cur == F[IDX(c, i)]) st = ST_F;
                    else { rc = VGK_EINVAL; goto done; }
                }
            } else if (st == ST_E) {
                PUSH(n, VGK_OP_D, 1);
                if (first && npred == 0) {           /* deletion opened straight from the root column */
                    if (rootE[i] != cur) { rc = VGK_EINVAL; goto done; }
                    PUSH(n, VGK_OP_I, i); done_walk = 1;
                } else {
                    int pc = c - 1;
                    if (first) {
                        pc = -1;
                        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                            const int q = col0[g->pred_idx[k] + 1] - 1; if (En[IDX(q, i)] == cur) { pc = q; break; } }
                        if (pc < 0) { rc = VGK_EINVAL; goto done; }
                    }
                    if (H[IDX(pc, i)] - go == cur) { st = ST_H; cur += go; } else cur += ge;
                    c = pc;
                }
            } else {
                PUSH(n, VGK_OP_I, 1);
                if (i == 0) { rc = VGK_EINVAL; goto done; }
                if (H[IDX(c, i - 1)] - go == cur) { st = ST_H; cur += go; } else cur += ge;
                i -= 1;
            }
        }
        for (uint32_t a = 0, b = nops ? nops - 1 : 0; a < b; ++a, --b) { vgk_op t = ops[a]; ops[a] = ops[b]; ops[b] = t; }
        res->n_ops = nops; res->first_offset = 0;
    }
done:
    res->status = rc;
    free(col0); free(node_of); free(rd); free(rf); free(H); free(E); free(F); free(En); free(rootH); free(rootE); free(seedH); free(seedE);
    return rc;
}
