/*
 * vgo_gssw.c — CPU ORACLE (test infrastructure, NOT product code) for the
 * graph Smith-Waterman that vg runs through the third-party library gssw.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call this.  The product path (vg_amd/csrc, libvgamd.so) never links it.
 *
 * PARITY STATUS.  vg's own arithmetic for this path lives in `vgteam/gssw`
 * (deps/gssw, .gitmodules:4-6), an UN-VENDORED submodule that is empty in the
 * reference snapshot, with no recoverable pinned revision.  This file restates
 * gssw's published algorithm (Farrar striped SW generalised to DAGs, Zhao et
 * al. SSW, vgteam/gssw `gssw_graph_fill_pinned` / `gssw_graph_trace_back`) and
 * is anchored on the reference's own call sites and known-answer tests:
 *   - call sites: src/aligner.cpp:396-402 (fill), :423-435 (pinned traceback),
 *     :537-545 (local traceback), :550-557 (score-only read-out)
 *   - golden vectors: src/unittest/aligner.cpp, src/unittest/pinned_alignment.cpp,
 *     test/t/04_vg_align.t (transcribed into tests/golden/ by
 *     tests/golden/extract_reference_tests.py)
 * Every rule that those tests do NOT pin is marked PARITY-UNPINNED below.
 *
 * Semantics restated (scores are exact integers; gssw's int8 -> int16 retry,
 * src/aligner.cpp:402 `score_size = 2`, only changes the container, so the
 * oracle computes in int32 and reports VGK_EOVERFLOW at gssw's int16 limit):
 *   columns = graph bases in topological node order, rows = read bases
 *   s(r,c)  = matrix[5*nt[ref c] + nt[read r]] + (r==0 ? start_bonus : 0)
 *                                             + (r==L-1 ? end_bonus : 0)
 *             (bonus folded into the query profile, gssw_qP_*)
 *   E[r][c] = max(0, H[r][c-1] - go, E[r][c-1] - ge)      gap in read (deletion)
 *   F[r][c] = max(0, H[r-1][c] - go, F[r-1][c] - ge)      gap in graph (insertion)
 *   H[r][c] = max(H[r-1][c-1] + s(r,c), E[r][c], F[r][c]) (>= 0 because E,F >= 0)
 *   "c-1" at the first column of a node = element-wise max over the
 *   predecessors' last columns of H and of E-for-the-next-column
 *   (gssw_create_seed_*); a node without predecessors is seeded with zeros.
 *   local end cell : first column (in node order) attaining the global max,
 *                    smallest read index in it (SSW end_ref / end_read rule;
 *                    graph->max_node = first node with strictly greater score1)
 *   pinned end cell: (last read base, last column) of the pinning node with the
 *                    best H there; first in node order on ties  [PARITY-UNPINNED tie]
 *   traceback      : state machine over H/E/F, stops when the running score
 *                    reaches 0; preference order in H: diagonal, then E
 *                    (deletion), then F (insertion)  [PARITY-UNPINNED order];
 *                    E/F prefer "gap open" over "gap extend" on ties
 *                    [PARITY-UNPINNED]; across a node boundary the first
 *                    predecessor (in pred list order) that explains the score
 *                    is taken [PARITY-UNPINNED beyond the unit-test graphs].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"

/* gssw_create_nt_table: a/A=0 c/C=1 g/G=2 t/T=3, everything else 4 (read side) */
static inline int nt_read(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
/* graph side passes through nonATGCNtoN first (src/aligner.cpp:39, src/utility.cpp:323-332):
 * only upper-case ACGT survive, everything else (incl. lower case) becomes N */
static inline int nt_ref(char ch) {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

typedef struct { int32_t *H, *E, *F; } mats_t;

#define IDX(c, r) ((size_t)(c) * (size_t)L + (size_t)(r))

static void reverse_ops(vgk_op* a, uint32_t n) {
    for (uint32_t i = 0, j = n ? n - 1 : 0; i < j; ++i, --j) { vgk_op t = a[i]; a[i] = a[j]; a[j] = t; }
}

/* Align one read to one DAG.  `ops` receives up to ops_cap elements in forward
 * (read) order.  Returns VGK_OK or a VGK_E* code; per-problem status is also
 * stored in res->status. */
int vgo_gssw_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                     vgk_result* res, vgk_op* ops, uint32_t ops_cap);
int vgo_gssw_align(const vgk_scoring* sc, const vgk_gssw_problem* p,
                   vgk_result* res, vgk_op* ops, uint32_t ops_cap) { return vgo_gssw_align_q(sc, NULL, p, res, ops, ops_cap); }

/* qa != NULL: quality-adjusted scoring (gssw_graph_fill_pinned_qual_adj, src/aligner.cpp:942-952): the substitution
 * score depends on the read base's quality, the end bonuses on the qualities of the end bases. */
int vgo_gssw_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                     vgk_result* res, vgk_op* ops, uint32_t ops_cap)
{
    const int L = (int)p->read_len;
    const vgk_graph* g = &p->graph;
    const int nV = (int)g->n_nodes;
    const int go = sc->gap_open, ge = sc->gap_extend;
    const int pinned = (p->flags & 15) == VGK_GSSW_PINNED;
    const int want_tb = (p->flags & VGK_GSSW_TRACEBACK) != 0;
    if (qa && !p->qual) { memset(res, 0, sizeof *res); res->status = VGK_EINVAL; return VGK_EINVAL; }
    const int start_bonus = qa ? qa->bonuses[p->qual[0]] : sc->full_length_bonus;
    const int end_bonus = pinned ? 0 : (qa ? qa->bonuses[p->qual[L > 0 ? L - 1 : 0]] : sc->full_length_bonus);   /* src/aligner.cpp:402 */

    memset(res, 0, sizeof *res);
    res->end_node = -1; res->end_offset = -1; res->end_read = -1;
    if (L <= 0 || nV <= 0) { res->status = VGK_EINVAL; return VGK_EINVAL; }

    /* column bookkeeping */
    int* col0 = (int*)malloc(sizeof(int) * (size_t)(nV + 1));
    col0[0] = 0;
    for (int n = 0; n < nV; ++n) {
        if (g->node_len[n] == 0) { free(col0); res->status = VGK_EINVAL; return VGK_EINVAL; }
        if (g->node_len[n] > 65535u) { free(col0); res->status = VGK_ETOOBIG; return VGK_ETOOBIG; }   /* vgk_op.len is 16 bits: a run inside one node must fit */
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

    mats_t m;
    m.H = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * (size_t)L);
    m.E = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * (size_t)L);
    m.F = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * (size_t)L);
    /* En[c][r] = E for the column after c = max(0, H[c][r]-go, E[c][r]-ge) (gssw keeps this as pvE) */
    int32_t* En = (int32_t*)malloc(sizeof(int32_t) * (size_t)R * (size_t)L);
    int32_t* seedH = (int32_t*)malloc(sizeof(int32_t) * (size_t)L);
    int32_t* seedE = (int32_t*)malloc(sizeof(int32_t) * (size_t)L);

#define SCORE(r, c) ((int)(qa ? qa->matrix[25 * p->qual[r] + 5 * rf[c] + rd[r]] : sc->matrix[5 * rf[c] + rd[r]]) + ((r) == 0 ? start_bonus : 0) + ((r) == L - 1 ? end_bonus : 0))

    int32_t best = 0; int best_c = -1, best_r = -1;
    for (int n = 0; n < nV; ++n) {
        /* seed = element-wise max over predecessors' last column (gssw_create_seed_*) */
        for (int r = 0; r < L; ++r) { seedH[r] = 0; seedE[r] = 0; }
        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
            int pc = col0[g->pred_idx[k] + 1] - 1;
            for (int r = 0; r < L; ++r) {
                if (m.H[IDX(pc, r)] > seedH[r]) seedH[r] = m.H[IDX(pc, r)];
                if (En[IDX(pc, r)] > seedE[r]) seedE[r] = En[IDX(pc, r)];
            }
        }
        for (int c = col0[n]; c < col0[n + 1]; ++c) {
            const int first = (c == col0[n]);
            int32_t colmax = 0; int colmax_r = -1;
            for (int r = 0; r < L; ++r) {
                int32_t e = first ? seedE[r] : En[IDX(c - 1, r)];
                int32_t f = 0;
                if (r > 0) {
                    int32_t a = m.H[IDX(c, r - 1)] - go, b = m.F[IDX(c, r - 1)] - ge;
                    f = a > b ? a : b; if (f < 0) f = 0;
                }
                int32_t d = (r == 0) ? 0 : (first ? seedH[r - 1] : m.H[IDX(c - 1, r - 1)]);
                int32_t h = d + SCORE(r, c);
                if (e > h) h = e;
                if (f > h) h = f;
                m.H[IDX(c, r)] = h; m.E[IDX(c, r)] = e; m.F[IDX(c, r)] = f;
                int32_t a = h - go, b = e - ge; int32_t en = a > b ? a : b; if (en < 0) en = 0;
                En[IDX(c, r)] = en;
                if (h > colmax) { colmax = h; colmax_r = r; }   /* smallest row with the column max */
            }
            if (colmax > best) { best = colmax; best_c = c; best_r = colmax_r; }  /* first column wins */
        }
    }

    int rc = VGK_OK;
    int32_t cur; int r, c;
    if (pinned) {
        cur = 0; c = -1; r = L - 1;
        for (int n = 0; n < nV; ++n) {
            if (!p->pinning || !p->pinning[n]) continue;
            int pc = col0[n + 1] - 1;
            if (c < 0 || m.H[IDX(pc, L - 1)] > cur) { cur = m.H[IDX(pc, L - 1)]; c = pc; }
        }
        if (c < 0) { rc = VGK_EINVAL; goto done; }
    } else {
        cur = best; c = best_c; r = best_r;
    }
    if (best >= 32767) { rc = VGK_EOVERFLOW; goto done; }    /* gssw word-mode limit */
    res->score = cur;
    if (cur <= 0) {   /* nothing aligned: the caller synthesises soft clips (src/aligner.cpp:486-527) */
        res->score = 0; goto done;
    }
    res->end_node = node_of[c]; res->end_offset = c - col0[node_of[c]]; res->end_read = r;
    if (!want_tb) goto done;

    {
        uint32_t nops = 0;
#define PUSH(NODE, OP, LEN) do { \
        if (nops > 0 && ops[nops - 1].node == (uint32_t)(NODE) && ops[nops - 1].op == (OP)) ops[nops - 1].len += (LEN); \
        else { if (nops >= ops_cap) { rc = VGK_EOPS; goto done; } \
               ops[nops].node = (uint32_t)(NODE); ops[nops].op = (uint8_t)(OP); ops[nops].len = (uint16_t)(LEN); ops[nops].pad = 0; ++nops; } } while (0)
        /* ops are produced back-to-front and reversed at the end */
        if (r < L - 1) PUSH(node_of[c], VGK_OP_S, L - 1 - r);
        enum { ST_H, ST_E, ST_F } st = ST_H;
        int first_c = c;
        while (1) {
            const int n = node_of[c];
            const int first = (c == col0[n]);
            if (st == ST_H) {
                if (cur == 0) break;
                int32_t s = SCORE(r, c);
                int32_t d;
                if (r == 0) d = 0;
                else if (!first) d = m.H[IDX(c - 1, r - 1)];
                else { d = 0; for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                           int pc = col0[g->pred_idx[k] + 1] - 1; if (m.H[IDX(pc, r - 1)] > d) d = m.H[IDX(pc, r - 1)]; } }
                if (cur == d + s) {
                    PUSH(n, VGK_OP_M, 1); first_c = c;
                    cur = d; r -= 1;
                    if (r < 0) break;
                    if (cur == 0) break;
                    if (!first) c -= 1;
                    else {
                        int found = -1;
                        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                            int pc = col0[g->pred_idx[k] + 1] - 1;
                            if (m.H[IDX(pc, r)] == cur) { found = pc; break; } }
                        if (found < 0) { rc = VGK_EINVAL; goto done; }
                        c = found;
                    }
                } else if (cur == m.E[IDX(c, r)]) st = ST_E;
                else if (cur == m.F[IDX(c, r)]) st = ST_F;
                else { rc = VGK_EINVAL; goto done; }
            } else if (st == ST_E) {
                PUSH(n, VGK_OP_D, 1); first_c = c;
                int pc;
                if (!first) pc = c - 1;
                else {
                    pc = -1;
                    for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) {
                        int q = col0[g->pred_idx[k] + 1] - 1;
                        if (En[IDX(q, r)] == cur) { pc = q; break; } }
                    if (pc < 0) { rc = VGK_EINVAL; goto done; }
                }
                if (m.H[IDX(pc, r)] - go == cur) { st = ST_H; cur += go; }
                else { cur += ge; }
                c = pc;
            } else { /* ST_F */
                PUSH(n, VGK_OP_I, 1);
                if (r == 0) { rc = VGK_EINVAL; goto done; }
                if (m.H[IDX(c, r - 1)] - go == cur) { st = ST_H; cur += go; }
                else { cur += ge; }
                r -= 1;
            }
        }
        if (r >= 0) PUSH(node_of[first_c], VGK_OP_S, r + 1);
        reverse_ops(ops, nops);
        res->n_ops = nops;
        res->first_offset = first_c - col0[node_of[first_c]];
    }
done:
    res->status = rc;
    free(col0); free(node_of); free(rd); free(rf); free(m.H); free(m.E); free(m.F); free(En); free(seedH); free(seedE);
    return rc;
}

/* Batch driver used by tests and by bench.py's cpu_baseline leg ("port").
 * ops for problem i land at ops[i*ops_per_problem ...]. */
int vgo_gssw_align_batch(const vgk_scoring* sc, const vgk_gssw_problem* probs, uint32_t n,
                         vgk_result* results, vgk_op* ops, uint32_t ops_per_problem)
{
    int worst = VGK_OK;
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int rc = vgo_gssw_align(sc, &probs[i], &results[i], ops + (size_t)i * ops_per_problem, ops_per_problem);
        results[i].ops_begin = (uint32_t)((size_t)i * ops_per_problem);
        if (rc != VGK_OK) {
            #pragma omp critical
            worst = rc;
        }
    }
    return worst;
}
