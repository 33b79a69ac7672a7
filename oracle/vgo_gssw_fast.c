/*
 * vgo_gssw_fast.c — the CPU BASELINE for bench.py (test infrastructure, NOT product code): the graph Smith-Waterman of
 * vgo_gssw.c restated the way a tuned CPU implementation would run it, so that the "GPU vs CPU" figure of the bench line is
 * taken against an honest CPU number and not against the scalar int32 checker.
 *
 *   * int16 cells, 16 read rows per AVX2 vector (gssw itself is Farrar-striped SSE2, 8 x int16 / 16 x int8 per vector);
 *   * one arena per thread, sized once, no allocation per read (vg keeps per-thread state the same way, src/aligner.cpp:336-341);
 *   * the vertical gap F of a column comes from a max-plus prefix scan over the rows instead of a serial loop
 *     (F[r] = max(0, max_{r' < r} (Ht[r'] + r' ge) - go - (r - 1) ge), Ht = max(diagonal + s, E); exact when go >= ge);
 *   * 1 byte of traceback per cell (the four decisions the H/E/F state machine takes) + the last column of every node,
 *     instead of the checker's four int32 matrices;
 *   * OpenMP over reads, schedule(dynamic).
 *
 * Same semantics and tie rules as vgo_gssw.c (whose header states what is pinned on the reference's tests and what is not);
 * tests/test_cpu_baseline.py holds the two equal on random DAGs, and bench.py checks every timed read against the checker's
 * result before it reports the rate.  Plain (not quality-adjusted) scoring, LOCAL and PINNED modes; anything else — and CPUs
 * without AVX2 — returns VGK_EUNSUPPORTED and the caller uses vgo_gssw.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include "../include/vgk.h"

typedef struct {
    int16_t *prof;      /* [5][Lp]   query profile incl. bonuses */
    int16_t *Hbuf[2];   /* [Lp + 32] H of the previous / current column, 16 leading pad entries (row -1 = 0) */
    int16_t *Ebuf[2];   /* [Lp]      E for the next column */
    int16_t *Fbuf;      /* [Lp + 32] F of the current column (+1 read for the row below) */
    int16_t *Dbuf;      /* [Lp]      diagonal + s of the current column */
    int16_t *lastH, *lastE;   /* [nV][Lp] last column of every node (H, E-for-the-next-column) */
    int16_t *ramp, *sub;      /* [Lp] r*ge, go + (r-1)*ge */
    uint8_t *tb;        /* [R][Lp] */
    uint8_t *rf; int32_t *col0; int32_t *node_of;
    size_t cap_L, cap_nodes, cap_cells, cap_R;
} fast_arena;

static void arena_free(fast_arena* a) {
    free(a->prof); free(a->Hbuf[0]); free(a->Hbuf[1]); free(a->Ebuf[0]); free(a->Ebuf[1]); free(a->Fbuf); free(a->Dbuf);
    free(a->lastH); free(a->lastE); free(a->ramp); free(a->sub); free(a->tb); free(a->rf); free(a->col0); free(a->node_of);
    memset(a, 0, sizeof *a);
}
static void* xalloc(size_t bytes) { void* p = NULL; if (posix_memalign(&p, 64, bytes ? bytes : 64)) return NULL; return p; }
static int arena_fit(fast_arena* a, size_t Lp, size_t nV, size_t R) {
    if (Lp > a->cap_L || nV * Lp > a->cap_nodes || R * Lp > a->cap_cells || R > a->cap_R || nV + 1 > a->cap_R) {
        const size_t L2 = Lp > a->cap_L ? Lp : a->cap_L;
        size_t nodes = nV * Lp > a->cap_nodes ? nV * Lp * 2 : a->cap_nodes, cells = R * Lp > a->cap_cells ? R * Lp * 2 : a->cap_cells;
        size_t R2 = (R > nV + 1 ? R : nV + 1); R2 = R2 > a->cap_R ? R2 * 2 : a->cap_R;
        arena_free(a);
        a->prof = xalloc(sizeof(int16_t) * 5 * L2);
        for (int k = 0; k < 2; ++k) { a->Hbuf[k] = xalloc(sizeof(int16_t) * (L2 + 32)); a->Ebuf[k] = xalloc(sizeof(int16_t) * L2); }
        a->Fbuf = xalloc(sizeof(int16_t) * (L2 + 32)); a->Dbuf = xalloc(sizeof(int16_t) * L2);
        a->lastH = xalloc(sizeof(int16_t) * nodes); a->lastE = xalloc(sizeof(int16_t) * nodes);
        a->ramp = xalloc(sizeof(int16_t) * L2); a->sub = xalloc(sizeof(int16_t) * L2);
        a->tb = xalloc(cells); a->rf = xalloc(R2); a->col0 = xalloc(sizeof(int32_t) * (R2 + 1)); a->node_of = xalloc(sizeof(int32_t) * R2);
        if (!a->prof || !a->Hbuf[0] || !a->Hbuf[1] || !a->Ebuf[0] || !a->Ebuf[1] || !a->Fbuf || !a->Dbuf || !a->lastH || !a->lastE ||
            !a->ramp || !a->sub || !a->tb || !a->rf || !a->col0 || !a->node_of) { arena_free(a); return 0; }
        a->cap_L = L2; a->cap_nodes = nodes; a->cap_cells = cells; a->cap_R = R2;
    }
    return 1;
}

static inline int nt_read(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static inline int nt_ref(char ch) {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

/* lanes move up by N rows (towards higher indices), zeros come in: values are >= 0 and only ever maximised */
#define SHIFT_UP(v, N) _mm256_alignr_epi8((v), _mm256_permute2x128_si256((v), (v), 0x08), 16 - 2 * (N))
__attribute__((target("avx2")))
static inline __m256i shift_up8(__m256i v) { return _mm256_permute2x128_si256(v, v, 0x08); }

__attribute__((target("avx2")))
static int fast_align(fast_arena* A, const vgk_scoring* sc, const vgk_gssw_problem* p, vgk_result* res, vgk_op* ops, uint32_t ops_cap) {
    const int L = (int)p->read_len;
    const vgk_graph* g = &p->graph;
    const int nV = (int)g->n_nodes;
    const int go = sc->gap_open, ge = sc->gap_extend;
    const int mode = (int)(p->flags & 15u);
    const int pinned = mode == VGK_GSSW_PINNED;
    const int want_tb = (p->flags & VGK_GSSW_TRACEBACK) != 0;
    memset(res, 0, sizeof *res);
    res->end_node = -1; res->end_offset = -1; res->end_read = -1;
    if (mode != VGK_GSSW_LOCAL && mode != VGK_GSSW_PINNED) return VGK_EUNSUPPORTED;
    if (L <= 0 || nV <= 0) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    int maxs = 0;
    for (int k = 0; k < 25; ++k) if (sc->matrix[k] > maxs) maxs = sc->matrix[k];
    if (go < ge || sc->full_length_bonus < 0 || (int64_t)L * (maxs + ge) + 2 * sc->full_length_bonus + go > 30000) return VGK_EUNSUPPORTED;
    const int Lp = (L + 31) & ~31;                 /* two vectors per step of the flag pass */
    int64_t R64 = 0;
    for (int n = 0; n < nV; ++n) {
        if (g->node_len[n] == 0 || g->pred_off[n + 1] < g->pred_off[n]) { res->status = VGK_EINVAL; return VGK_EINVAL; }
        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k) if ((int)g->pred_idx[k] >= n) { res->status = VGK_EINVAL; return VGK_EINVAL; }
        R64 += g->node_len[n];
    }
    if (R64 >= (1 << 20)) return VGK_EUNSUPPORTED;
    const int R = (int)R64;
    if (!arena_fit(A, (size_t)Lp, (size_t)nV, (size_t)R)) { res->status = VGK_ENOMEM; return VGK_ENOMEM; }
    int32_t* col0 = A->col0; int32_t* node_of = A->node_of; uint8_t* rf = A->rf;
    col0[0] = 0;
    for (int n = 0; n < nV; ++n) { col0[n + 1] = col0[n] + (int)g->node_len[n]; for (int c = col0[n]; c < col0[n + 1]; ++c) node_of[c] = n; }
    for (int c = 0; c < R; ++c) rf[c] = (uint8_t)nt_ref(g->seq[c]);
    const int start_bonus = sc->full_length_bonus, end_bonus = pinned ? 0 : sc->full_length_bonus;
    for (int b = 0; b < 5; ++b) {
        int16_t* pr = A->prof + (size_t)b * Lp;
        for (int r = 0; r < L; ++r) pr[r] = (int16_t)(sc->matrix[5 * b + nt_read(p->read[r])] + (r == 0 ? start_bonus : 0) + (r == L - 1 ? end_bonus : 0));
        for (int r = L; r < Lp; ++r) pr[r] = 0;
    }
    for (int r = 0; r < Lp; ++r) { A->ramp[r] = (int16_t)(r * ge); A->sub[r] = (int16_t)(go + (r - 1) * ge); }
    for (int k = 0; k < 2; ++k) memset(A->Hbuf[k], 0, sizeof(int16_t) * ((size_t)Lp + 32));
    memset(A->Fbuf, 0, sizeof(int16_t) * ((size_t)Lp + 32));
    const __m256i vgo = _mm256_set1_epi16((short)go), vge = _mm256_set1_epi16((short)ge), zero = _mm256_setzero_si256();
    const __m256i b1 = _mm256_set1_epi16(1), b2 = _mm256_set1_epi16(2), b4 = _mm256_set1_epi16(4), b8 = _mm256_set1_epi16(8);
    int32_t best = 0; int best_c = -1, best_r = -1;
    int cur_buf = 0;
    for (int n = 0; n < nV; ++n) {
        /* seed = element-wise max over the predecessors' last columns (gssw_create_seed_*); zeros without predecessors */
        int16_t* Hp = A->Hbuf[cur_buf] + 16; int16_t* Ep = A->Ebuf[cur_buf];
        const uint32_t pb = g->pred_off[n], pe = g->pred_off[n + 1];
        const int chain = (pe - pb == 1) && (int)g->pred_idx[pb] == n - 1;
        if (!chain) {                                  /* (a chain link finds its seed in the rolling buffers already) */
            for (int r = 0; r < Lp; r += 16) { _mm256_store_si256((__m256i*)(Hp + r), zero); _mm256_store_si256((__m256i*)(Ep + r), zero); }
            for (uint32_t k = pb; k < pe; ++k) {
                const int16_t* lh = A->lastH + (size_t)g->pred_idx[k] * Lp; const int16_t* le = A->lastE + (size_t)g->pred_idx[k] * Lp;
                for (int r = 0; r < Lp; r += 16) {
                    _mm256_store_si256((__m256i*)(Hp + r), _mm256_max_epi16(_mm256_load_si256((const __m256i*)(Hp + r)), _mm256_load_si256((const __m256i*)(lh + r))));
                    _mm256_store_si256((__m256i*)(Ep + r), _mm256_max_epi16(_mm256_load_si256((const __m256i*)(Ep + r)), _mm256_load_si256((const __m256i*)(le + r))));
                }
            }
        }
        for (int c = col0[n]; c < col0[n + 1]; ++c) {
            const int16_t* Hprev = A->Hbuf[cur_buf] + 16; const int16_t* Eprev = A->Ebuf[cur_buf];
            int16_t* Hcur = A->Hbuf[cur_buf ^ 1] + 16; int16_t* Ecur = A->Ebuf[cur_buf ^ 1];
            const int16_t* pr = A->prof + (size_t)rf[c] * Lp;
            int16_t* F = A->Fbuf + 16; int16_t* D = A->Dbuf;
            uint8_t* tb = A->tb + (size_t)c * Lp;
            __m256i carry = zero, colmax = zero;
            for (int r = 0; r < Lp; r += 16) {
                const __m256i dg = _mm256_add_epi16(_mm256_loadu_si256((const __m256i*)(Hprev + r - 1)), _mm256_load_si256((const __m256i*)(pr + r)));
                const __m256i e = _mm256_load_si256((const __m256i*)(Eprev + r));
                const __m256i ht = _mm256_max_epi16(dg, e);
                /* inclusive prefix max of Ht + r ge over the rows of this vector, then the rows before it through `carry` */
                __m256i s = _mm256_add_epi16(ht, _mm256_load_si256((const __m256i*)(A->ramp + r)));
                s = _mm256_max_epi16(s, SHIFT_UP(s, 1)); s = _mm256_max_epi16(s, SHIFT_UP(s, 2));
                s = _mm256_max_epi16(s, SHIFT_UP(s, 4)); s = _mm256_max_epi16(s, shift_up8(s));
                const __m256i excl = _mm256_max_epi16(SHIFT_UP(s, 1), carry);
                carry = _mm256_max_epi16(carry, _mm256_set1_epi16((short)_mm256_extract_epi16(s, 15)));
                const __m256i f = _mm256_subs_epu16(excl, _mm256_load_si256((const __m256i*)(A->sub + r)));
                const __m256i h = _mm256_max_epi16(ht, f);
                const __m256i en = _mm256_max_epi16(_mm256_subs_epu16(h, vgo), _mm256_subs_epu16(e, vge));
                _mm256_store_si256((__m256i*)(Hcur + r), h); _mm256_store_si256((__m256i*)(Ecur + r), en);
                _mm256_store_si256((__m256i*)(F + r), f); _mm256_store_si256((__m256i*)(D + r), dg);
                colmax = _mm256_max_epi16(colmax, r + 16 <= L ? h : _mm256_and_si256(h, _mm256_cmpgt_epi16(_mm256_set1_epi16((short)(L - r)), _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15))));
            }
            if (want_tb) for (int r = 0; r < Lp; r += 32) {
                __m256i code[2];
                for (int q = 0; q < 2; ++q) {
                    const int rr = r + 16 * q;
                    const __m256i h = _mm256_load_si256((const __m256i*)(Hcur + rr)), e = _mm256_load_si256((const __m256i*)(Eprev + rr));
                    const __m256i hg = _mm256_sub_epi16(h, vgo);
                    const __m256i nd = _mm256_andnot_si256(_mm256_cmpeq_epi16(h, _mm256_load_si256((const __m256i*)(D + rr))), b1);     /* H not from the diagonal */
                    const __m256i hf = _mm256_andnot_si256(_mm256_cmpeq_epi16(h, e), b2);                                                /* ... and not from E: from F */
                    const __m256i ee = _mm256_andnot_si256(_mm256_cmpeq_epi16(_mm256_load_si256((const __m256i*)(Ecur + rr)), hg), b4);  /* next-column E extends */
                    const __m256i fe = _mm256_andnot_si256(_mm256_cmpeq_epi16(_mm256_loadu_si256((const __m256i*)(F + rr + 1)), hg), b8); /* next-row F extends */
                    code[q] = _mm256_or_si256(_mm256_or_si256(nd, hf), _mm256_or_si256(ee, fe));
                }
                _mm256_store_si256((__m256i*)(tb + r), _mm256_permute4x64_epi64(_mm256_packus_epi16(code[0], code[1]), 0xD8));
            }
            /* local end cell: first column with a strictly greater maximum, smallest row in it (SSW's rule) */
            {
                __m256i m = _mm256_max_epi16(colmax, _mm256_permute2x128_si256(colmax, colmax, 1));
                m = _mm256_max_epi16(m, _mm256_srli_si256(m, 8)); m = _mm256_max_epi16(m, _mm256_srli_si256(m, 4)); m = _mm256_max_epi16(m, _mm256_srli_si256(m, 2));
                const int cm = (int16_t)_mm256_extract_epi16(m, 0);
                if (cm > best) {
                    best = cm; best_c = c;
                    for (int r = 0; r < L; ++r) if (Hcur[r] == cm) { best_r = r; break; }
                }
            }
            cur_buf ^= 1;
        }
        memcpy(A->lastH + (size_t)n * Lp, A->Hbuf[cur_buf] + 16, sizeof(int16_t) * (size_t)Lp);
        memcpy(A->lastE + (size_t)n * Lp, A->Ebuf[cur_buf], sizeof(int16_t) * (size_t)Lp);
    }

    int rc = VGK_OK;
    int32_t cur; int r, c;
    if (pinned) {
        cur = 0; c = -1; r = L - 1;
        for (int n = 0; n < nV; ++n) {
            if (!p->pinning || !p->pinning[n]) continue;
            const int32_t v = A->lastH[(size_t)n * Lp + (L - 1)];
            if (c < 0 || v > cur) { cur = v; c = col0[n + 1] - 1; }
        }
        if (c < 0) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    } else { cur = best; c = best_c; r = best_r; }
    res->score = cur;
    if (cur <= 0) { res->score = 0; return VGK_OK; }
    res->end_node = node_of[c]; res->end_offset = c - col0[node_of[c]]; res->end_read = r;
    if (!want_tb) return VGK_OK;
    {
        uint32_t nops = 0;
#define PUSH(NODE, OP, LEN) do { \
        if (nops > 0 && ops[nops - 1].node == (uint32_t)(NODE) && ops[nops - 1].op == (OP)) ops[nops - 1].len += (LEN); \
        else { if (nops >= ops_cap) { rc = VGK_EOPS; goto done; } \
               ops[nops].node = (uint32_t)(NODE); ops[nops].op = (uint8_t)(OP); ops[nops].len = (uint16_t)(LEN); ops[nops].pad = 0; ++nops; } } while (0)
#define TB(C, RR) (A->tb[(size_t)(C) * Lp + (RR)])
        if (r < L - 1) PUSH(node_of[c], VGK_OP_S, L - 1 - r);
        enum { ST_H, ST_E, ST_F } st = ST_H;
        int first_c = c;
        for (;;) {
            const int n = node_of[c];
            const int first = (c == col0[n]);
            if (st == ST_H) {
                if (cur == 0) break;
                const uint8_t code = TB(c, r);
                if (!(code & 1)) {
                    PUSH(n, VGK_OP_M, 1); first_c = c;
                    cur -= A->prof[(size_t)rf[c] * Lp + r]; r -= 1;
                    if (r < 0 || cur == 0) break;
                    if (!first) c -= 1;
                    else {
                        int found = -1;
                        for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k)
                            if (A->lastH[(size_t)g->pred_idx[k] * Lp + r] == cur) { found = col0[g->pred_idx[k] + 1] - 1; break; }
                        if (found < 0) { rc = VGK_EINVAL; goto done; }
                        c = found;
                    }
                } else st = (code & 2) ? ST_F : ST_E;
            } else if (st == ST_E) {
                PUSH(n, VGK_OP_D, 1); first_c = c;
                int pc;
                if (!first) pc = c - 1;
                else {
                    pc = -1;
                    for (uint32_t k = g->pred_off[n]; k < g->pred_off[n + 1]; ++k)
                        if (A->lastE[(size_t)g->pred_idx[k] * Lp + r] == cur) { pc = col0[g->pred_idx[k] + 1] - 1; break; }
                    if (pc < 0) { rc = VGK_EINVAL; goto done; }
                }
                if (!(TB(pc, r) & 4)) { st = ST_H; cur += go; } else cur += ge;
                c = pc;
            } else {
                PUSH(n, VGK_OP_I, 1);
                if (r == 0) { rc = VGK_EINVAL; goto done; }
                if (!(TB(c, r - 1) & 8)) { st = ST_H; cur += go; } else cur += ge;
                r -= 1;
            }
        }
        if (r >= 0) PUSH(node_of[first_c], VGK_OP_S, r + 1);
        for (uint32_t i = 0, j = nops ? nops - 1 : 0; i < j; ++i, --j) { vgk_op t = ops[i]; ops[i] = ops[j]; ops[j] = t; }
        res->n_ops = nops;
        res->first_offset = first_c - col0[node_of[first_c]];
    }
done:
    res->status = rc;
    return rc;
}

int vgo_gssw_fast_supported(void) { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* Batch driver for bench.py's cpu_baseline leg: ops of problem i land at ops[i * ops_per_problem ...].  Returns VGK_EUNSUPPORTED
 * (and touches nothing) when some problem or the CPU is outside the fast path's range. */
int vgo_gssw_fast_batch(const vgk_scoring* sc, const vgk_gssw_problem* probs, uint32_t n, vgk_result* results, vgk_op* ops, uint32_t ops_per_problem) {
    if (!vgo_gssw_fast_supported()) return VGK_EUNSUPPORTED;
    int worst = VGK_OK;
    #pragma omp parallel
    {
        fast_arena A; memset(&A, 0, sizeof A);
        #pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const int rc = fast_align(&A, sc, &probs[i], &results[i], ops + (size_t)i * ops_per_problem, ops_per_problem);
            results[i].ops_begin = (uint32_t)((size_t)i * ops_per_problem);
            if (rc != VGK_OK) {
                #pragma omp critical
                { if (worst == VGK_OK || rc == VGK_EUNSUPPORTED) worst = rc; }
            }
        }
        arena_free(&A);
    }
    return worst;
}
