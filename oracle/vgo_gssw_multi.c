/*
 * vgo_gssw_multi.c — CPU ORACLE for the k-best pinned tracebacks behind Aligner::align_pinned_multi
 * (reference src/aligner.cpp:423-435: gssw_graph_trace_back_pinned_multi).
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * PARITY STATUS: PARITY-UNPINNED beyond the reference's nine property tests.  gssw (deps/gssw) is an empty submodule in
 * the snapshot, so the rules of its multi-traceback are not available.  What the reference's call site and unit tests
 * (src/unittest/pinned_alignment.cpp:1951-2530) fix is: the first alignment is the single pinned traceback; alternates come
 * in non-increasing score order and stop at the first score <= 0 (src/aligner.cpp:455-462); equally good optima are all
 * found; alternates that take the other branch of a bubble are found; no alignment is returned twice.  The enumeration
 * below is this engine's own, modelled on the in-tree BandedGlobalAligner::AltTracebackStack (src/banded_global_aligner.cpp):
 *
 *   A traceback is the walk of vgo_gssw.c's state machine over H / E / F.  At every state the possible sources are listed
 *   in a fixed order — H: diagonal via each predecessor column (list order), then E, then F; E: per predecessor column
 *   (list order) gap-open then gap-extend; F: gap-open then gap-extend — each with its loss = value of the state minus
 *   value through that source.  The default walk takes the first source with loss 0 (exactly the single traceback); an
 *   alternate is a walk that takes a named other source at some states (its "deflections") and the default elsewhere, and
 *   scores the start value minus the losses.  Alternates are expanded best-first: while the walk of an alternate runs past
 *   its last deflection, every other source it passes is proposed as a new alternate.  Walks start at the last read row of
 *   each pinning node's last column.  To keep renderings distinct, a gap is never opened directly after a gap of the same
 *   kind was opened (the two would print as one longer gap), sources worth 0 or less are not entered, diagonal sources through
 *   predecessor cells worth 0 count as one (the alignment starts at the current cell either way), and a walk ends only where
 *   the DP value reaches 0 or the read is used up.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"

static inline int nt_read(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static inline int nt_ref(char ch) {
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

enum { ST_H = 0, ST_E = 1, ST_F = 2 };
typedef struct { int32_t st, r, c, opt; } Defl;                   /* at state (st, r, c) take source number `opt` */
typedef struct { int32_t score, start; Defl* d; uint32_t n; } Alt;

typedef struct {
    const vgk_gssw_problem* p; const vgk_graph* g; int L, nV, R, go, ge;
    int* col0; int* node_of; int32_t *H, *E, *F;
    const int8_t* rd; const int8_t* rf; const vgk_scoring* sc; const vgk_qual_adj* qa; int start_bonus;
} Ctx;
#define IDX(c, r) ((size_t)(c) * (size_t)x->L + (size_t)(r))

static int32_t score_of(const Ctx* x, int r, int c) {
    const int base = x->qa ? x->qa->matrix[25 * x->p->qual[r] + 5 * x->rf[c] + x->rd[r]] : x->sc->matrix[5 * x->rf[c] + x->rd[r]];
    return base + (r == 0 ? x->start_bonus : 0);                  /* pinned: no bonus at the pinned end (src/aligner.cpp:402) */
}
static int32_t en_of(const Ctx* x, int c, int r) {                /* E for the column after c */
    int32_t a = x->H[IDX(c, r)] - x->go, b = x->E[IDX(c, r)] - x->ge, m = a > b ? a : b;
    return m > 0 ? m : 0;
}
/* predecessor columns of column c, in list order */
static uint32_t pred_cols(const Ctx* x, int c, int* out, uint32_t cap) {
    const int n = x->node_of[c];
    if (c != x->col0[n]) { out[0] = c - 1; return 1; }
    uint32_t k = 0;
    for (uint32_t q = x->g->pred_off[n]; q < x->g->pred_off[n + 1] && k < cap; ++q) out[k++] = x->col0[x->g->pred_idx[q] + 1] - 1;
    return k;
}

/* the sources of a state, in the fixed order: value through the source and the state it leads to (st < 0: the walk ends) */
typedef struct { int32_t value, st, r, c; } Src;
static uint32_t sources(const Ctx* x, int st, int r, int c, int no_e, int no_f, Src* out, uint32_t cap) {
    int pc[64]; const uint32_t np = pred_cols(x, c, pc, 64);
    uint32_t k = 0;
    if (st == ST_H) {
        const int32_t s = score_of(x, r, c);
        if (r == 0 || np == 0) { if (k < cap) out[k++] = (Src){ s, -1, r - 1, c }; }          /* the read (or the graph) starts here */
        else {
            int zero_seen = 0;                                    /* predecessors whose cell is worth 0 all mean "the alignment starts here": one source */
            for (uint32_t q = 0; q < np && k < cap; ++q) {
                const int32_t d = x->H[IDX(pc[q], r - 1)];
                if (d == 0) { if (zero_seen) continue; zero_seen = 1; }
                out[k++] = (Src){ d + s, ST_H, r - 1, pc[q] };
            }
        }
        if (k < cap) out[k++] = (Src){ no_e ? 0 : x->E[IDX(c, r)], ST_E, r, c };
        if (k < cap) out[k++] = (Src){ no_f ? 0 : x->F[IDX(c, r)], ST_F, r, c };
    } else if (st == ST_E) {
        for (uint32_t q = 0; q < np && k + 1 < cap; ++q) {
            out[k++] = (Src){ x->H[IDX(pc[q], r)] - x->go, ST_H, r, pc[q] };
            out[k++] = (Src){ x->E[IDX(pc[q], r)] - x->ge, ST_E, r, pc[q] };
        }
    } else if (r > 0) {
        out[k++] = (Src){ x->H[IDX(c, r - 1)] - x->go, ST_H, r - 1, c };
        out[k++] = (Src){ x->F[IDX(c, r - 1)] - x->ge, ST_F, r - 1, c };
    }
    return k;
}

/* can a walk that may not enter E (F) at H(r, c) still explain the cell's value? */
static int viable(const Ctx* x, int r, int c, int no_e, int no_f) {
    const int32_t value = x->H[IDX(c, r)];
    if (value == 0) return 1;
    Src src[130]; const uint32_t ns = sources(x, ST_H, r, c, no_e, no_f, src, 130);
    for (uint32_t k = 0; k < ns; ++k) if (src[k].value == value) return 1;
    return 0;
}

typedef struct { Alt* a; uint32_t n, cap; } Queue;                /* sorted: best first, earlier proposals first among equals */
static void q_insert(Queue* q, Alt a, uint32_t room) {
    if (!room) { free(a.d); return; }
    uint32_t at = q->n;
    while (at > 0 && q->a[at - 1].score < a.score) --at;
    if (at >= room) { free(a.d); return; }
    if (q->n == q->cap) { q->cap = q->cap ? 2 * q->cap : 16; q->a = (Alt*)realloc(q->a, sizeof(Alt) * q->cap); }
    memmove(q->a + at + 1, q->a + at, sizeof(Alt) * (q->n - at));
    q->a[at] = a; ++q->n;
    while (q->n > room) { --q->n; free(q->a[q->n].d); }
}

typedef struct { vgk_op* ops; uint32_t n, cap; } Ops;
static void push_op(Ops* o, uint32_t node, int op, uint32_t len) {
    if (o->n && o->ops[o->n - 1].node == node && o->ops[o->n - 1].op == op) { o->ops[o->n - 1].len = (uint16_t)(o->ops[o->n - 1].len + len); return; }
    if (o->n == o->cap) { o->cap = o->cap ? 2 * o->cap : 32; o->ops = (vgk_op*)realloc(o->ops, sizeof(vgk_op) * o->cap); }
    o->ops[o->n].node = node; o->ops[o->n].op = (uint8_t)op; o->ops[o->n].len = (uint16_t)len; o->ops[o->n].pad = 0; ++o->n;
}

/* walk one alternate: emits its ops (back to front, reversed at the end) and proposes the alternates that branch off it */
static void walk(const Ctx* x, const Alt* alt, const int* start_cols, Queue* q, uint32_t room, vgk_result* res, Ops* o) {
    int st = ST_H, r = x->L - 1, c = start_cols[alt->start], no_e = 0, no_f = 0, first_c = c;
    uint32_t next_defl = 0;
    int32_t lost = 0;                                             /* losses taken so far along this walk */
    const int32_t start_value = x->H[IDX(c, r)];
    memset(res, 0, sizeof *res);
    res->score = alt->score; res->end_node = x->node_of[c]; res->end_offset = c - x->col0[x->node_of[c]]; res->end_read = r;
    o->n = 0;
    for (;;) {
        const int32_t value = st == ST_H ? x->H[IDX(c, r)] : st == ST_E ? x->E[IDX(c, r)] : x->F[IDX(c, r)];
        if (st == ST_H && value == 0) break;                      /* the local alignment starts after this cell */
        Src src[130]; const uint32_t ns = sources(x, st, r, c, no_e, no_f, src, 130);
        uint32_t take = ns;
        if (next_defl < alt->n && alt->d[next_defl].st == st && alt->d[next_defl].r == r && alt->d[next_defl].c == c) take = (uint32_t)alt->d[next_defl++].opt;
        else {
            for (uint32_t k = 0; k < ns; ++k) if (src[k].value == value) { take = k; break; }
            if (take >= ns && (no_e || no_f)) {                   /* gap_open == gap_extend: the single traceback re-opens the gap; so does the default walk */
                no_e = no_f = 0;
                const uint32_t ns2 = sources(x, st, r, c, 0, 0, src, 130);
                for (uint32_t k = 0; k < ns2; ++k) if (src[k].value == value) { take = k; break; }
            }
        }
        if (take >= ns) break;                                    /* cannot happen on consistent matrices */
        if (next_defl == alt->n && !(alt->n && alt->d[alt->n - 1].st == st && alt->d[alt->n - 1].r == r && alt->d[alt->n - 1].c == c)) {
            /* past the last deflection: every other source worth more than 0 starts a new alternate */
            for (uint32_t k = 0; k < ns; ++k) {
                if (k == take || src[k].value <= 0) continue;
                if (st != ST_H && src[k].st == ST_H && !viable(x, src[k].r, src[k].c, st == ST_E, st == ST_F)) continue;   /* would print as the gap-extend walk */
                const int32_t sc2 = start_value - lost - (value - src[k].value);
                if (sc2 <= 0) continue;
                Alt a; a.score = sc2; a.start = alt->start; a.n = alt->n + 1; a.d = (Defl*)malloc(sizeof(Defl) * a.n);
                if (alt->n) memcpy(a.d, alt->d, sizeof(Defl) * alt->n);
                a.d[alt->n] = (Defl){ st, r, c, (int32_t)k };
                q_insert(q, a, room);
            }
        }
        lost += value - src[take].value;
        const uint32_t node = (uint32_t)x->node_of[c];
        if (st == ST_H && (take + 2 < ns || src[take].st < 0)) {          /* a diagonal source */
            push_op(o, node, VGK_OP_M, 1); first_c = c; no_e = no_f = 0;
            if (src[take].st < 0) { r -= 1; break; }
            st = ST_H; r = src[take].r; c = src[take].c;
        } else if (st == ST_H) { st = src[take].st; }                      /* into E or F of the same cell */
        else if (st == ST_E) {
            push_op(o, node, VGK_OP_D, 1); first_c = c;
            no_e = src[take].st == ST_H; no_f = 0;                        /* opened: the H state it came from may not open a deletion again */
            st = src[take].st; c = src[take].c;
        } else {
            push_op(o, node, VGK_OP_I, 1);
            no_f = src[take].st == ST_H; no_e = 0;
            st = src[take].st; r = src[take].r;
        }
    }
    if (r >= 0) push_op(o, (uint32_t)x->node_of[first_c], VGK_OP_S, (uint32_t)r + 1);
    for (uint32_t i = 0, j = o->n; i + 1 < j; ++i) { --j; const vgk_op t = o->ops[i]; o->ops[i] = o->ops[j]; o->ops[j] = t; }
    res->n_ops = o->n; res->first_offset = first_c - x->col0[x->node_of[first_c]];
}

/* Up to max_alt_alns pinned alignments of one problem, best first.  *results / *ops are malloc'ed by this call. */
int vgo_gssw_pinned_multi(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p, uint32_t max_alt_alns,
                          vgk_result** results_out, uint32_t* n_out, vgk_op** ops_out, uint32_t* n_ops_out) {
    *results_out = NULL; *ops_out = NULL; *n_out = 0; *n_ops_out = 0;
    Ctx X; Ctx* x = &X; memset(x, 0, sizeof X);
    x->p = p; x->g = &p->graph; x->L = (int)p->read_len; x->nV = (int)p->graph.n_nodes; x->go = sc->gap_open; x->ge = sc->gap_extend; x->sc = sc; x->qa = qa;
    if ((p->flags & 15u) != VGK_GSSW_PINNED || !p->pinning || x->L <= 0 || x->nV <= 0 || !max_alt_alns || (qa && !p->qual)) return VGK_EINVAL;
    x->start_bonus = qa ? qa->bonuses[p->qual[0]] : sc->full_length_bonus;
    x->col0 = (int*)malloc(sizeof(int) * (size_t)(x->nV + 1)); x->col0[0] = 0;
    for (int n = 0; n < x->nV; ++n) {
        if (x->g->node_len[n] == 0) { free(x->col0); return VGK_EINVAL; }
        x->col0[n + 1] = x->col0[n] + (int)x->g->node_len[n];
        for (uint32_t k = x->g->pred_off[n]; k < x->g->pred_off[n + 1]; ++k) if ((int)x->g->pred_idx[k] >= n) { free(x->col0); return VGK_EINVAL; }
        if (x->g->pred_off[n + 1] - x->g->pred_off[n] > 64) { free(x->col0); return VGK_ETOOBIG; }
    }
    x->R = x->col0[x->nV];
    x->node_of = (int*)malloc(sizeof(int) * (size_t)x->R);
    for (int n = 0; n < x->nV; ++n) for (int c = x->col0[n]; c < x->col0[n + 1]; ++c) x->node_of[c] = n;
    int8_t* rd = (int8_t*)malloc((size_t)x->L); int8_t* rf = (int8_t*)malloc((size_t)x->R);
    for (int r = 0; r < x->L; ++r) rd[r] = (int8_t)nt_read(p->read[r]);
    for (int c = 0; c < x->R; ++c) rf[c] = (int8_t)nt_ref(x->g->seq[c]);
    x->rd = rd; x->rf = rf;
    const size_t cells = (size_t)x->R * (size_t)x->L;
    x->H = (int32_t*)malloc(sizeof(int32_t) * cells); x->E = (int32_t*)malloc(sizeof(int32_t) * cells); x->F = (int32_t*)malloc(sizeof(int32_t) * cells);
    /* the fill of vgo_gssw.c (pinned: zero-floored H / E / F, seeds = element-wise max over the predecessors' last columns) */
    int rc = VGK_OK;
    for (int c = 0; c < x->R; ++c) {
        int pc[64]; const uint32_t np = pred_cols(x, c, pc, 64);
        for (int r = 0; r < x->L; ++r) {
            int32_t e = 0, d = 0;
            for (uint32_t q = 0; q < np; ++q) {
                const int32_t en = en_of(x, pc[q], r); if (en > e) e = en;
                if (r > 0 && x->H[IDX(pc[q], r - 1)] > d) d = x->H[IDX(pc[q], r - 1)];
            }
            int32_t f = 0;
            if (r > 0) { const int32_t a = x->H[IDX(c, r - 1)] - x->go, b = x->F[IDX(c, r - 1)] - x->ge; f = a > b ? a : b; if (f < 0) f = 0; }
            int32_t h = (r == 0 ? 0 : d) + score_of(x, r, c);
            if (e > h) h = e;
            if (f > h) h = f;
            if (h >= 32767) rc = VGK_EOVERFLOW;
            x->H[IDX(c, r)] = h; x->E[IDX(c, r)] = e; x->F[IDX(c, r)] = f;
        }
    }
    vgk_result* results = NULL; Ops all = { NULL, 0, 0 }; uint32_t n_res = 0;
    if (rc == VGK_OK) {
        /* start cells: the pinning nodes' last columns, best first (node order among equals) */
        int* start_cols = (int*)malloc(sizeof(int) * (size_t)x->nV); uint32_t n_starts = 0;
        for (int n = 0; n < x->nV; ++n) if (p->pinning[n]) start_cols[n_starts++] = x->col0[n + 1] - 1;
        Queue q = { NULL, 0, 0 };
        for (uint32_t s = 0; s < n_starts; ++s) {
            const int32_t v = x->H[IDX(start_cols[s], x->L - 1)];
            if (v <= 0) continue;
            Alt a = { v, (int32_t)s, NULL, 0 };
            q_insert(&q, a, max_alt_alns);
        }
        results = (vgk_result*)calloc(max_alt_alns, sizeof(vgk_result));
        Ops one = { NULL, 0, 0 };
        while (q.n && n_res < max_alt_alns) {
            Alt alt = q.a[0]; memmove(q.a, q.a + 1, sizeof(Alt) * (q.n - 1)); --q.n;
            walk(x, &alt, start_cols, &q, max_alt_alns - n_res - 1, &results[n_res], &one);
            results[n_res].ops_begin = all.n;
            for (uint32_t k = 0; k < one.n; ++k) { if (all.n == all.cap) { all.cap = all.cap ? 2 * all.cap : 64; all.ops = (vgk_op*)realloc(all.ops, sizeof(vgk_op) * all.cap); } all.ops[all.n++] = one.ops[k]; }
            ++n_res; free(alt.d);
        }
        for (uint32_t k = 0; k < q.n; ++k) free(q.a[k].d);
        free(q.a); free(one.ops); free(start_cols);
        if (n_starts == 0) rc = VGK_EINVAL;
    }
    free(x->col0); free(x->node_of); free(rd); free(rf); free(x->H); free(x->E); free(x->F);
    if (rc != VGK_OK) { free(results); free(all.ops); return rc; }
    *results_out = results; *n_out = n_res; *ops_out = all.ops; *n_ops_out = all.n;
    return VGK_OK;
}
