/*
 * vgo_banded.c — CPU ORACLE for vg's banded global graph aligner.
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * A restatement, in absolute (read row r, node column j) coordinates, of the algorithm in the
 * reference's src/banded_global_aligner.cpp:
 *   - band geometry and masking          find_banded_paths           (:2174-2268), path_lengths_to_sinks (:2122-2170)
 *   - shortest lead-deletion lengths     shortest_seq_paths          (:2271-2293)
 *   - cell budget                        BandMatricesTooBigException (:1999-2015)
 *   - three-matrix fill                  BAMatrix::fill_matrix       (:251-742)
 *   - choice of the end cell             AltTracebackStack ctor      (:2426-2563), insert_traceback (:2691-2740)
 *   - traceback inside a node            BAMatrix::traceback         (:756-1126)
 *   - traceback across an edge           BAMatrix::traceback_over_edge (:1129-1780)
 *   - edits                              BABuilder                   (:44-205)
 * Only the primary alignment (max_multi_alns == 1) is produced.
 *
 * The band of a node is the set of diagonals d = r - j in [top, bot]; the reference stores it as a
 * rectangle whose row k is diagonal top + k, and so do we (cell (r, j) lives at row r - j - top).
 *
 * Parity status: pinned on the reference's own known-answer tests (src/unittest/banded_global_aligner.cpp,
 * transcribed to tests/golden/ref_banded_global_aligner.json).  One tie rule cannot be pinned:
 * PARITY-UNPINNED(sink order) — among several sink nodes with equal best scores the reference keeps the
 * first one it meets while iterating an unordered_set<BAMatrix*> (pointer hash order, :2332-2336, :2442);
 * we iterate sinks in topological order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"

#define NEG (-(1 << 28))
#define LIVE(v) ((v) > NEG / 2)

enum { MM = 0, IC = 1, IR = 2 };       /* match, insert-column (graph base vs gap), insert-row (read base vs gap) */

typedef struct {
    int64_t top, bot;                  /* inclusive diagonals */
    int64_t len, cum;                  /* node length; shortest sequence from any source to the node's left edge */
    int     masked;
    size_t  off;                       /* offset of the node's H x len rectangle in each matrix */
    const char* seq;
} BNode;

typedef struct { int seed; int path_off, path_len; } SeedRef;   /* a non-empty predecessor reached through path[] of empty nodes */

typedef struct {
    const vgk_banded_problem* p;
    const int8_t* mat; const int8_t* qmat;      /* 25 or 256x25 */
    int go, ge;
    int64_t L;
    BNode* nd;
    int32_t *M, *Ic, *Ir;
    /* flattened seeds of the node being looked at */
    SeedRef* seeds; int n_seeds, cap_seeds;
    int* pool; int n_pool, cap_pool;
    int as_source; int src_path_off, src_path_len;
    /* reversed edit runs */
    vgk_op* runs; size_t n_runs, cap_runs;
    struct Stack* st;                  /* alternate tracebacks */
} B;

static int nt5(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': return 3; default: return 4; }
}
/* substitution score of read base r against base j of node n (aligner.cpp scorer tables; fill_matrix :341-347) */
static inline int sub(const B* b, const BNode* n, int64_t r, int64_t j) {
    int g = nt5(n->seq[j]), q = nt5(b->p->read[r]);
    return b->qmat ? b->qmat[25 * (int)b->p->qual[r] + 5 * g + q] : b->mat[5 * g + q];
}
static inline size_t at(const BNode* n, int64_t r, int64_t j) { return n->off + (size_t)(r - j - n->top) * (size_t)n->len + (size_t)j; }
static inline int in_band(const B* b, const BNode* n, int64_t r, int64_t j) {
    return r >= 0 && r < b->L && r - j >= n->top && r - j <= n->bot;
}
static inline int max2(int a, int c) { return a > c ? a : c; }
static inline int best3(const B* b, size_t i) { return max2(max2(b->M[i], b->Ir[i]), b->Ic[i]); }
static inline int open_col(const B* b, size_t i) { return max2(max2(b->M[i] - b->go, b->Ir[i] - b->go), b->Ic[i] - b->ge); }
static inline int open_row(const B* b, size_t i) { return max2(max2(b->M[i] - b->go, b->Ir[i] - b->ge), b->Ic[i] - b->go); }

/* ---- flattened predecessor lists: the LIFO walk both fill_matrix (:305-330) and traceback_over_edge (:1226-1262, :1311-1330) do ---- */
static int pool_push(B* b, const int* src, int n, int extra) {
    if (b->n_pool + n + 1 > b->cap_pool) { b->cap_pool = (b->n_pool + n + 1) * 2 + 64; b->pool = (int*)realloc(b->pool, sizeof(int) * b->cap_pool); }
    int off = b->n_pool;
    for (int i = 0; i < n; ++i) b->pool[b->n_pool++] = src[i];
    if (extra >= 0) b->pool[b->n_pool++] = extra;
    return off;
}
static void flatten_seeds(B* b, int node) {
    const vgk_graph* g = &b->p->graph;
    b->n_seeds = 0; b->n_pool = 0; b->src_path_off = 0; b->src_path_len = 0;
    b->as_source = g->pred_off[node] == g->pred_off[node + 1];
    /* explicit stack of (node, path offset, path length) */
    int cap = 64, top = 0; int* st = (int*)malloc(sizeof(int) * 3 * cap);
    for (uint32_t e = g->pred_off[node]; e < g->pred_off[node + 1]; ++e) {
        if (top == cap) { cap *= 2; st = (int*)realloc(st, sizeof(int) * 3 * cap); }
        st[3 * top] = (int)g->pred_idx[e]; st[3 * top + 1] = 0; st[3 * top + 2] = 0; ++top;
    }
    while (top) {
        --top; int s = st[3 * top], poff = st[3 * top + 1], plen = st[3 * top + 2];
        if (b->nd[s].masked) continue;
        if (b->nd[s].len == 0) {
            int noff = pool_push(b, b->pool + poff, plen, s);
            if (g->pred_off[s] == g->pred_off[s + 1]) { b->as_source = 1; b->src_path_off = noff; b->src_path_len = plen + 1; }
            for (uint32_t e = g->pred_off[s]; e < g->pred_off[s + 1]; ++e) {
                if (top == cap) { cap *= 2; st = (int*)realloc(st, sizeof(int) * 3 * cap); }
                st[3 * top] = (int)g->pred_idx[e]; st[3 * top + 1] = noff; st[3 * top + 2] = plen + 1; ++top;
            }
            continue;
        }
        if (b->n_seeds == b->cap_seeds) { b->cap_seeds = b->cap_seeds * 2 + 16; b->seeds = (SeedRef*)realloc(b->seeds, sizeof(SeedRef) * b->cap_seeds); }
        b->seeds[b->n_seeds].seed = s; b->seeds[b->n_seeds].path_off = poff; b->seeds[b->n_seeds].path_len = plen; ++b->n_seeds;
    }
    free(st);
}

/* ---- fill (fill_matrix :251-742) ---- */
static void fill_node(B* b, int node) {
    BNode* n = &b->nd[node];
    if (n->len == 0) return;
    const int go = b->go, ge = b->ge; const int64_t L = b->L;
    int64_t lo0 = n->top < 0 ? 0 : n->top, hi0 = n->bot >= L ? L - 1 : n->bot;
    for (int64_t r = lo0; r <= hi0; ++r) { b->M[at(n, r, 0)] = NEG; b->Ic[at(n, r, 0)] = NEG; }
    if (lo0 <= hi0) b->Ir[at(n, lo0, 0)] = NEG;
    flatten_seeds(b, node);
    for (int si = 0; si < b->n_seeds; ++si) {
        const BNode* s = &b->nd[b->seeds[si].seed];
        int64_t snt = s->top + s->len, snb = s->bot + s->len;          /* diagonals this seed reaches in column 0 */
        int64_t lo = snt < 0 ? 0 : snt, hi = snb >= L ? L - 1 : snb;
        int64_t ext = s->cum + s->len, sj = s->len - 1;
        if (lo > hi) continue;
        /* first row (:343-377) */
        size_t i = at(n, lo, 0);
        int ms = sub(b, n, lo, 0);
        if (snt < 0) {
            b->M[i]  = max2(b->M[i], ms - go - (int)(ext - 1) * ge);
            b->Ir[i] = max2(b->Ir[i], -2 * go - (int)ext * ge);
        } else if (snt == 0) {
            b->M[i]  = max2(b->M[i], ms - go - (int)(ext - 1) * ge);
        } else {
            b->M[i]  = max2(b->M[i], ms + best3(b, at(s, lo - 1, sj)));
        }
        if (snt < snb) b->Ic[i] = max2(b->Ic[i], open_col(b, at(s, lo, sj)));
        /* interior rows (:379-400) */
        for (int64_t r = lo + 1; r < hi; ++r) {
            i = at(n, r, 0);
            b->M[i]  = max2(b->M[i], sub(b, n, r, 0) + best3(b, at(s, r - 1, sj)));
            b->Ic[i] = max2(b->Ic[i], open_col(b, at(s, r, sj)));
        }
        /* last row (:403-429): the column gap only exists when the band was cut by the bottom of the matrix */
        if (hi != lo) {
            i = at(n, hi, 0);
            b->M[i] = max2(b->M[i], sub(b, n, hi, 0) + best3(b, at(s, hi - 1, sj)));
            if (snb >= L) b->Ic[i] = max2(b->Ic[i], open_col(b, at(s, hi, sj)));
        }
    }
    if (b->as_source) {
        /* implied lead gaps of a source column (:433-476) */
        size_t i = at(n, 0, 0);
        b->M[i] = b->qmat ? max2(b->M[i], sub(b, n, 0, 0)) : sub(b, n, 0, 0);
        b->Ir[i] = max2(b->Ir[i], -2 * go);
        b->Ic[i] = max2(b->Ic[i], -2 * go);
        for (int64_t r = 1; r <= hi0; ++r) {
            i = at(n, r, 0); size_t up = at(n, r - 1, 0);
            b->M[i]  = max2(b->M[i], sub(b, n, r, 0) - go - (int)(r - 1) * ge);
            b->Ir[i] = open_row(b, up);
            b->Ic[i] = max2(b->Ic[i], -2 * go - (int)r * ge);
        }
        b->Ic[i] = NEG;
    } else {
        for (int64_t r = lo0 + 1; r <= hi0; ++r) b->Ir[at(n, r, 0)] = open_row(b, at(n, r - 1, 0));
    }
    /* remaining columns (:492-590) */
    int64_t H = n->bot - n->top + 1;
    for (int64_t j = 1; j < n->len; ++j) {
        int64_t lo = n->top + j < 0 ? 0 : n->top + j, hi = n->bot + j >= L ? L - 1 : n->bot + j;
        if (lo > hi) continue;
        size_t i = at(n, lo, j);
        int ms = sub(b, n, lo, j);
        b->M[i]  = n->top + j <= 0 ? ms - go - (int)(n->cum + j - 1) * ge : ms + best3(b, at(n, lo - 1, j - 1));
        b->Ir[i] = n->top + j <  0 ? -2 * go - (int)(n->cum + j) * ge : NEG;
        b->Ic[i] = H != 1 ? open_col(b, at(n, lo, j - 1)) : NEG;
        for (int64_t r = lo + 1; r < hi; ++r) {
            i = at(n, r, j);
            b->M[i]  = sub(b, n, r, j) + best3(b, at(n, r - 1, j - 1));
            b->Ir[i] = open_row(b, at(n, r - 1, j));
            b->Ic[i] = open_col(b, at(n, r, j - 1));
        }
        if (hi > lo) {
            i = at(n, hi, j);
            b->M[i]  = sub(b, n, hi, j) + best3(b, at(n, hi - 1, j - 1));
            b->Ir[i] = open_row(b, at(n, hi - 1, j));
            b->Ic[i] = n->bot + j >= L ? open_col(b, at(n, hi, j - 1)) : NEG;
        }
    }
}

/* ---- edits (BABuilder :44-205), built back to front ---- */
static void emit(B* b, int node, int op, int inc) {
    if (b->n_runs && b->runs[b->n_runs - 1].node == (uint32_t)node) {
        vgk_op* c = &b->runs[b->n_runs - 1];
        if (c->op == op) { c->len = (uint16_t)(c->len + inc); return; }
        if (c->len == 0 && b->nd[node].len == 0) { c->op = (uint8_t)op; c->len = (uint16_t)inc; return; }   /* empty node: the zero edit is replaced (:69-72) */
    }
    if (b->n_runs == b->cap_runs) { b->cap_runs = b->cap_runs * 2 + 64; b->runs = (vgk_op*)realloc(b->runs, sizeof(vgk_op) * b->cap_runs); }
    vgk_op* c = &b->runs[b->n_runs++];
    c->node = (uint32_t)node; c->op = (uint8_t)op; c->len = (uint16_t)inc; c->pad = 0;
}
static int op_of(int mat) { return mat == MM ? VGK_OP_M : mat == IR ? VGK_OP_I : VGK_OP_D; }

/* ---- the stack of alternate tracebacks (AltTracebackStack :2426-2790) ----
 * A traceback is the list of its deflections: the first names the start (end node, matrix), each later one a cell where the
 * traceback leaves the optimal choice for a named predecessor state.  The stack keeps up to `max` of them in descending score
 * order, equal scores in the order they were found. */
typedef struct { int from_node; int64_t r, j; int to_node, to_mat; } Defl;
typedef struct { Defl* d; int n; int score; int* prefix; int plen; } Trace;
typedef struct Stack {
    Trace* v; int n; int max;
    int cur, cur_defl;              /* the traceback being traced, its next unconsumed deflection */
    int** empty_paths; int* empty_len; int n_empty, next_empty;   /* source-to-sink chains of empty nodes, sink first */
    int empty_score;
} Stack;

static void trace_free(Trace* t) { free(t->d); free(t->prefix); t->d = NULL; t->prefix = NULL; }
static void stack_pop_back(Stack* st) { trace_free(&st->v[st->n - 1]); --st->n; }
/* insert_traceback (:2691-2740) */
static void stack_insert(Stack* st, const Defl* prefix_d, int n_prefix, int score, Defl last, const int* empty_prefix, int plen) {
    int pos = st->n;                                   /* after every element with score >= the new one */
    while (pos > 0 && score > st->v[pos - 1].score) --pos;
    if (st->n && pos == st->n && st->n >= st->max) return;
    st->v = (Trace*)realloc(st->v, sizeof(Trace) * (size_t)(st->n + 1));
    memmove(st->v + pos + 1, st->v + pos, sizeof(Trace) * (size_t)(st->n - pos));
    Trace* t = &st->v[pos];
    t->d = (Defl*)malloc(sizeof(Defl) * (size_t)(n_prefix + 1)); if (n_prefix) memcpy(t->d, prefix_d, sizeof(Defl) * (size_t)n_prefix);
    t->d[n_prefix] = last; t->n = n_prefix + 1; t->score = score;
    t->prefix = (int*)malloc(sizeof(int) * (size_t)(plen + 1)); if (plen) memcpy(t->prefix, empty_prefix, sizeof(int) * (size_t)plen); t->plen = plen;
    ++st->n;
    if (st->n > st->max) stack_pop_back(st);
}
/* propose_deflection (:2671-2689) */
static void propose(B* b, int alt_score, int from_node, int64_t r, int64_t j, int to_node, int to_mat) {
    Stack* st = b->st;
    Trace* c = &st->v[st->cur];
    if (st->cur_defl != c->n) return;                  /* only once the prescribed deflections are used up */
    if (alt_score <= st->v[st->n - 1].score && st->n >= st->max) return;
    Defl last = { from_node, r, j, to_node, to_mat };
    /* copy what stack_insert needs first: the realloc may move the current trace */
    Defl* pd = (Defl*)malloc(sizeof(Defl) * (size_t)(c->n + 1)); memcpy(pd, c->d, sizeof(Defl) * (size_t)c->n);
    int* pp = (int*)malloc(sizeof(int) * (size_t)(c->plen + 1)); if (c->plen) memcpy(pp, c->prefix, sizeof(int) * (size_t)c->plen);
    const int n = c->n, plen = c->plen;
    stack_insert(st, pd, n, alt_score, last, pp, plen);
    free(pd); free(pp);
}
static int at_deflection(const B* b, int node, int64_t r, int64_t j) {
    const Stack* st = b->st; const Trace* c = &st->v[st->cur];
    return st->cur_defl < c->n && c->d[st->cur_defl].from_node == node && c->d[st->cur_defl].r == r && c->d[st->cur_defl].j == j;
}
/* the three predecessor states of a transition, in the reference's order match, insert-col, insert-row (:812): the first
   that explains `cur` is taken, every other live one is proposed as a deflection (:830-886 and its siblings) */
static int pick_and_propose(B* b, size_t i, int cur, int dm, int dc, int dr, int from_node, int64_t r, int64_t j, int to_node) {
    const int S = b->st->v[b->st->cur].score;
    int found = -1;
    { const int src = b->M[i], diff = cur - (src + dm);
      if (diff == 0) found = MM; else if (LIVE(src)) propose(b, S - diff, from_node, r, j, to_node, MM); }
    { const int src = b->Ic[i]; if (LIVE(src)) { const int diff = cur - (src + dc);
      if (found < 0 && diff == 0) found = IC; else propose(b, S - diff, from_node, r, j, to_node, IC); } }
    { const int src = b->Ir[i]; if (LIVE(src)) { const int diff = cur - (src + dr);
      if (found < 0 && diff == 0) found = IR; else propose(b, S - diff, from_node, r, j, to_node, IR); } }
    return found;
}

/* where a deflection across an edge lands: the predecessors are searched in their own order, each through its empty nodes
   depth-first (:1163-1196); the empty nodes on the way are written to path[] */
static int find_deflect_seed(const B* b, int node, int target, int* path, int* plen) {
    const vgk_graph* g = &b->p->graph;
    int cap = 64, top; int* st = (int*)malloc(sizeof(int) * cap);
    int found = 0;
    for (uint32_t e0 = g->pred_off[node]; e0 < g->pred_off[node + 1] && !found; ++e0) {
        top = 0; st[top++] = (int)g->pred_idx[e0]; *plen = 0;
        while (top) {
            const int s = st[--top];
            if (s < 0) { --*plen; continue; }
            if (b->nd[s].masked) continue;
            if (s == target) { found = 1; break; }
            if (b->nd[s].len == 0) {
                path[(*plen)++] = s;
                if (top + 2 + (int)(g->pred_off[s + 1] - g->pred_off[s]) > cap) { cap = cap * 2 + (int)(g->pred_off[s + 1] - g->pred_off[s]); st = (int*)realloc(st, sizeof(int) * cap); }
                st[top++] = -1;
                for (uint32_t e = g->pred_off[s]; e < g->pred_off[s + 1]; ++e) st[top++] = (int)g->pred_idx[e];
            }
        }
    }
    free(st);
    return found;
}

static int traceback(B* b, int node, int mat) {
    const int go = b->go, ge = b->ge;
    Stack* st = b->st;
    const int S = st->v[st->cur].score;
    BNode* n = &b->nd[node];
    int64_t r = b->L - 1, j = n->len - 1;
    int lead = 0;
    for (;;) {
        n = &b->nd[node];
        /* inside the node (:775-1112) */
        while ((j > 0 || mat == IR) && !lead) {
            emit(b, node, op_of(mat), 1);
            if (at_deflection(b, node, r, j)) {           /* (:789-809) */
                if (mat == MM) { --r; --j; } else if (mat == IR) --r; else --j;
                mat = st->v[st->cur].d[st->cur_defl++].to_mat;
                continue;
            }
            if (mat == MM) {
                if (r == 0) { mat = IC; --j; r = -1; lead = 1; break; }
                const int ms = sub(b, n, r, j);
                int src = pick_and_propose(b, at(n, r - 1, j - 1), b->M[at(n, r, j)], ms, ms, ms, node, r, j, node);
                if (src < 0) return VGK_EINVAL;
                mat = src; --r; --j;
            } else if (mat == IR) {
                if (r == 0) { lead = 1; r = -1; break; }
                int src = pick_and_propose(b, at(n, r - 1, j), b->Ir[at(n, r, j)], -go, -go, -ge, node, r, j, node);
                if (src < 0) return VGK_EINVAL;
                mat = src; --r;
            } else {
                int src = pick_and_propose(b, at(n, r, j - 1), b->Ic[at(n, r, j)], -go, -ge, -go, node, r, j, node);
                if (src < 0) return VGK_EINVAL;
                mat = src; --j;
            }
        }
        if (lead) { mat = IC; while (j > 0) { emit(b, node, VGK_OP_D, 1); --j; } }     /* (:1114-1124) */

        /* across the left edge (:1129-1780) */
        if (at_deflection(b, node, r, 0)) {               /* (:1158-1222) */
            emit(b, node, op_of(mat), 1);
            const Defl d = st->v[st->cur].d[st->cur_defl++];
            int* path = (int*)malloc(sizeof(int) * (size_t)(b->p->graph.n_nodes + 1)); int plen = 0;
            if (!find_deflect_seed(b, node, d.to_node, path, &plen)) { free(path); return VGK_EINVAL; }
            for (int k = 0; k < plen; ++k) emit(b, path[k], op_of(mat), 0);
            free(path);
            if (r == 0 && mat == MM) lead = 1;
            if (mat == MM) --r;                             /* a column gap stays in its row */
            mat = d.to_mat; node = d.to_node; j = b->nd[node].len - 1;
            continue;
        }
        flatten_seeds(b, node);
        int found = -1, fmat = MM, flead = lead;
        if (lead) {
            emit(b, node, VGK_OP_D, 1);
            for (int si = 0; si < b->n_seeds; ++si) {
                const BNode* s = &b->nd[b->seeds[si].seed];
                const int diff = (int)((int64_t)ge * (s->cum + s->len - n->cum));
                if (diff == 0 && found < 0) found = si;
                else propose(b, S - diff, node, r, 0, b->seeds[si].seed, IC);
            }
            if (found < 0) {
                if (!b->as_source) return VGK_EINVAL;
                for (int k = 0; k < b->src_path_len; ++k) emit(b, b->pool[b->src_path_off + k], VGK_OP_D, 0);
                return VGK_OK;
            }
        } else {
            emit(b, node, op_of(mat), 1);
            int cur = mat == MM ? b->M[at(n, r, 0)] : b->Ic[at(n, r, 0)];
            int ms = mat == MM ? sub(b, n, r, 0) : 0;
            for (int si = 0; si < b->n_seeds; ++si) {
                const int seed = b->seeds[si].seed;
                const BNode* s = &b->nd[seed];
                int64_t snt = s->top + s->len, snb = s->bot + s->len, sj = s->len - 1;
                if (r > snb - (mat == IC) || r < snt) continue;
                if (mat == MM) {
                    if (r == 0) {          /* the diagonal neighbour is the implied lead-gap row (:1352-1372) */
                        const int diff = cur - (-go - (int)(s->cum + s->len - 1) * ge + ms);
                        if (diff == 0 && found < 0) { found = si; fmat = IC; flead = 1; }
                        else propose(b, S - diff, node, r, 0, seed, IC);
                        continue;
                    }
                    /* (:1374-1430): every state of every predecessor is looked at; the first exact one wins */
                    { const size_t i = at(s, r - 1, sj);
                      { const int src = b->M[i], diff = cur - (src + ms);
                        if (diff == 0 && found < 0) { found = si; fmat = MM; } else if (LIVE(src)) propose(b, S - diff, node, r, 0, seed, MM); }
                      { const int src = b->Ic[i]; if (LIVE(src)) { const int diff = cur - (src + ms);
                        if (diff == 0 && found < 0) { found = si; fmat = IC; } else propose(b, S - diff, node, r, 0, seed, IC); } }
                      { const int src = b->Ir[i]; if (LIVE(src)) { const int diff = cur - (src + ms);
                        if (diff == 0 && found < 0) { found = si; fmat = IR; } else propose(b, S - diff, node, r, 0, seed, IR); } } }
                } else {
                    const size_t i = at(s, r, sj);
                    { const int src = b->M[i], diff = cur - (src - go);
                      if (diff == 0 && found < 0) { found = si; fmat = MM; } else if (LIVE(src)) propose(b, S - diff, node, r, 0, seed, MM); }
                    { const int src = b->Ic[i]; if (LIVE(src)) { const int diff = cur - (src - ge);
                      if (diff == 0 && found < 0) { found = si; fmat = IC; } else propose(b, S - diff, node, r, 0, seed, IC); } }
                    { const int src = b->Ir[i]; if (LIVE(src)) { const int diff = cur - (src - go);
                      if (diff == 0 && found < 0) { found = si; fmat = IR; } else propose(b, S - diff, node, r, 0, seed, IR); } }
                }
            }
            if (found < 0) {
                if (!b->as_source) return VGK_EINVAL;
                int64_t ins;                 /* read bases inserted before the first graph base (:1655-1720) */
                if (mat == MM) { if (cur != (r > 0 ? -go - (int)(r - 1) * ge : 0) + ms) return VGK_EINVAL; ins = r; }
                else           { if (cur != -go - (int)r * ge - go) return VGK_EINVAL; ins = r + 1; }
                for (int k = 0; k < b->src_path_len; ++k) emit(b, b->pool[b->src_path_off + k], VGK_OP_D, 0);
                int end_node = b->src_path_len ? b->pool[b->src_path_off + b->src_path_len - 1] : node;
                for (int64_t k = 0; k < ins; ++k) emit(b, end_node, VGK_OP_I, 1);
                return VGK_OK;
            }
        }
        const SeedRef* sr = &b->seeds[found];
        for (int k = 0; k < sr->path_len; ++k) emit(b, b->pool[sr->path_off + k], op_of(mat), 0);
        if (!lead) { if (mat == MM) --r; mat = fmat; lead = flead; }
        node = sr->seed; j = b->nd[node].len - 1;
    }
}

/* Up to max_alns alignments in descending score order (BandedGlobalAligner::traceback :2329-2423); results[k].ops_begin
   indexes `ops`.  Returns the status of the problem (also in results[0].status). */
int vgo_banded_align_multi(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_banded_problem* p, uint32_t max_alns,
                           vgk_result* results, uint32_t* n_alns, vgk_op* ops, uint32_t ops_cap) {
    vgk_result* res = &results[0];
    *n_alns = 0;
    if (!max_alns) return VGK_EINVAL;
    memset(res, 0, sizeof *res);
    const vgk_graph* g = &p->graph;
    const int N = (int)g->n_nodes; const int64_t L = p->read_len;
    if (!N || !L) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    if (qa && !p->qual) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    for (int v = 0; v < N; ++v) for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e)
        if (g->pred_idx[e] >= (uint32_t)v) { res->status = VGK_EINVAL; return VGK_EINVAL; }      /* not in topological order */
    /* vgk_op.len is 16 bits (include/vgk.h): the ABI cannot express a longer run, so both sides refuse such problems */
    for (int v = 0; v < N; ++v) if (g->node_len[v] > 65535u) { res->status = VGK_ETOOBIG; return VGK_ETOOBIG; }
    if (L > 65535) { res->status = VGK_ETOOBIG; return VGK_ETOOBIG; }
    B b; memset(&b, 0, sizeof b);
    b.p = p; b.mat = sc->matrix; b.qmat = qa ? qa->matrix : NULL; b.go = sc->gap_open; b.ge = sc->gap_extend; b.L = L;
    b.nd = (BNode*)calloc((size_t)N, sizeof(BNode));
    int rc = VGK_OK;
    /* successor lists */
    uint32_t* succ_off = (uint32_t*)calloc((size_t)N + 1, sizeof(uint32_t));
    uint32_t* succ = (uint32_t*)malloc(sizeof(uint32_t) * (g->pred_off[N] + 1));
    for (int v = 0; v < N; ++v) for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e) ++succ_off[g->pred_idx[e] + 1];
    for (int v = 0; v < N; ++v) succ_off[v + 1] += succ_off[v];
    { uint32_t* fillp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(N + 1)); memcpy(fillp, succ_off, sizeof(uint32_t) * (size_t)(N + 1));
      for (int v = 0; v < N; ++v) for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e) succ[fillp[g->pred_idx[e]]++] = (uint32_t)v;
      free(fillp); }
    size_t soff = 0;
    for (int v = 0; v < N; ++v) { b.nd[v].len = g->node_len[v]; b.nd[v].seq = g->seq + soff; soff += g->node_len[v]; }
    /* path lengths to sinks (:2122-2170) */
    int64_t* shortest = (int64_t*)malloc(sizeof(int64_t) * (size_t)N); int64_t* longest = (int64_t*)calloc((size_t)N, sizeof(int64_t));
    for (int v = 0; v < N; ++v) shortest[v] = succ_off[v] == succ_off[v + 1] ? 0 : INT64_MAX;
    for (int v = N - 1; v >= 0; --v) {
        int64_t lg = longest[v] + b.nd[v].len, sh = shortest[v] + b.nd[v].len;
        for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e) {
            uint32_t u = g->pred_idx[e];
            if (longest[u] < lg) longest[u] = lg;
            if (shortest[u] > sh) shortest[u] = sh;
        }
    }
    /* band ends and masking (:2174-2268) */
    const int permissive = (p->flags & VGK_BANDED_PERMISSIVE) != 0; const int64_t pad = p->band_padding;
    for (int v = 0; v < N; ++v) { b.nd[v].top = INT64_MAX; b.nd[v].bot = INT64_MIN; }
    for (int v = 0; v < N; ++v) if (g->pred_off[v] == g->pred_off[v + 1]) {
        if (permissive) {
            int64_t t = L - (b.nd[v].len + longest[v]) - pad, u = L - (b.nd[v].len + shortest[v]) + pad;
            b.nd[v].top = t < -pad ? t : -pad; b.nd[v].bot = u > pad ? u : pad;
        } else { b.nd[v].top = -pad; b.nd[v].bot = pad; }
    }
    uint64_t total_cells = 0;
    for (int v = 0; v < N; ++v) {
        BNode* n = &b.nd[v];
        if (n->top > n->bot) { n->masked = 1; continue; }          /* never reached from an unmasked node */
        int64_t et = n->top + n->len, eb = n->bot + n->len;
        if (et + shortest[v] > L || eb + longest[v] < L) { n->masked = 1; continue; }
        for (uint32_t e = succ_off[v]; e < succ_off[v + 1]; ++e) {
            BNode* o = &b.nd[succ[e]];
            if (et < o->top) o->top = et;
            if (eb > o->bot) o->bot = eb;
        }
        total_cells += (uint64_t)(n->bot - n->top + 1) * (uint64_t)n->len;
    }
    if (p->max_cells && total_cells > p->max_cells) { rc = VGK_ETOOBIG; goto done; }
    /* shortest sequence leading to each node (:2271-2293) */
    for (int v = 0; v < N; ++v) b.nd[v].cum = g->pred_off[v] == g->pred_off[v + 1] ? 0 : INT64_MAX;
    for (int v = 0; v < N; ++v) {
        int64_t through = b.nd[v].cum + b.nd[v].len;
        for (uint32_t e = succ_off[v]; e < succ_off[v + 1]; ++e) if (through < b.nd[succ[e]].cum) b.nd[succ[e]].cum = through;
    }
    if (!permissive) {
        int any = 0;
        for (int v = 0; v < N; ++v) if (succ_off[v] == succ_off[v + 1] && !b.nd[v].masked) any = 1;
        if (!any) { rc = VGK_ENOBAND; goto done; }
    }
    {
        size_t cells = 0;
        for (int v = 0; v < N; ++v) if (!b.nd[v].masked) { b.nd[v].off = cells; cells += (size_t)(b.nd[v].bot - b.nd[v].top + 1) * (size_t)b.nd[v].len; }
        b.M = (int32_t*)malloc(sizeof(int32_t) * (cells + 1)); b.Ic = (int32_t*)malloc(sizeof(int32_t) * (cells + 1)); b.Ir = (int32_t*)malloc(sizeof(int32_t) * (cells + 1));
        if (!b.M || !b.Ic || !b.Ir) { rc = VGK_ENOMEM; goto done; }
        for (size_t i = 0; i <= cells; ++i) { b.M[i] = NEG; b.Ic[i] = NEG; b.Ir[i] = NEG; }
    }
    for (int v = 0; v < N; ++v) if (!b.nd[v].masked) fill_node(&b, v);

    /* the stack starts with the end cells (:2426-2563): sinks in topological order [PARITY-UNPINNED: the reference walks an
       unordered_set of matrix pointers], looking through empty sinks to their predecessors; per end node match, insert-row,
       insert-col (:2522-2552).  Chains of empty nodes from a source to a sink are kept aside: each is the whole read inserted. */
    {
        Stack st; memset(&st, 0, sizeof st); st.max = (int)max_alns; st.empty_score = -b.go - (int)(L - 1) * b.ge;
        b.st = &st;
        int cap = 64; int* stk = (int*)malloc(sizeof(int) * cap); int* path = (int*)malloc(sizeof(int) * (size_t)(N + 1)); int plen = 0;
        for (int v = 0; v < N; ++v) {
            if (succ_off[v] != succ_off[v + 1] || b.nd[v].masked) continue;
            int top = 0; stk[top++] = v; plen = 0;
            while (top) {
                int u = stk[--top];
                if (u < 0) { --plen; continue; }
                if (b.nd[u].masked) continue;
                if (b.nd[u].len == 0) {
                    path[plen++] = u;
                    if (top + 2 + (int)(g->pred_off[u + 1] - g->pred_off[u]) > cap) { cap = cap * 2 + (int)(g->pred_off[u + 1] - g->pred_off[u]); stk = (int*)realloc(stk, sizeof(int) * cap); }
                    stk[top++] = -1;
                    if (g->pred_off[u] == g->pred_off[u + 1]) {
                        st.empty_paths = (int**)realloc(st.empty_paths, sizeof(int*) * (size_t)(st.n_empty + 1)); st.empty_len = (int*)realloc(st.empty_len, sizeof(int) * (size_t)(st.n_empty + 1));
                        st.empty_paths[st.n_empty] = (int*)malloc(sizeof(int) * (size_t)plen); memcpy(st.empty_paths[st.n_empty], path, sizeof(int) * (size_t)plen);
                        st.empty_len[st.n_empty++] = plen;
                        continue;
                    }
                    for (uint32_t e = g->pred_off[u]; e < g->pred_off[u + 1]; ++e) stk[top++] = (int)g->pred_idx[e];
                    continue;
                }
                const BNode* n = &b.nd[u];
                if (!in_band(&b, n, L - 1, n->len - 1)) continue;
                size_t i = at(n, L - 1, n->len - 1);
                const int cand[3] = { b.M[i], b.Ir[i], b.Ic[i] }; const int cmat[3] = { MM, IR, IC };
                for (int k = 0; k < 3; ++k) if (LIVE(cand[k])) {
                    Defl start = { u, L - 1, n->len - 1, u, cmat[k] };
                    stack_insert(&st, NULL, 0, cand[k], start, path, plen);
                }
            }
        }
        free(stk); free(path);
        /* one alignment per turn (:2346-2421) */
        size_t n_out = 0; uint32_t made = 0;
        if (!st.n && !st.n_empty) rc = VGK_ENOBAND;
        while (rc == VGK_OK && (st.cur < st.n || st.next_empty < st.n_empty)) {
            vgk_result* out = &results[made]; memset(out, 0, sizeof *out); out->ops_begin = (uint32_t)n_out;
            const int take_empty = st.cur >= st.n ? 1 : (st.empty_score >= st.v[st.cur].score && st.next_empty < st.n_empty);
            if (take_empty) {
                /* the read is one insertion on the first node of the chain (next_empty_alignment :2616-2668) */
                const int* ep = st.empty_paths[st.next_empty]; const int el = st.empty_len[st.next_empty]; ++st.next_empty;
                out->score = st.empty_score;
                if (n_out + (size_t)el > ops_cap) { rc = VGK_EOPS; break; }
                for (int k = el - 1; k >= 0; --k) {      /* path[] runs sink-first; the alignment runs source-first */
                    vgk_op o; o.node = (uint32_t)ep[k]; o.pad = 0;
                    if (k == el - 1) { o.op = VGK_OP_I; o.len = (uint16_t)L; } else { o.op = VGK_OP_M; o.len = 0; }
                    ops[n_out++] = o;
                }
                --st.max;
                if (st.n > st.max) { if (st.cur == st.n - 1) { stack_pop_back(&st); st.next_empty = st.n_empty; } else stack_pop_back(&st); }
            } else {
                const Trace* t = &st.v[st.cur];
                st.cur_defl = 1;                           /* the first deflection names the start (:2574-2587) */
                b.n_runs = 0;
                rc = traceback(&b, t->d[0].from_node, t->d[0].to_mat);
                if (rc != VGK_OK) break;
                t = &st.v[st.cur];
                out->score = t->score;
                if (n_out + b.n_runs + (size_t)t->plen > ops_cap) { rc = VGK_EOPS; break; }
                for (size_t k = b.n_runs; k-- > 0;) { vgk_op o = b.runs[k]; if (o.len == 0) o.op = VGK_OP_M; ops[n_out++] = o; }
                for (int k = t->plen - 1; k >= 0; --k) { vgk_op o; o.node = (uint32_t)t->prefix[k]; o.op = VGK_OP_M; o.len = 0; o.pad = 0; ops[n_out++] = o; }
                ++st.cur;
                if (st.cur >= st.n) st.next_empty = st.n_empty;          /* (:2604-2608) */
            }
            out->n_ops = (uint32_t)(n_out - out->ops_begin); out->status = VGK_OK;
            ++made;
            if (made >= max_alns) break;
        }
        if (rc == VGK_OK && !made) rc = VGK_ENOBAND;
        *n_alns = rc == VGK_OK ? made : 0;
        for (int k = 0; k < st.n; ++k) trace_free(&st.v[k]);
        free(st.v);
        for (int k = 0; k < st.n_empty; ++k) free(st.empty_paths[k]);
        free(st.empty_paths); free(st.empty_len);
        b.st = NULL;
    }
done:
    if (rc != VGK_OK) { memset(res, 0, sizeof *res); res->status = rc; }
    free(b.M); free(b.Ic); free(b.Ir); free(b.nd); free(b.seeds); free(b.pool); free(b.runs);
    free(succ_off); free(succ); free(shortest); free(longest);
    return rc;
}

int vgo_banded_align(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_banded_problem* p,
                     vgk_result* res, vgk_op* ops, uint32_t ops_cap) {
    uint32_t n = 0;
    return vgo_banded_align_multi(sc, qa, p, 1, res, &n, ops, ops_cap);
}
