/*
 * vgo_gapless.c — CPU ORACLE for haplotype-consistent gapless seed extension.
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * Restates GaplessExtender::extend and its helpers from the reference's src/gbwt_extender.cpp:
 *   set_score :201-209, match_initial / match_forward / match_backward :213-296, handle_full_length :301-329,
 *   remove_duplicates :332-365, find_mismatches :368-387, trim_mismatches :421-529, extend :533-737,
 *   GaplessExtension::contains / overlap :17-103.
 *
 * The haplotype index.  The reference walks a GBWTGraph (gbwt + gbwtgraph: un-vendored submodules, absent from the
 * snapshot).  What the extender needs from it is small — node sequences in both orientations, a bidirectional search
 * state (node + range of visits on each strand), `follow_paths` = the non-empty one-node extensions of a state in
 * the order of the node's outgoing edges, `bd_find`, and the number of haplotype visits a state covers — and is
 * restated here from the published GBWT design (Siren et al. 2020) [prior knowledge]: every thread is indexed in both
 * orientations; the visits of an oriented node are ordered by (predecessor node, rank within the predecessor's record),
 * threads that start at the node first, in thread order; extending a range [sp, ep] with successor w maps it to
 * offset(v -> w) + rank_w(body[0, sp)) ...; the opposite strand's range shrinks by the number of visits in the range
 * whose successor x has reverse(x) < reverse(w).  Records are built directly in that order (no compression), in a
 * topological pass when the threads are acyclic as oriented-node sequences and by prefix doubling otherwise.
 *
 * Parity status: pinned on the reference's known-answer tests for this path (src/unittest/gbwt_extender.cpp:576-1158,
 * hand-transcribed in tests/test_gapless.py).  PARITY-UNPINNED: (i) the order in which the seeds of a cluster are
 * visited (the reference iterates a hash set, gbwt_extender.hpp:143) — here: the order given; (ii) the order std::sort
 * leaves equal elements in (handle_full_length, remove_duplicates) — here: stable; (iii) the numeric values of the
 * opposite-strand ranges, which only break ties in remove_duplicates' sort.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"
#include "vgo_haplo.h"

static char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}

uint64_t vgk_haplo_run_nodes(const vgk_haplo* index) { return index ? index->n_nodes : 0; }
uint64_t vgk_haplo_search_nodes(const vgk_haplo* index) { return index ? index->n_nodes : 0; }      /* (the engine's search may walk merged runs; the oracle's walks the nodes) */
void vgo_haplo_destroy(vgk_haplo* h) {
    if (!h) return;
    free(h->len); free(h->seq_off); free(h->seq); free(h->count); free(h->edge_off); free(h->edge_to); free(h->edge_base);
    free(h->body_off); free(h->body); free(h);
}

typedef struct { int32_t pred; uint32_t order, seq, pos; } Arrival;
static int cmp_arrival(const void* a, const void* b) {
    const Arrival* x = (const Arrival*)a; const Arrival* y = (const Arrival*)b;
    if (x->pred != y->pred) return x->pred < y->pred ? -1 : 1;
    return x->order < y->order ? -1 : x->order > y->order;
}
static int cmp_i32(const void* a, const void* b) { const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return x < y ? -1 : x > y; }

/* Threads that revisit a node (cycles) have no topological order of records.  The general rule is the same one — the
   visits of a node are ordered by their reversed prefixes (predecessor, its predecessor, ..., thread start; starts by thread
   number) — and is evaluated here by prefix doubling over all visits.  Fills `arr` (record order), `succ` and the number of
   distinct successors like the topological pass does. */
typedef struct { uint64_t key; uint32_t v; } RankKey;
static int cmp_rankkey(const void* a, const void* b) {
    const RankKey* x = (const RankKey*)a; const RankKey* y = (const RankKey*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->v < y->v ? -1 : x->v > y->v;
}
static void order_visits_general(uint32_t S, const uint32_t* soff, const int32_t* sn, uint32_t V, uint32_t O, const size_t* body_off,
                                 const uint32_t* count, Arrival* arr, int32_t* succ, uint32_t* e_cnt) {
    uint32_t* seq_of = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    uint32_t* rank = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    uint32_t* next = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    RankKey* rk = (RankKey*)malloc(sizeof(RankKey) * (V + 1));
    uint32_t maxlen = 0;
    for (uint32_t s = 0; s < S; ++s) {
        if (soff[s + 1] - soff[s] > maxlen) maxlen = soff[s + 1] - soff[s];
        for (uint32_t v = soff[s]; v < soff[s + 1]; ++v) { seq_of[v] = s; rank[v] = v == soff[s] ? s : S + (uint32_t)sn[v - 1]; }
    }
    for (uint32_t hstep = 1; hstep <= maxlen; hstep *= 2) {
        for (uint32_t v = 0; v < V; ++v) {
            const uint32_t k = v - soff[seq_of[v]];
            rk[v].key = ((uint64_t)rank[v] << 32) | (k >= hstep ? (uint64_t)rank[v - hstep] + 1u : 0u);
            rk[v].v = v;
        }
        qsort(rk, V, sizeof(RankKey), cmp_rankkey);
        uint32_t distinct = 0;
        for (uint32_t i = 0; i < V; ++i) { if (i && rk[i].key != rk[i - 1].key) ++distinct; next[rk[i].v] = distinct; }
        memcpy(rank, next, sizeof(uint32_t) * V);
        if (distinct + 1 == V) break;
    }
    for (uint32_t v = 0; v < V; ++v) { rk[v].key = ((uint64_t)(uint32_t)sn[v] << 32) | rank[v]; rk[v].v = v; }
    qsort(rk, V, sizeof(RankKey), cmp_rankkey);          /* by node, then by rank: the records, one after the other */
    for (uint32_t i = 0; i < V; ++i) {
        const uint32_t v = rk[i].v, s = seq_of[v], k = v - soff[s];
        arr[i] = (Arrival){ k ? sn[v - 1] : -1, 0, s, k };
        succ[i] = v + 1 < soff[s + 1] ? sn[v + 1] : -1;
    }
    for (uint32_t o = 0; o < O; ++o) {
        const uint32_t n = count[o]; e_cnt[o] = 0; if (!n) continue;
        int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * n); memcpy(tmp, succ + body_off[o], sizeof(int32_t) * n);
        qsort(tmp, n, sizeof(int32_t), cmp_i32);
        for (uint32_t i = 0; i < n; ++i) if (!i || tmp[i] != tmp[i - 1]) ++e_cnt[o];
        free(tmp);
    }
    free(seq_of); free(rank); free(next); free(rk);
}

int vgo_haplo_create(const vgk_haplotypes* d, vgk_haplo** out) {
    if (!d || !out || !d->n_nodes || !d->node_len || !d->seq || (d->n_threads && (!d->thread_off || !d->thread_nodes))) return VGK_EINVAL;
    const uint32_t N = d->n_nodes, O = 2 * N, S = 2 * d->n_threads;
    for (uint32_t t = 0; t < d->n_threads; ++t) for (uint32_t k = d->thread_off[t]; k < d->thread_off[t + 1]; ++k) if (d->thread_nodes[k] >= O) return VGK_EINVAL;
    vgk_haplo* h = (vgk_haplo*)calloc(1, sizeof *h);
    h->n_nodes = N; h->n_oriented = O;
    h->len = (uint32_t*)malloc(sizeof(uint32_t) * O); h->seq_off = (size_t*)malloc(sizeof(size_t) * O);
    size_t total = 0; for (uint32_t i = 0; i < N; ++i) total += d->node_len[i];
    h->seq = (char*)malloc(2 * total + 1);
    { size_t at = 0, rat = total;
      for (uint32_t i = 0; i < N; ++i) {
          const uint32_t L = d->node_len[i];
          h->len[2 * i] = h->len[2 * i + 1] = L; h->seq_off[2 * i] = at; h->seq_off[2 * i + 1] = rat;
          memcpy(h->seq + at, d->seq + at, L);
          for (uint32_t k = 0; k < L; ++k) h->seq[rat + k] = comp(d->seq[at + L - 1 - k]);
          at += L; rat += L;
      } }
    /* sequences: thread t forward = 2t, its reverse complement = 2t + 1 */
    uint32_t* soff = (uint32_t*)malloc(sizeof(uint32_t) * (S + 1)); soff[0] = 0;
    for (uint32_t t = 0; t < d->n_threads; ++t) { const uint32_t n = d->thread_off[t + 1] - d->thread_off[t]; soff[2 * t + 1] = soff[2 * t] + n; soff[2 * t + 2] = soff[2 * t + 1] + n; }
    const uint32_t V = soff[S];
    int32_t* sn = (int32_t*)malloc(sizeof(int32_t) * (V + 1));
    for (uint32_t t = 0; t < d->n_threads; ++t) {
        const uint32_t n = d->thread_off[t + 1] - d->thread_off[t];
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t o = d->thread_nodes[d->thread_off[t] + k];
            sn[soff[2 * t] + k] = (int32_t)o; sn[soff[2 * t + 1] + (n - 1 - k)] = (int32_t)(o ^ 1u);
        }
    }
    h->count = (uint32_t*)calloc(O, sizeof(uint32_t));
    for (uint32_t i = 0; i < V; ++i) ++h->count[sn[i]];
    h->body_off = (size_t*)malloc(sizeof(size_t) * (O + 1)); h->body_off[0] = 0;
    for (uint32_t o = 0; o < O; ++o) h->body_off[o + 1] = h->body_off[o] + h->count[o];
    h->body = (uint32_t*)malloc(sizeof(uint32_t) * (V + 1));
    /* every node collects its arrivals (predecessor, rank in the predecessor's record); a node is laid down once all of
       them are in, i.e. in a topological order of the oriented-node sequences */
    Arrival* arr = (Arrival*)malloc(sizeof(Arrival) * (V + 1));
    uint32_t* got = (uint32_t*)calloc(O, sizeof(uint32_t));
    for (uint32_t s = 0; s < S; ++s) if (soff[s + 1] > soff[s]) {
        const int32_t o = sn[soff[s]];
        arr[h->body_off[o] + got[o]++] = (Arrival){ -1, s, s, 0 };
    }
    uint32_t* queue = (uint32_t*)malloc(sizeof(uint32_t) * (O + 1)); uint32_t qh = 0, qt = 0;
    for (uint32_t o = 0; o < O; ++o) if (h->count[o] && got[o] == h->count[o]) queue[qt++] = o;
    int32_t* succ = (int32_t*)malloc(sizeof(int32_t) * (V + 1));         /* successor of each visit, record order */
    uint32_t* e_cnt = (uint32_t*)calloc(O + 1, sizeof(uint32_t));
    uint32_t laid = 0, with_visits = 0;
    for (uint32_t o = 0; o < O; ++o) if (h->count[o]) ++with_visits;
    while (qh < qt) {
        const uint32_t o = queue[qh++]; ++laid;
        Arrival* a = arr + h->body_off[o]; const uint32_t n = h->count[o];
        qsort(a, n, sizeof(Arrival), cmp_arrival);
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t slen = soff[a[i].seq + 1] - soff[a[i].seq];
            const int32_t w = a[i].pos + 1 < slen ? sn[soff[a[i].seq] + a[i].pos + 1] : -1;
            succ[h->body_off[o] + i] = w;
            if (w < 0) continue;
            arr[h->body_off[w] + got[w]++] = (Arrival){ (int32_t)o, i, a[i].seq, a[i].pos + 1 };
            if (got[w] == h->count[w]) queue[qt++] = (uint32_t)w;
        }
        /* distinct successors */
        int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (n + 1)); memcpy(tmp, succ + h->body_off[o], sizeof(int32_t) * n);
        qsort(tmp, n, sizeof(int32_t), cmp_i32);
        uint32_t k = 0; for (uint32_t i = 0; i < n; ++i) if (!i || tmp[i] != tmp[i - 1]) ++k;
        e_cnt[o] = k; free(tmp);
    }
    int rc = VGK_OK;
    if (laid != with_visits) order_visits_general(S, soff, sn, V, O, h->body_off, h->count, arr, succ, e_cnt);   /* a cycle among the threads */
    if (rc == VGK_OK) {
        h->edge_off = (uint32_t*)malloc(sizeof(uint32_t) * (O + 1)); h->edge_off[0] = 0;
        for (uint32_t o = 0; o < O; ++o) h->edge_off[o + 1] = h->edge_off[o] + e_cnt[o];
        const uint32_t E = h->edge_off[O];
        h->edge_to = (int32_t*)malloc(sizeof(int32_t) * (E + 1)); h->edge_base = (uint32_t*)calloc(E + 1, sizeof(uint32_t));
        for (uint32_t o = 0; o < O; ++o) {
            const uint32_t n = h->count[o]; if (!n) continue;
            int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * n); memcpy(tmp, succ + h->body_off[o], sizeof(int32_t) * n);
            qsort(tmp, n, sizeof(int32_t), cmp_i32);
            uint32_t k = 0; int32_t* et = h->edge_to + h->edge_off[o];
            for (uint32_t i = 0; i < n; ++i) if (!i || tmp[i] != tmp[i - 1]) et[k++] = tmp[i];
            free(tmp);
            for (uint32_t i = 0; i < n; ++i) { uint32_t e = 0; while (et[e] != succ[h->body_off[o] + i]) ++e; h->body[h->body_off[o] + i] = e; }
        }
        /* where the visits coming from o start in w's record: arrivals of w are sorted by predecessor */
        for (uint32_t w = 0; w < O; ++w) {
            const Arrival* a = arr + h->body_off[w];
            for (uint32_t i = 0; i < h->count[w]; ++i) if (a[i].pred >= 0 && (!i || a[i - 1].pred != a[i].pred)) {
                const uint32_t o = (uint32_t)a[i].pred; uint32_t e = h->edge_off[o];
                while (h->edge_to[e] != (int32_t)w) ++e;
                h->edge_base[e] = i;
            }
        }
    }
    free(soff); free(sn); free(arr); free(got); free(queue); free(succ); free(e_cnt);
    if (rc != VGK_OK) { vgo_haplo_destroy(h); return rc; }
    *out = h;
    return VGK_OK;
}

/* ---- search states ---- */
static int sempty(SState s) { return s.lo > s.hi; }
static SState sfind(const vgk_haplo* h, int32_t node) { SState s = { node, 0, (int32_t)h->count[node] - 1 }; return s; }
static BState bd_find_node(const vgk_haplo* h, int32_t node) { BState b = { sfind(h, node), sfind(h, node ^ 1) }; return b; }
static int32_t rkey(int32_t x) { return x < 0 ? -1 : (x ^ 1); }
/* extend the forward strand with successor `to`; the opposite strand's range shrinks accordingly */
static BState bd_extend_forward(const vgk_haplo* h, BState s, int32_t to) {
    const uint32_t o = (uint32_t)s.f.node;
    const uint32_t* body = h->body + h->body_off[o]; const int32_t* et = h->edge_to + h->edge_off[o];
    const uint32_t ne = h->edge_off[o + 1] - h->edge_off[o];
    uint32_t e = 0; while (e < ne && et[e] != to) ++e;
    BState r = s; r.f.node = to;
    if (e == ne || sempty(s.f)) { r.f.lo = 0; r.f.hi = -1; r.b.hi = r.b.lo - 1; return r; }
    int32_t before = 0, inside = 0, rev_off = 0;
    for (int32_t i = 0; i <= s.f.hi; ++i) {
        if (body[i] == e) { if (i < s.f.lo) ++before; else ++inside; }
        else if (i >= s.f.lo && rkey(et[body[i]]) < rkey(to)) ++rev_off;
    }
    r.f.lo = (int32_t)h->edge_base[h->edge_off[o] + e] + before; r.f.hi = r.f.lo + inside - 1;
    r.b.lo = s.b.lo + rev_off; r.b.hi = r.b.lo + inside - 1;
    return r;
}
static BState bd_flip(BState s) { BState r = { s.b, s.f }; return r; }
static uint32_t bd_size(BState s) { return sempty(s.f) ? 0u : (uint32_t)(s.f.hi - s.f.lo + 1); }

/* ---- extensions ---- */
typedef struct {
    int32_t* path; uint32_t path_len, path_cap;
    uint32_t offset; BState state; uint32_t r0, r1;       /* read interval [r0, r1) */
    uint32_t* mism; uint32_t n_mism;
    int32_t score; int left_full, right_full, left_max, right_max; uint32_t internal, old;
} Ext;
typedef struct { Ext e; uint64_t number; } QItem;

static void ext_free(Ext* e) { free(e->path); free(e->mism); e->path = NULL; e->mism = NULL; }
static Ext ext_copy_meta(const Ext* c) { Ext n = *c; n.path = NULL; n.path_len = n.path_cap = 0; n.mism = NULL; n.n_mism = 0; return n; }
static void set_path(Ext* e, const int32_t* first, uint32_t n_first, int32_t extra, int front) {
    e->path = (int32_t*)malloc(sizeof(int32_t) * (n_first + 1)); e->path_len = e->path_cap = n_first + 1;
    if (front) { e->path[0] = extra; memcpy(e->path + 1, first, sizeof(int32_t) * n_first); }
    else { memcpy(e->path, first, sizeof(int32_t) * n_first); e->path[n_first] = extra; }
}
static int ext_full(const Ext* e) { return e->left_full && e->right_full; }
static uint32_t ext_len(const Ext* e) { return e->r1 - e->r0; }

typedef struct { const vgk_haplo* h; const char* seq; uint32_t L; int match, mismatch, bonus; } G;
static const char* nseq(const G* g, int32_t o) { return g->h->seq + g->h->seq_off[o]; }

static void set_score(const G* g, Ext* e) {                               /* :201-209 */
    e->score = (int32_t)(ext_len(e) * (uint32_t)g->match) - (int32_t)(e->internal * (uint32_t)(g->match + g->mismatch))
             + e->left_full * g->bonus + e->right_full * g->bonus;
}
static void match_initial(const G* g, Ext* m) {                          /* :213-237 */
    const char* t = nseq(g, m->path[0]); uint32_t node_offset = m->offset;
    uint32_t left = g->L - m->r1 < g->h->len[m->path[0]] - node_offset ? g->L - m->r1 : g->h->len[m->path[0]] - node_offset;
    while (left--) { if (g->seq[m->r1] != t[node_offset]) ++m->internal; ++m->r1; ++node_offset; }
    m->old = m->internal;
}
static uint32_t match_forward(const G* g, Ext* m, int32_t node, uint32_t limit) {     /* :239-266 */
    const char* t = nseq(g, node); uint32_t node_offset = 0;
    uint32_t left = g->L - m->r1 < g->h->len[node] ? g->L - m->r1 : g->h->len[node];
    while (left--) {
        if (g->seq[m->r1] != t[node_offset]) { if (m->internal + 1 >= limit) return node_offset; ++m->internal; }
        ++m->r1; ++node_offset;
    }
    return node_offset;
}
static void match_backward(const G* g, Ext* m, int32_t node, uint32_t limit) {        /* :268-296 */
    const char* t = nseq(g, node);
    uint32_t left = m->r0 < m->offset ? m->r0 : m->offset;
    while (left--) {
        if (g->seq[m->r0 - 1] != t[m->offset - 1]) { if (m->internal + 1 >= limit) return; ++m->internal; }
        --m->r0; --m->offset;
    }
}
/* does the extension pass through the seed (handle, read_offset - node_offset)?  (:17-41) */
static int ext_contains(const G* g, const Ext* e, int32_t node, int64_t diff) {
    uint32_t read_offset = e->r0, node_offset = e->offset;
    for (uint32_t i = 0; i < e->path_len; ++i) {
        const uint32_t a = g->h->len[e->path[i]] - node_offset, b = e->r1 - read_offset; const uint32_t len = a < b ? a : b;
        if (e->path[i] == node && (int64_t)read_offset - (int64_t)node_offset == diff) return 1;
        read_offset += len; node_offset = 0;
    }
    return 0;
}
static uint32_t ext_overlap(const G* g, const Ext* x, const Ext* y) {     /* :69-103 */
    uint32_t result = 0, xp = x->r0, yp = y->r0, xi = 0, yi = 0, xo = x->offset, yo = y->offset;
    while (xp < x->r1 && yp < y->r1) {
        if (xp == yp && x->path[xi] == y->path[yi] && xo == yo) {
            uint32_t len = g->h->len[x->path[xi]] - xo;
            if (x->r1 - xp < len) len = x->r1 - xp;
            if (y->r1 - yp < len) len = y->r1 - yp;
            result += len; xp += len; yp += len; ++xi; ++yi; xo = yo = 0;
        } else if (xp <= yp) { xp += g->h->len[x->path[xi]] - xo; ++xi; xo = 0; }
        else { yp += g->h->len[y->path[yi]] - yo; ++yi; yo = 0; }
    }
    return result;
}
static BState bd_find_path(const vgk_haplo* h, const int32_t* path, uint32_t n) {
    BState s = bd_find_node(h, path[0]);
    for (uint32_t i = 1; i < n; ++i) s = bd_extend_forward(h, s, path[i]);
    return s;
}
static void find_mismatches(const G* g, Ext* e) {                          /* :368-387 */
    if (!e->internal) return;
    e->mism = (uint32_t*)malloc(sizeof(uint32_t) * (e->internal + 1)); e->n_mism = 0;
    uint32_t node_offset = e->offset, read_offset = e->r0;
    for (uint32_t i = 0; i < e->path_len; ++i) {
        const char* t = nseq(g, e->path[i]); const uint32_t tl = g->h->len[e->path[i]];
        while (node_offset < tl && read_offset < e->r1) { if (t[node_offset] != g->seq[read_offset]) e->mism[e->n_mism++] = read_offset; ++node_offset; ++read_offset; }
        node_offset = 0;
    }
}
static int trim_mismatches(const G* g, Ext* e) {                           /* :421-529 */
    if (!e->n_mism) return 0;
    uint32_t mi = 0;
    uint32_t c0 = e->r0, c1 = e->mism[0];
    int32_t cur = (int32_t)(c1 - c0) * g->match + (e->left_full ? g->bonus : 0);
    uint32_t b0 = c0, b1 = c1; int32_t best = cur;
    while (mi < e->n_mism) {
        if (cur >= g->mismatch) { ++c1; cur -= g->mismatch; }
        else { c0 = c1 = e->mism[mi] + 1; cur = 0; }
        ++mi;
        if (mi == e->n_mism) { cur += (int32_t)(e->r1 - c1) * g->match; c1 = e->r1; if (e->right_full) cur += g->bonus; }
        else { cur += (int32_t)(e->mism[mi] - c1) * g->match; c1 = e->mism[mi]; }
        if (cur > best || (cur > 0 && cur == best && c1 - c0 > b1 - b0)) { b0 = c0; b1 = c1; best = cur; }
    }
    if (b0 == e->r0 && b1 == e->r1) return 0;
    if (b1 == b0) { e->path_len = 0; e->r0 = b0; e->r1 = b1; e->n_mism = 0; e->score = 0; e->left_full = e->right_full = 0; return 1; }
    if (b0 > e->r0) e->left_full = 0;
    if (b1 < e->r1) e->right_full = 0;
    uint32_t node_offset = e->offset, read_offset = e->r0;
    e->r0 = b0; e->r1 = b1; e->score = best;
    uint32_t head = 0;
    while (head < e->path_len) {
        const uint32_t nl = g->h->len[e->path[head]];
        read_offset += nl - node_offset; node_offset = 0;
        if (read_offset > e->r0) { e->offset = nl - (read_offset - e->r0); break; }
        ++head;
    }
    uint32_t tail = head + 1;
    while (read_offset < e->r1) { read_offset += g->h->len[e->path[tail]]; ++tail; }
    if (head > 0 || tail < e->path_len) {
        memmove(e->path, e->path + head, sizeof(int32_t) * (tail - head)); e->path_len = tail - head;
        e->state = bd_find_path(g->h, e->path, e->path_len);
    }
    uint32_t mh = 0; while (mh < e->n_mism && e->mism[mh] < e->r0) ++mh;
    uint32_t mt = mh; while (mt < e->n_mism && e->mism[mt] < e->r1) ++mt;
    memmove(e->mism, e->mism + mh, sizeof(uint32_t) * (mt - mh)); e->n_mism = mt - mh;
    return 1;
}

/* priority queue of (extension, insertion number): highest score first, the later insertion among equals (:567-571) */
typedef struct { QItem* a; uint32_t n, cap; } Heap;
static int q_less(const QItem* x, const QItem* y) { return x->e.score < y->e.score || (x->e.score == y->e.score && x->number < y->number); }
static void heap_push(Heap* hp, QItem it) {
    if (hp->n == hp->cap) { hp->cap = hp->cap * 2 + 16; hp->a = (QItem*)realloc(hp->a, sizeof(QItem) * hp->cap); }
    uint32_t i = hp->n++; hp->a[i] = it;
    while (i) { const uint32_t p = (i - 1) / 2; if (!q_less(&hp->a[p], &hp->a[i])) break; QItem t = hp->a[p]; hp->a[p] = hp->a[i]; hp->a[i] = t; i = p; }
}
static QItem heap_pop(Heap* hp) {
    QItem top = hp->a[0]; hp->a[0] = hp->a[--hp->n];
    uint32_t i = 0;
    for (;;) { uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < hp->n && q_less(&hp->a[m], &hp->a[l])) m = l;
        if (r < hp->n && q_less(&hp->a[m], &hp->a[r])) m = r;
        if (m == i) break;
        QItem t = hp->a[m]; hp->a[m] = hp->a[i]; hp->a[i] = t; i = m; }
    return top;
}

static int state_eq(BState a, BState b) { return a.f.node == b.f.node && a.f.lo == b.f.lo && a.f.hi == b.f.hi && a.b.node == b.b.node && a.b.lo == b.b.lo && a.b.hi == b.b.hi; }
static int ext_eq(const Ext* a, const Ext* b) { return a->r0 == b->r0 && a->r1 == b->r1 && state_eq(a->state, b->state) && a->offset == b->offset; }   /* hpp operator== */
static int dup_less(const Ext* a, const Ext* b) {                          /* remove_duplicates' sort order (:333-349) */
    if (a->r0 != b->r0) return a->r0 < b->r0;
    if (a->r1 != b->r1) return a->r1 < b->r1;
    if (a->state.b.node != b->state.b.node) return a->state.b.node < b->state.b.node;
    if (a->state.f.node != b->state.f.node) return a->state.f.node < b->state.f.node;
    if (a->state.b.lo != b->state.b.lo) return a->state.b.lo < b->state.b.lo;
    if (a->state.b.hi != b->state.b.hi) return a->state.b.hi < b->state.b.hi;
    if (a->state.f.lo != b->state.f.lo) return a->state.f.lo < b->state.f.lo;
    if (a->state.f.hi != b->state.f.hi) return a->state.f.hi < b->state.f.hi;
    return a->offset < b->offset;
}
static int full_less(const Ext* a, const Ext* b) {                         /* handle_full_length's sort order (:302-307) */
    if (ext_full(a) && ext_full(b)) return a->internal < b->internal;
    return ext_full(a) && !ext_full(b);
}
static void stable_sort(Ext* v, uint32_t n, int (*less)(const Ext*, const Ext*)) {
    for (uint32_t i = 1; i < n; ++i) { Ext x = v[i]; uint32_t j = i; while (j && less(&x, &v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = x; }
}
static uint32_t remove_duplicates(Ext* v, uint32_t n) {
    stable_sort(v, n, dup_less);
    uint32_t tail = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (ext_len(&v[i]) == 0) { ext_free(&v[i]); continue; }
        if (tail == 0 || !ext_eq(&v[i], &v[tail - 1])) { if (i > tail) v[tail] = v[i]; ++tail; }
        else ext_free(&v[i]);
    }
    return tail;
}

int vgo_gapless_extend(const vgk_scoring* sc, const vgk_haplo* h, const vgk_gapless_problem* p, vgk_gapless_result* res,
                       vgk_extension* ext_out, uint32_t ext_cap, uint32_t* nodes_out, uint32_t nodes_cap,
                       uint32_t* mism_out, uint32_t mism_cap, uint32_t* n_nodes_out, uint32_t* n_mism_out) {
    memset(res, 0, sizeof *res); *n_nodes_out = 0; *n_mism_out = 0;
    if (!h || !p->read || !p->read_len || !p->n_seeds) return VGK_OK;       /* empty result, like :535-537 */
    G g; g.h = h; g.L = p->read_len; g.match = sc->matrix[0]; g.mismatch = -sc->matrix[1]; g.bonus = sc->full_length_bonus;
    char* seq = (char*)malloc(p->read_len + 1);
    for (uint32_t i = 0; i < p->read_len; ++i) { const char c = p->read[i]; seq[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'X'; }   /* ReadMasker :165-176 */
    g.seq = seq;
    const uint32_t max_mm = p->max_mismatches;
    Ext* result = (Ext*)calloc(p->n_seeds + 1, sizeof(Ext)); uint32_t n_res = 0;
    uint32_t best_alignment = UINT32_MAX;
    Heap hp = { NULL, 0, 0 };
    int rc = VGK_OK;
    for (uint32_t si = 0; si < p->n_seeds; ++si) {
        const int32_t snode = (int32_t)p->seeds[si].node; const int64_t diff = p->seeds[si].diff;
        if (snode < 0 || (uint32_t)snode >= h->n_oriented) { rc = VGK_EINVAL; break; }
        if (best_alignment < n_res && result[best_alignment].internal == 0 && ext_contains(&g, &result[best_alignment], snode, diff)) continue;
        Ext best; memset(&best, 0, sizeof best); best.score = INT32_MIN; best.internal = best.old = UINT32_MAX;
        uint64_t number = 0;
        {
            const uint32_t read_offset = diff < 0 ? 0u : (uint32_t)diff, node_offset = diff < 0 ? (uint32_t)(-diff) : 0u;
            if (read_offset > g.L || node_offset > h->len[snode]) { rc = VGK_EINVAL; break; }
            Ext m; memset(&m, 0, sizeof m);
            m.path = (int32_t*)malloc(sizeof(int32_t)); m.path[0] = snode; m.path_len = m.path_cap = 1;
            m.offset = node_offset; m.state = bd_find_node(h, snode); m.r0 = m.r1 = read_offset;
            match_initial(&g, &m);
            if (m.r0 == 0) m.left_full = m.left_max = 1;
            if (m.r1 >= g.L) m.right_full = m.right_max = 1;
            set_score(&g, &m);
            heap_push(&hp, (QItem){ m, number++ });
        }
        while (hp.n) {
            Ext curr = heap_pop(&hp).e;
            if (!curr.right_max) {
                uint32_t num_ext = 0;
                const uint32_t lim_a = max_mm + 1, lim_b = max_mm / 2 + curr.old + 1, limit = lim_a > lim_b ? lim_a : lim_b;
                const uint32_t o = (uint32_t)curr.state.f.node;
                for (uint32_t e = h->edge_off[o]; e < h->edge_off[o + 1]; ++e) {
                    const int32_t w = h->edge_to[e]; if (w < 0) continue;
                    const BState ns = bd_extend_forward(h, curr.state, w);
                    if (!bd_size(ns)) continue;
                    Ext next = ext_copy_meta(&curr); next.state = ns;
                    const uint32_t node_offset = match_forward(&g, &next, w, limit);
                    if (node_offset == 0) continue;
                    set_path(&next, curr.path, curr.path_len, w, 0);
                    if (next.r1 >= g.L) { next.right_full = next.right_max = 1; next.old = next.internal; }
                    else if (node_offset < h->len[w]) { next.right_max = 1; next.old = next.internal; }
                    set_score(&g, &next);
                    num_ext += bd_size(next.state);
                    heap_push(&hp, (QItem){ next, number++ });
                }
                if (num_ext < bd_size(curr.state)) { curr.right_max = 1; curr.old = curr.internal; heap_push(&hp, (QItem){ curr, number++ }); }
                else ext_free(&curr);
                continue;
            }
            if (!curr.left_max) {
                int found = 0;
                const uint32_t lim_a = max_mm + 1, lim_b = max_mm / 2 + curr.old + 1, limit = lim_a > lim_b ? lim_a : lim_b;
                const uint32_t o = (uint32_t)curr.state.b.node;
                for (uint32_t e = h->edge_off[o]; e < h->edge_off[o + 1]; ++e) {
                    const int32_t x = h->edge_to[e]; if (x < 0) continue;
                    const BState ns = bd_flip(bd_extend_forward(h, bd_flip(curr.state), x));     /* bdExtendBackward */
                    if (!bd_size(ns)) continue;
                    const int32_t w = ns.b.node ^ 1;                                           /* the predecessor, read forward */
                    Ext next = ext_copy_meta(&curr); next.state = ns; next.offset = h->len[w];
                    match_backward(&g, &next, w, limit);
                    if (next.offset >= h->len[w]) continue;
                    set_path(&next, curr.path, curr.path_len, w, 1);
                    if (next.r0 == 0) next.left_full = next.left_max = 1;
                    else if (next.offset > 0) next.left_max = 1;
                    set_score(&g, &next);
                    heap_push(&hp, (QItem){ next, number++ });
                    found = 1;
                }
                if (!found) curr.left_max = 1;
                else { ext_free(&curr); continue; }
            }
            if (best.score < curr.score) { ext_free(&best); best = curr; } else ext_free(&curr);
        }
        if (ext_len(&best) > 0) {
            if (ext_full(&best) && (best_alignment >= n_res || best.internal < result[best_alignment].internal)) best_alignment = n_res;
            result[n_res++] = best;
        } else ext_free(&best);
    }
    free(hp.a);
    if (rc == VGK_OK) {
        if (best_alignment < n_res && result[best_alignment].internal <= max_mm) {
            /* handle_full_length (:301-329) */
            stable_sort(result, n_res, full_less);
            uint32_t tail = 0;
            for (uint32_t i = 0; i < n_res; ++i) {
                if (!ext_full(&result[i])) { for (uint32_t k = i; k < n_res; ++k) ext_free(&result[k]); break; }
                int ov = 0;
                for (uint32_t prev = 0; prev < tail; ++prev)
                    if ((double)ext_overlap(&g, &result[i], &result[prev]) > p->overlap_threshold * (double)ext_len(&result[prev])) { ov = 1; break; }
                if (ov) { ext_free(&result[i]); continue; }
                if (i > tail) result[tail] = result[i];
                ++tail;
            }
            n_res = tail;
            for (uint32_t i = 0; i < n_res; ++i) find_mismatches(&g, &result[i]);
            res->full_length = 1;
        } else {
            n_res = remove_duplicates(result, n_res);
            for (uint32_t i = 0; i < n_res; ++i) find_mismatches(&g, &result[i]);
            if (p->flags & VGK_GAPLESS_TRIM) {
                int trimmed = 0;
                for (uint32_t i = 0; i < n_res; ++i) trimmed |= trim_mismatches(&g, &result[i]);
                if (trimmed) n_res = remove_duplicates(result, n_res);
            }
        }
        uint32_t nn = 0, nm = 0;
        for (uint32_t i = 0; i < n_res && rc == VGK_OK; ++i) {
            const Ext* e = &result[i];
            if (i >= ext_cap || nn + e->path_len > nodes_cap || nm + e->n_mism > mism_cap) { rc = VGK_EOPS; break; }
            vgk_extension* x = &ext_out[i]; memset(x, 0, sizeof *x);
            x->path_begin = nn; x->path_len = e->path_len; x->offset = e->offset; x->read_begin = e->r0; x->read_end = e->r1;
            x->mism_begin = nm; x->n_mismatches = e->n_mism; x->score = e->score; x->left_full = (uint8_t)e->left_full; x->right_full = (uint8_t)e->right_full;
            x->state[0] = (uint32_t)e->state.f.node; x->state[1] = (uint32_t)e->state.f.lo; x->state[2] = (uint32_t)e->state.f.hi;
            x->state[3] = (uint32_t)e->state.b.node; x->state[4] = (uint32_t)e->state.b.lo; x->state[5] = (uint32_t)e->state.b.hi;
            for (uint32_t k = 0; k < e->path_len; ++k) nodes_out[nn++] = (uint32_t)e->path[k];
            for (uint32_t k = 0; k < e->n_mism; ++k) mism_out[nm++] = e->mism[k];
        }
        if (rc == VGK_OK) { res->n_ext = n_res; *n_nodes_out = nn; *n_mism_out = nm; }
    }
    for (uint32_t i = 0; i < n_res; ++i) ext_free(&result[i]);
    free(result); free(seq);
    res->status = rc;
    return rc;
}
