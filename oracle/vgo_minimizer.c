/*
 * vgo_minimizer.c — CPU ORACLE for minimizer seeding (first half).
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * What MinimizerMapper::find_minimizers / find_seeds (src/minimizer_mapper.cpp:3918-3965, :4109-4290) get from gbwtgraph's
 * MinimizerIndex — minimizer_regions(sequence) and find(minimizer) — restated from the published scheme [prior knowledge; gbwtgraph is
 * an un-vendored submodule, absent from the snapshot]: k-mers as 2-bit keys, Thomas Wang's 64-bit hash, the orientation with the
 * smaller hash canonical, the leftmost smallest candidate of every window of w k-mers, each position once; the index files the
 * minimizers of every haplotype thread under the position their canonical orientation starts at.  Written the plain way — every
 * window scanned in full, the index a sorted array searched by bisection — where the engine keeps a ring and a hash table.
 *
 * Parity status: PARITY-UNPINNED against the reference (no vectors for this path in the snapshot).  tests/test_minimizer.py pins this
 * file and the engine on a third, brute-force construction in Python and on the property that matters downstream: every seed of a
 * read sampled from a haplotype lies on the read's true diagonal.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "../include/vgk.h"

typedef struct { uint64_t key, hash; uint32_t node, offset; } Entry;
struct vgk_minimizer_index { uint32_t k, w, n_nodes; uint32_t* node_len; Entry* e; size_t n; uint64_t n_keys; int policy_on; vgk_seed_policy policy; };

static uint64_t wang(uint64_t key) {
    key = (~key) + (key << 21); key ^= key >> 24; key = (key + (key << 3)) + (key << 8); key ^= key >> 14;
    key = (key + (key << 2)) + (key << 4); key ^= key >> 28; key += key << 31;
    return key;
}
static int code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

/* canonical k-mer at seq[p, p + k): 0 when it holds a character that is not ACGT */
static int kmer_at(const char* seq, uint32_t p, uint32_t k, uint64_t* key, uint64_t* hash, int* reverse) {
    uint64_t f = 0, r = 0;
    for (uint32_t i = 0; i < k; ++i) {
        const int x = code(seq[p + i]); if (x < 0) return 0;
        f = (f << 2) | (uint64_t)x;
        r |= (uint64_t)(3 - x) << (2 * i);
    }
    const uint64_t hf = wang(f), hr = wang(r);
    *reverse = hr < hf; *hash = *reverse ? hr : hf; *key = *reverse ? r : f;
    return 1;
}
/* minimizers of seq[0, L): for every window of w k-mers the leftmost smallest candidate; positions ascending, each once */
typedef void (*emit_fn)(void* ctx, uint32_t p, uint64_t key, uint64_t hash, int reverse);
static void minimizers(const char* seq, uint32_t L, uint32_t k, uint32_t w, emit_fn emit, void* ctx) {
    if (L < k + w - 1) return;
    const uint32_t n = L - k + 1;
    uint64_t* key = (uint64_t*)malloc(sizeof(uint64_t) * n); uint64_t* hash = (uint64_t*)malloc(sizeof(uint64_t) * n);
    int* rev = (int*)malloc(sizeof(int) * n); char* ok = (char*)malloc(n);
    for (uint32_t p = 0; p < n; ++p) ok[p] = (char)kmer_at(seq, p, k, &key[p], &hash[p], &rev[p]);
    int64_t last = -1;
    for (uint32_t s = 0; s + w <= n; ++s) {
        int64_t best = -1;
        for (uint32_t p = s; p < s + w; ++p) if (ok[p] && (best < 0 || hash[p] < hash[best])) best = p;
        if (best >= 0 && best != last) { emit(ctx, (uint32_t)best, key[best], hash[best], rev[best]); last = best; }
    }
    free(key); free(hash); free(rev); free(ok);
}

typedef struct { Entry* e; size_t n, cap; const uint32_t* tn; const uint64_t* start; const uint32_t* step; const uint32_t* node_len; uint32_t k; int oom; } Build;
static void build_emit(void* c, uint32_t p, uint64_t key, uint64_t hash, int reverse) {
    Build* b = (Build*)c;
    if (b->n == b->cap) { const size_t cap = b->cap ? 2 * b->cap : 1024; Entry* q = (Entry*)realloc(b->e, sizeof(Entry) * cap); if (!q) { b->oom = 1; return; } b->e = q; b->cap = cap; }
    Entry* e = &b->e[b->n++]; e->key = key; e->hash = hash;
    if (!reverse) { const uint32_t x = b->step[p]; e->node = b->tn[x]; e->offset = (uint32_t)(p - b->start[x]); }
    else { const uint32_t q = p + b->k - 1, x = b->step[q]; e->node = b->tn[x] ^ 1u; e->offset = b->node_len[b->tn[x] >> 1] - 1 - (uint32_t)(q - b->start[x]); }
}
static int cmp_entry(const void* a, const void* b) {
    const Entry* x = (const Entry*)a; const Entry* y = (const Entry*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->node != y->node) return x->node < y->node ? -1 : 1;
    return x->offset < y->offset ? -1 : x->offset > y->offset;
}

void vgk_minimizer_index_destroy(vgk_minimizer_index* ix) { if (ix) { free(ix->node_len); free(ix->e); free(ix); } }
int vgk_minimizer_index_create(vgk_ctx* ctx, const vgk_haplotypes* d, uint32_t k, uint32_t w, vgk_minimizer_index** out) {
    if (!ctx || !d || !out || !d->n_nodes || !d->node_len || !d->seq || (d->n_threads && (!d->thread_off || !d->thread_nodes))) return VGK_EINVAL;
    if (k == 0 || k > 31 || w == 0 || w > 64) return VGK_EINVAL;
    *out = NULL;
    uint64_t* node_at = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)d->n_nodes + 1));
    node_at[0] = 0; for (uint32_t i = 0; i < d->n_nodes; ++i) node_at[i + 1] = node_at[i] + d->node_len[i];
    Build b; memset(&b, 0, sizeof b); b.node_len = d->node_len; b.k = k;
    for (uint32_t t = 0; t < d->n_threads; ++t) {
        const uint32_t* tn = d->thread_nodes + d->thread_off[t]; const uint32_t len = d->thread_off[t + 1] - d->thread_off[t];
        size_t bases = 0;
        for (uint32_t x = 0; x < len; ++x) { if (tn[x] >= 2 * d->n_nodes) { free(node_at); free(b.e); return VGK_EINVAL; } bases += d->node_len[tn[x] >> 1]; }
        char* seq = (char*)malloc(bases + 1); uint32_t* step = (uint32_t*)malloc(sizeof(uint32_t) * (bases + 1)); uint64_t* start = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)len + 1));
        size_t at = 0;
        for (uint32_t x = 0; x < len; ++x) {
            const uint32_t o = tn[x], v = o >> 1, L = d->node_len[v]; const char* s = d->seq + node_at[v];
            start[x] = at;
            for (uint32_t i = 0; i < L; ++i) { seq[at] = (o & 1) ? comp(s[L - 1 - i]) : s[i]; step[at] = x; ++at; }
        }
        b.tn = tn; b.start = start; b.step = step;
        minimizers(seq, (uint32_t)bases, k, w, build_emit, &b);
        free(seq); free(step); free(start);
        if (b.oom) { free(node_at); free(b.e); return VGK_ENOMEM; }
    }
    free(node_at);
    qsort(b.e, b.n, sizeof(Entry), cmp_entry);
    size_t m = 0;
    for (size_t i = 0; i < b.n; ++i) if (i == 0 || cmp_entry(&b.e[i], &b.e[m - 1]) != 0) b.e[m++] = b.e[i];
    vgk_minimizer_index* ix = (vgk_minimizer_index*)calloc(1, sizeof *ix);
    ix->k = k; ix->w = w; ix->n_nodes = d->n_nodes; ix->e = b.e; ix->n = m;
    ix->node_len = (uint32_t*)malloc(sizeof(uint32_t) * d->n_nodes); memcpy(ix->node_len, d->node_len, sizeof(uint32_t) * d->n_nodes);
    for (size_t i = 0; i < m; ++i) if (i == 0 || ix->e[i].key != ix->e[i - 1].key) ++ix->n_keys;
    *out = ix;
    return VGK_OK;
}
uint64_t vgk_minimizer_index_keys(const vgk_minimizer_index* ix) { return ix ? ix->n_keys : 0; }
uint64_t vgk_minimizer_index_hits(const vgk_minimizer_index* ix) { return ix ? ix->n : 0; }
int vgk_minimizer_index_fetch(const vgk_minimizer_index* ix, vgk_minimizer_hit* hits, size_t cap) {
    if (!ix || (!hits && cap)) return VGK_EINVAL;
    if (cap < ix->n) return VGK_EOPS;
    for (size_t i = 0; i < ix->n; ++i) { hits[i].key = ix->e[i].key; hits[i].node = ix->e[i].node; hits[i].offset = ix->e[i].offset; }
    return VGK_OK;
}

static void key_range(const vgk_minimizer_index* ix, uint64_t key, size_t* first, size_t* end) {      /* the entries with this key */
    size_t lo = 0, hi = ix->n;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (ix->e[mid].key < key) lo = mid + 1; else hi = mid; }
    size_t e = lo; while (e < ix->n && ix->e[e].key == key) ++e;
    *first = lo; *end = e;
}
/* ---- find_seeds' choice of minimizers (include/vgk.h: vgk_seed_policy), restating src/minimizer_mapper.cpp:3927-3937 (scores), :4074-4107 (the
 * order: runs of one key together, by descending score, equal scores by key — and the runs tied with the BEST score shuffled as
 * sort_shuffling_ties does it, src/utility.hpp:720-727, :771-799, with the generator LazyRNG seeds from the read, src/utility.cpp:911-927),
 * :4395-4440 (the run logic) and the filters any-hits :4270, hard-hit-cap :4277 and hit-cap || score-fraction :4358-4378.  At most 64 minimizers. */
typedef struct { const vgk_minimizer_index* ix; uint64_t key[64]; uint32_t hits[64]; uint32_t n; } Listing;
static void list_emit(void* c, uint32_t p, uint64_t key, uint64_t hash, int reverse) {
    Listing* l = (Listing*)c; (void)p; (void)hash; (void)reverse;
    if (l->n < 64) { size_t a, b; key_range(l->ix, key, &a, &b); l->key[l->n] = key; l->hits[l->n] = (uint32_t)(b - a); }
    ++l->n;
}
/* the generator: std::minstd_rand (Lehmer, multiplier 48271 modulo 2^31 - 1; a seed that is 0 modulo that starts it at 1) */
static uint32_t lehmer_start(uint32_t seed) { const uint32_t x = seed % 2147483647u; return x ? x : 1u; }
static uint32_t lehmer_next(uint32_t* x) { *x = (uint32_t)(((uint64_t)*x * 48271ull) % 2147483647ull); return *x; }
/* seq / L: the read as the caller gave it.  *unsure: the read holds a base other than A, C, G, T and the order inside its top tie can change the choice —
 * the engine, which keeps reads masked, does not choose for such a read (VGK_MINIMIZERS_POLICY_SKIPPED), and neither does this checker */
static uint64_t choose(const vgk_seed_policy* P, const uint64_t* key, const uint32_t* hits, uint32_t n, const char* seq, uint32_t L, int* unsure) {
    double score[64]; uint32_t order[64];
    *unsure = 0;
    const double base_score = 1.0 + log((double)P->hard_hit_cap);
    for (uint32_t i = 0; i < n; ++i) score[i] = !hits[i] ? 0.0 : (hits[i] <= P->hard_hit_cap ? base_score - log((double)hits[i]) : 1.0);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    for (uint32_t i = 1; i < n; ++i) {                                    /* insertion sort: score descending, key ascending, read position */
        const uint32_t x = order[i]; uint32_t j = i;
        while (j && (score[order[j - 1]] < score[x] || (score[order[j - 1]] == score[x] && key[order[j - 1]] > key[x]))) { order[j] = order[j - 1]; --j; }
        order[j] = x;
    }
    const int use_score = P->hit_cap != 0 || P->minimizer_score_fraction != 1.0;
    {   /* the ties at the top: the run boundaries of the leading stretch of equal score, a Knuth shuffle of those runs, the runs laid out again */
        uint32_t first[65], n_runs = 0, r = 0;
        while (r < n && score[order[r]] == score[order[0]]) { first[n_runs++] = r; const uint64_t k = key[order[r]]; while (r < n && key[order[r]] == k) ++r; }
        first[n_runs] = r;
        if (n_runs >= 2) {
            uint32_t seed = 0; int other = 0;
            for (uint32_t i = 0; i < L; ++i) { const char c = seq[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') other = 1; seed = seed * 13u + (uint32_t)(unsigned char)c; }
            if ((other || P->paired) && use_score && hits[order[0]] > P->hit_cap) { *unsure = 1; return 0; }      /* (paired: the pair's one generator is the caller's, include/vgk.h) */
            uint32_t which[64], x = lehmer_start(seed), laid[64], at = 0;
            for (uint32_t i = 0; i < n_runs; ++i) which[i] = i;
            for (uint32_t i = 1; i < n_runs; ++i) { const uint32_t j = lehmer_next(&x) % (i + 1u), t2 = which[j]; which[j] = which[i]; which[i] = t2; }
            for (uint32_t i = 0; i < n_runs; ++i) for (uint32_t e = first[which[i]]; e < first[which[i] + 1]; ++e) laid[at++] = order[e];
            for (uint32_t e = 0; e < at; ++e) order[e] = laid[e];
        }
    }
    volatile double base_target = 0.0, target = 0.0, selected = 0.0, t;
    if (use_score) { for (uint32_t r = 0; r < n; ++r) base_target = base_target + score[order[r]]; t = base_target * P->minimizer_score_fraction; target = t + 0.000001; }
    uint64_t mask = 0; uint32_t at = 0;
    while (at < n) {
        uint32_t end = at + 1; uint64_t run_hits = hits[order[at]];
        while (end < n && key[order[end]] == key[order[at]]) { run_hits += hits[order[end]]; ++end; }
        int taking_run = 0;
        for (uint32_t r = at; r < end; ++r) {
            const uint32_t i = order[r];
            int pass = hits[i] > 0 && run_hits <= P->hard_hit_cap;
            if (pass && use_score) {
                t = selected + score[i];
                if (hits[i] <= P->hit_cap || t <= target || taking_run) selected = t;
                else { pass = 0; target = selected; }
            }
            if (pass) { mask |= 1ull << i; taking_run = 1; }
        }
        at = end;
    }
    return mask;
}
int vgk_minimizer_set_policy(vgk_minimizer_index* ix, const vgk_seed_policy* policy) {
    if (!ix) return VGK_EINVAL;
    if (!policy) { ix->policy_on = 0; return VGK_OK; }
    if (!policy->hard_hit_cap || policy->hard_hit_cap > 65535u || !(policy->minimizer_score_fraction >= 0.0 && policy->minimizer_score_fraction <= 1.0)) return VGK_EINVAL;
    ix->policy = *policy; ix->policy_on = 1;
    return VGK_OK;
}
/* checker-side only: the minimizers of one read as the choice sees them -> their number; key / read offset of the k-mer's first base / hits of
 * the first `cap` of them (tests/test_seed_policy.py holds the device's choice to the host shim's select_minimizers through this) */
typedef struct { const vgk_minimizer_index* ix; uint64_t* key; uint32_t* offset; uint32_t* hits; uint32_t n, cap; } Dump;
static void dump_emit(void* c, uint32_t p, uint64_t key, uint64_t hash, int reverse) {
    Dump* d = (Dump*)c; (void)hash; (void)reverse;
    if (d->n < d->cap) { size_t a, b; key_range(d->ix, key, &a, &b); d->key[d->n] = key; d->offset[d->n] = p; d->hits[d->n] = (uint32_t)(b - a); }
    ++d->n;
}
uint32_t vgo_minimizer_list(const vgk_minimizer_index* ix, const char* read, uint32_t len, uint64_t* key, uint32_t* offset, uint32_t* hits, uint32_t cap) {
    Dump d; d.ix = ix; d.key = key; d.offset = offset; d.hits = hits; d.n = 0; d.cap = cap;
    minimizers(read, len, ix->k, ix->w, dump_emit, &d);
    return d.n;
}

typedef struct { const vgk_minimizer_index* ix; uint32_t hit_cap; vgk_seed seeds[64]; uint32_t n_seeds, n_min; int truncated; uint64_t chosen; } Query;
static void query_emit(void* c, uint32_t p, uint64_t key, uint64_t hash, int reverse) {
    Query* q = (Query*)c; const vgk_minimizer_index* ix = q->ix; (void)hash;
    const uint32_t ordinal = q->n_min++;
    size_t lo, end; key_range(ix, key, &lo, &end);
    if (end == lo || end - lo > q->hit_cap) return;
    if (ordinal < 64 && !((q->chosen >> ordinal) & 1ull)) return;
    for (size_t h = lo; h < end; ++h) {
        if (q->n_seeds >= 64) { q->truncated = 1; break; }                /* the cap: this hit and the rest are never looked at */
        vgk_seed s;
        if (!reverse) { s.node = ix->e[h].node; s.diff = (int32_t)p - (int32_t)ix->e[h].offset; }
        else { s.node = ix->e[h].node ^ 1u; s.diff = (int32_t)(p + ix->k - 1) - (int32_t)(ix->node_len[ix->e[h].node >> 1] - 1 - ix->e[h].offset); }
        int dup = 0;
        for (uint32_t j = 0; j < q->n_seeds && !dup; ++j) dup = q->seeds[j].node == s.node && q->seeds[j].diff == s.diff;
        if (!dup) q->seeds[q->n_seeds++] = s;
    }
}
int vgk_minimizer_seeds(vgk_ctx* ctx, const vgk_minimizer_index* ix, const vgk_haplo* graph, const char* reads, const uint64_t* read_off, uint32_t n,
                        uint32_t hit_cap, uint32_t* seed_off, uint32_t* mins, vgk_seed* seeds, size_t seeds_cap, size_t* written) {
    if (!ctx || !ix || !graph || (n && (!reads || !read_off || !seed_off))) return VGK_EINVAL;
    if (written) *written = 0;
    size_t total = 0; int rc = VGK_OK;
    if (seed_off) seed_off[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Query q; q.ix = ix; q.hit_cap = hit_cap ? hit_cap : 0xffffffffu; q.n_seeds = 0; q.n_min = 0; q.truncated = 0; q.chosen = ~0ull;
        int skipped = 0;
        if (ix->policy_on) {                                              /* the choice first, over the whole read; then the seeds of the chosen */
            Listing l; l.ix = ix; l.n = 0;
            minimizers(reads + read_off[i], (uint32_t)(read_off[i + 1] - read_off[i]), ix->k, ix->w, list_emit, &l);
            if (l.n > 64) { skipped = 1; q.hit_cap = ix->policy.hard_hit_cap; }
            else {
                int unsure = 0;
                q.chosen = choose(&ix->policy, l.key, l.hits, l.n, reads + read_off[i], (uint32_t)(read_off[i + 1] - read_off[i]), &unsure); q.hit_cap = 0xffffffffu;
                if (unsure) { skipped = 1; q.chosen = ~0ull; q.hit_cap = ix->policy.hard_hit_cap; }
            }
        }
        minimizers(reads + read_off[i], (uint32_t)(read_off[i + 1] - read_off[i]), ix->k, ix->w, query_emit, &q);
        if (mins) mins[i] = q.n_min | (q.truncated ? VGK_MINIMIZERS_TRUNCATED : 0u) | (skipped ? VGK_MINIMIZERS_POLICY_SKIPPED : 0u);
        if (total + q.n_seeds <= seeds_cap && seeds) memcpy(seeds + total, q.seeds, sizeof(vgk_seed) * q.n_seeds); else if (q.n_seeds) rc = VGK_EOPS;
        total += q.n_seeds; seed_off[i + 1] = (uint32_t)total;
    }
    if (written) *written = total;
    return rc;
}
double vgk_minimizer_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }

/* ---- reads of any length (include/vgk.h: vgk_minimizer_list / vgk_minimizer_seeds_of): every minimizer, then one seed per hit of the taken ones ---- */
typedef struct { const vgk_minimizer_index* ix; vgk_read_minimizer* out; size_t n, cap; } LongList;
static void long_emit(void* c, uint32_t p, uint64_t key, uint64_t hash, int reverse) {
    LongList* l = (LongList*)c; (void)hash;
    if (l->out && l->n < l->cap) { size_t a, b; key_range(l->ix, key, &a, &b); vgk_read_minimizer r; r.key = key; r.offset = p; r.hits = (uint32_t)(b - a); r.flags = reverse ? VGK_MINIMIZER_REVERSE : 0u; r.reserved = 0; l->out[l->n] = r; }
    ++l->n;
}
int vgk_minimizer_list(vgk_ctx* ctx, const vgk_minimizer_index* ix, const char* reads, const uint64_t* read_off, uint32_t n,
                       uint64_t* minimizer_off, vgk_read_minimizer* out, size_t cap, size_t* written) {
    if (!ctx || !ix || !minimizer_off || (n && (!reads || !read_off)) || (!out && cap)) return VGK_EINVAL;
    LongList l; l.ix = ix; l.out = out; l.n = 0; l.cap = cap;
    minimizer_off[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (read_off[i + 1] < read_off[i]) return VGK_EINVAL;
        minimizers(reads + read_off[i], (uint32_t)(read_off[i + 1] - read_off[i]), ix->k, ix->w, long_emit, &l);
        minimizer_off[i + 1] = l.n;
    }
    if (written) *written = l.n;
    return l.n > cap ? VGK_EOPS : VGK_OK;
}
int vgk_minimizer_seeds_of(vgk_ctx* ctx, const vgk_minimizer_index* ix, const vgk_read_minimizer* minimizers, const uint8_t* take, size_t n_minimizers,
                           uint64_t* seed_off, vgk_seed* seeds, size_t cap, size_t* written) {
    if (!ctx || !ix || !seed_off || (n_minimizers && (!minimizers || !take)) || (!seeds && cap)) return VGK_EINVAL;
    size_t total = 0;
    seed_off[0] = 0;
    for (size_t j = 0; j < n_minimizers; ++j) {
        if (take[j]) {
            size_t lo, end; key_range(ix, minimizers[j].key, &lo, &end);
            const uint32_t p = minimizers[j].offset; const int reverse = (minimizers[j].flags & VGK_MINIMIZER_REVERSE) != 0;
            for (size_t h = lo; h < end; ++h, ++total) {
                if (total >= cap) continue;
                vgk_seed s;
                if (!reverse) { s.node = ix->e[h].node; s.diff = (int32_t)p - (int32_t)ix->e[h].offset; }
                else { s.node = ix->e[h].node ^ 1u; s.diff = (int32_t)(p + ix->k - 1) - (int32_t)(ix->node_len[ix->e[h].node >> 1] - 1 - ix->e[h].offset); }
                seeds[total] = s;
            }
        }
        seed_off[j + 1] = total;
    }
    if (written) *written = total;
    return total > cap ? VGK_EOPS : VGK_OK;
}
