/*
 * vgo_wfa.c — CPU ORACLE for haplotype-consistent wavefront alignment.
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * Restates WFAExtender::connect / suffix / prefix and the WFATree they run on, from the reference's
 * src/gbwt_extender.cpp: MatchPos :1276-1365, WFAPoint :1368-1420, WFANode :1434-1557, WFATree :1567-2046
 * (extend :1656, next_score :1672, next :1709, predecessors :1791-1823, trim :1849, extend_over :1874, get_diagonals
 * :1977, expand_if_necessary :1992, find_pos :2015), connect :2052-2235, suffix :2237, prefix :2248,
 * WFAAlignment::final_offset / flip / append :821-859, ErrorModel::Event::evaluate gbwt_extender.hpp:371.
 * The haplotype index is the one of vgo_gapless.c (vgo_haplo.h); only the forward search state is used here.
 *
 * Differences in representation, none in result: the wavefronts of a tree node are open-addressing tables keyed by
 * (score, diagonal) instead of hash_map; a MatchPos keeps (current tree node, the node the lookup started from) instead
 * of a stack of tree offsets — the stack is the tree path between the two; the children of a tree node are consecutive.
 *
 * Parity status: pinned on the reference's known-answer tests for this path (src/unittest/gbwt_extender.cpp:1531-2640,
 * transcribed by tests/golden/extract_wfa_tests.py into tests/golden/ref_wfa_extender.json: scores, success/failure and
 * the validity rules of check_alignment).  PARITY-UNPINNED: WFATree::trim (:1849-1868) keeps the first of several equally
 * good partial alignments in hash_map iteration order; here ties go to the smallest (tree node, penalty, diagonal).
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"
#include "vgo_haplo.h"

enum { W_MATCH = 0, W_INS = 1, W_DEL = 2 };                       /* WFANode::MATCHES / INSERTIONS / DELETIONS */
enum { E_MATCH = 0, E_MISMATCH = 1, E_INS = 2, E_DEL = 3 };       /* WFAAlignment::Edit */
#define TARGET_LENGTH 1024u                                       /* WFANode::TARGET_LENGTH */
#define NO_OFFSET 0xffffffffu

typedef struct { int32_t score, diag; uint32_t seq, off; int used; } Slot;
typedef struct { Slot* s; uint32_t cap, n; } PMap;

static uint32_t pm_hash(int32_t score, int32_t diag) { return ((uint32_t)score * 2654435761u) ^ ((uint32_t)diag * 40503u + 0x9e37u); }
static Slot* pm_find(const PMap* m, int32_t score, int32_t diag) {
    if (!m->cap) return NULL;
    for (uint32_t i = pm_hash(score, diag) & (m->cap - 1);; i = (i + 1) & (m->cap - 1)) {
        Slot* s = &m->s[i];
        if (!s->used) return NULL;
        if (s->score == score && s->diag == diag) return s;
    }
}
static void pm_put(PMap* m, int32_t score, int32_t diag, uint32_t seq, uint32_t off) {          /* WFANode::update :1517-1530 */
    Slot* f = pm_find(m, score, diag);
    if (f) { f->seq = seq; f->off = off; return; }
    if (2 * (m->n + 1) > m->cap) {
        PMap g = { NULL, m->cap ? 2 * m->cap : 16, 0 };
        g.s = (Slot*)calloc(g.cap, sizeof(Slot));
        for (uint32_t i = 0; i < m->cap; ++i) if (m->s[i].used) pm_put(&g, m->s[i].score, m->s[i].diag, m->s[i].seq, m->s[i].off);
        free(m->s); *m = g;
    }
    uint32_t i = pm_hash(score, diag) & (m->cap - 1);
    while (m->s[i].used) i = (i + 1) & (m->cap - 1);
    m->s[i] = (Slot){ score, diag, seq, off, 1 }; ++m->n;
}

typedef struct {
    int32_t* path; uint32_t path_len, path_cap;                   /* oriented nodes */
    SState   state;                                               /* search state at the end of the path */
    char*    seq; uint32_t len, seq_cap;                          /* concatenated node sequences */
    uint32_t parent, first_child, n_children;
    uint32_t target_offset;
    int      dead_end;
    PMap     wf[3];
} TNode;

typedef struct { int32_t score, min_d, max_d; int gap; } PScore;  /* possible_scores entry :1596-1606 */
typedef struct { int32_t score, diag; uint32_t seq, off; } Point; /* WFAPoint */

typedef struct {
    const vgk_haplo* h; const char* seq; uint32_t L;
    int32_t to_node; uint32_t to_off; int no_to;
    TNode* nodes; uint32_t n_nodes, cap_nodes;
    Point cand; uint32_t cand_node;
    int32_t match, mismatch, gap_open, gap_extend, score_bound, max_distance, min_distance;
    PScore* ps; uint32_t n_ps, cap_ps;                            /* ascending by score */
} Tree;

/* ---- forward search states: the non-empty one-node extensions, in the order of the node's edges (follow_paths) ---- */
static uint32_t follow(const vgk_haplo* h, SState s, SState* out, uint32_t max_out) {
    if (s.lo > s.hi) return 0;
    const uint32_t o = (uint32_t)s.node;
    const uint32_t* body = h->body + h->body_off[o]; const int32_t* et = h->edge_to + h->edge_off[o];
    const uint32_t ne = h->edge_off[o + 1] - h->edge_off[o];
    uint32_t k = 0;
    for (uint32_t e = 0; e < ne; ++e) {
        if (et[e] < 0) continue;
        int32_t before = 0, inside = 0;
        for (int32_t i = 0; i <= s.hi; ++i) if (body[i] == e) { if (i < s.lo) ++before; else ++inside; }
        if (!inside) continue;
        if (k < max_out) { out[k].node = et[e]; out[k].lo = (int32_t)h->edge_base[h->edge_off[o] + e] + before; out[k].hi = out[k].lo + inside - 1; }
        ++k;
    }
    return k;
}

/* ---- tree nodes (WFANode) ---- */
static int append_node(const Tree* t, TNode* n, SState next) {                                   /* :1546-1556 */
    n->state = next;
    if (n->path_len == n->path_cap) { n->path_cap = n->path_cap ? 2 * n->path_cap : 8; n->path = (int32_t*)realloc(n->path, sizeof(int32_t) * n->path_cap); }
    n->path[n->path_len++] = next.node;
    const uint32_t nl = t->h->len[next.node];
    if (n->len + nl + 1 > n->seq_cap) { n->seq_cap = 2 * (n->len + nl) + 16; n->seq = (char*)realloc(n->seq, n->seq_cap); }
    memcpy(n->seq + n->len, t->h->seq + t->h->seq_off[next.node], nl); n->len += nl;
    if (!t->no_to && t->to_node == next.node) { n->target_offset = n->len - (nl - t->to_off); return 1; }
    return 0;
}
static void tnode_init(const Tree* t, TNode* n, SState state, uint32_t parent) {                 /* :1463-1488 */
    memset(n, 0, sizeof *n);
    n->parent = parent; n->target_offset = NO_OFFSET;
    if (append_node(t, n, state)) return;
    while (n->len < TARGET_LENGTH) {
        SState next[2];
        const uint32_t successors = follow(t->h, n->state, next, 1);
        if (successors == 0) { n->dead_end = 1; break; }
        if (successors > 1) break;
        if (append_node(t, n, next[0])) break;
    }
}
static int is_leaf(const TNode* n) { return !n->n_children || n->dead_end; }
static int expanded(const TNode* n) { return n->n_children || n->dead_end; }

/* ---- positions (MatchPos) ---- */
typedef struct { uint32_t seq, off, cur, origin; int empty; } MPos;
static MPos mp_none(void) { MPos p = { 0, 0, 0, 0, 1 }; return p; }
static int mp_less(MPos a, MPos b) { if (a.empty) return !b.empty; if (b.empty) return 0; return a.seq < b.seq; }     /* :1356-1364 */
static int32_t mp_distance(MPos p, int32_t diagonal) { return 2 * (int32_t)p.seq - diagonal; }
static int mp_at_last(MPos p) { return p.cur == p.origin; }
static void mp_pop(const Tree* t, MPos* p) {                      /* one step down the tree path towards the origin */
    uint32_t x = p->origin;
    while (t->nodes[x].parent != p->cur) x = t->nodes[x].parent;
    p->cur = x;
}

static int at_dead_end(const Tree* t, MPos p) { return t->nodes[p.cur].dead_end && p.off >= t->nodes[p.cur].len; }     /* :2043 */

static MPos find_pos(const Tree* t, int type, uint32_t node, int32_t score, int32_t diag, int ext_seq, int ext_graph) { /* :2015-2040 */
    if (score < 0) return mp_none();
    const uint32_t origin = node;
    for (;;) {
        const Slot* s = pm_find(&t->nodes[node].wf[type], score, diag);
        if (s) {
            MPos p = { s->seq, s->off, node, origin, 0 };
            if (ext_seq && p.seq >= t->L) return mp_none();
            if (ext_graph && at_dead_end(t, p)) return mp_none();
            return p;
        }
        if (node == 0) return mp_none();
        node = t->nodes[node].parent;
    }
}
static void update(Tree* t, int type, int32_t score, int32_t diag, MPos p) { pm_put(&t->nodes[p.cur].wf[type], score, diag, p.seq, p.off); }

static MPos ins_predecessor(const Tree* t, uint32_t node, int32_t score, int32_t diag, int* edit) {                  /* :1791-1795 */
    MPos open = find_pos(t, W_MATCH, node, score - t->gap_open - t->gap_extend, diag - 1, 1, 0);
    MPos ext = find_pos(t, W_INS, node, score - t->gap_extend, diag - 1, 1, 0);
    if (mp_less(open, ext)) { *edit = E_INS; return ext; }
    *edit = E_MATCH; return open;
}
static MPos del_predecessor(const Tree* t, uint32_t node, int32_t score, int32_t diag, int* edit) {                  /* :1800-1804 */
    MPos open = find_pos(t, W_MATCH, node, score - t->gap_open - t->gap_extend, diag + 1, 0, 1);
    MPos ext = find_pos(t, W_DEL, node, score - t->gap_extend, diag + 1, 0, 1);
    if (mp_less(open, ext)) { *edit = E_DEL; return ext; }
    *edit = E_MATCH; return open;
}
static MPos match_predecessor(const Tree* t, uint32_t node, int32_t score, int32_t diag, int* edit) {                /* :1809-1823 */
    MPos ins = find_pos(t, W_INS, node, score, diag, 0, 0);
    MPos del = find_pos(t, W_DEL, node, score, diag, 0, 0);
    MPos subst = find_pos(t, W_MATCH, node, score - t->mismatch, diag, 0, 0);
    if (!subst.empty) { subst.seq++; subst.off++; }
    if (mp_less(ins, del)) {
        if (mp_less(del, subst)) { *edit = E_MISMATCH; return subst; }
        *edit = E_DEL; return del;
    }
    if (mp_less(ins, subst)) { *edit = E_MISMATCH; return subst; }
    *edit = E_INS; return ins;
}
static void successor_offset(const Tree* t, MPos* p) {                                            /* :1827-1832 */
    if (p->off >= t->nodes[p->cur].len) { mp_pop(t, p); p->off = 0; }
    p->off++;
}
static void predecessor_offset(const Tree* t, uint32_t* node, uint32_t* off) {                    /* :1835-1842 */
    if (*off > 0) --*off;
    else { *node = t->nodes[*node].parent; *off = t->nodes[*node].len - 1; }
}

static void expand_if_necessary(Tree* t, MPos p) {                                                /* :1992-2008 */
    const uint32_t node = p.cur;
    if (expanded(&t->nodes[node]) || p.off < t->nodes[node].len) return;
    SState next[256];
    const uint32_t k = follow(t->h, t->nodes[node].state, next, 256);
    if (!k) { t->nodes[node].dead_end = 1; return; }
    if (t->n_nodes + k > t->cap_nodes) { while (t->n_nodes + k > t->cap_nodes) t->cap_nodes *= 2; t->nodes = (TNode*)realloc(t->nodes, sizeof(TNode) * t->cap_nodes); }
    t->nodes[node].first_child = t->n_nodes; t->nodes[node].n_children = k < 256 ? k : 256;
    for (uint32_t c = 0; c < k && c < 256; ++c) { tnode_init(t, &t->nodes[t->n_nodes], next[c], node); ++t->n_nodes; }
}

/* ---- possible scores ---- */
static PScore* ps_find(const Tree* t, int32_t score) {
    for (uint32_t i = 0; i < t->n_ps; ++i) if (t->ps[i].score == score) return &t->ps[i];
    return NULL;
}
static PScore* ps_insert(Tree* t, PScore v) {
    if (t->n_ps == t->cap_ps) { t->cap_ps *= 2; t->ps = (PScore*)realloc(t->ps, sizeof(PScore) * t->cap_ps); }
    uint32_t i = t->n_ps++;
    while (i > 0 && t->ps[i - 1].score > v.score) { t->ps[i] = t->ps[i - 1]; --i; }
    t->ps[i] = v;
    return &t->ps[i];
}
static int32_t gap_penalty(const Tree* t, uint32_t length) { return t->gap_open + (int32_t)length * t->gap_extend; }  /* :1649 */

static uint32_t get_leaves(const Tree* t, uint32_t* out) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < t->n_nodes; ++i) if (is_leaf(&t->nodes[i])) out[k++] = i;
    return k;
}

static void match_forward(const Tree* t, const TNode* n, MPos* p) {                               /* :1533-1542 */
    while (p->seq < t->L && p->off < n->len && t->seq[p->seq] == n->seq[p->off]) { p->seq++; p->off++; }
}

static void extend_over(Tree* t, int32_t score, int32_t diag, const uint32_t* leaves, uint32_t n_leaves) {             /* :1874-1935 */
    for (uint32_t li = 0; li < n_leaves; ++li) {
        const uint32_t leaf = leaves[li];
        MPos pos = find_pos(t, W_MATCH, leaf, score, diag, 0, 0);
        if (pos.empty) continue;
        for (;;) {
            TNode* node = &t->nodes[pos.cur];
            const int may_reach_target = node->target_offset >= pos.off && node->target_offset < node->len;
            match_forward(t, node, &pos);
            if ((may_reach_target && pos.off >= node->target_offset) || (t->no_to && pos.seq >= t->L)) {
                const uint32_t overshoot = t->no_to ? 0 : pos.off - node->target_offset;
                const uint32_t gap_length = (t->L - pos.seq) + overshoot;
                const int32_t gap_score = gap_length > 0 ? gap_penalty(t, gap_length) : 0;
                if (score + gap_score < t->cand.score) {
                    t->cand = (Point){ score + gap_score, diag, pos.seq - overshoot, node->target_offset };
                    t->cand_node = pos.cur;
                }
            }
            if (mp_distance(pos, diag) > t->max_distance) t->max_distance = mp_distance(pos, diag);
            update(t, W_MATCH, score, diag, pos);
            if (pos.off < node->len) break;
            expand_if_necessary(t, pos);                                                          /* may move t->nodes */
            if (mp_at_last(pos)) {
                const uint32_t nc = t->nodes[pos.cur].n_children, fc = t->nodes[pos.cur].first_child;
                uint32_t* kids = (uint32_t*)malloc(sizeof(uint32_t) * (nc + 1));
                for (uint32_t c = 0; c < nc; ++c) kids[c] = fc + c;
                extend_over(t, score, diag, kids, nc);
                free(kids);
                break;
            }
            mp_pop(t, &pos); pos.off = 0;
        }
    }
}

static void extend(Tree* t, int32_t score) {                                                      /* :1656-1666 */
    const PScore* p = ps_find(t, score);
    if (!p) return;
    const int32_t lo = p->min_d, hi = p->max_d;
    for (int64_t diag = lo; diag <= hi; ++diag) {
        uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * (t->n_nodes + 1));
        const uint32_t n = get_leaves(t, leaves);
        extend_over(t, score, (int32_t)diag, leaves, n);
        free(leaves);
    }
}

static int32_t next_score(Tree* t, int32_t match_score) {                                         /* :1672-1704 */
    const int32_t mismatch_score = match_score + t->mismatch;
    if (!ps_find(t, mismatch_score)) ps_insert(t, (PScore){ mismatch_score, 0, 0, 0 });
    if (ps_find(t, match_score)->gap) {
        const int32_t extend_score = match_score + t->gap_extend;
        PScore* e = ps_find(t, extend_score);
        if (e) e->gap = 1; else ps_insert(t, (PScore){ extend_score, 0, 0, 1 });
    }
    const int32_t open_score = match_score + t->gap_open + t->gap_extend;
    PScore* o = ps_find(t, open_score);
    if (o) o->gap = 1; else ps_insert(t, (PScore){ open_score, 0, 0, 1 });
    const PScore* m = ps_find(t, match_score);
    return (m + 1)->score;
}

static void update_range(const Tree* t, int32_t* lo, int32_t* hi, int32_t score) {               /* :1956-1968 */
    if (score < 0) return;
    const PScore* p = ps_find(t, score);
    if (!p) return;
    if (p->min_d < *lo) *lo = p->min_d;
    if (p->max_d > *hi) *hi = p->max_d;
}

static void next(Tree* t, int32_t score) {                                                        /* :1709-1786 */
    int32_t lo = INT32_MAX, hi = INT32_MIN;                                                       /* get_diagonals :1977-1988 */
    update_range(t, &lo, &hi, score - t->mismatch);
    update_range(t, &lo, &hi, score - t->gap_open - t->gap_extend);
    update_range(t, &lo, &hi, score - t->gap_extend);
    if (lo <= hi) { --lo; ++hi; }
    int32_t alo = INT32_MAX, ahi = INT32_MIN;
    #define ADJUST(d) do { if ((d) < alo) alo = (d); if ((d) > ahi) ahi = (d); } while (0)
    for (int64_t d64 = lo; d64 <= hi; ++d64) {
        const int32_t diag = (int32_t)d64;
        uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * (t->n_nodes + 1));
        const uint32_t n_leaves = get_leaves(t, leaves);
        for (uint32_t li = 0; li < n_leaves; ++li) {
            const uint32_t leaf = leaves[li];
            int edit;
            MPos ins = ins_predecessor(t, leaf, score, diag, &edit);
            if (!ins.empty) {
                ins.seq++;
                if (mp_distance(ins, diag) >= t->min_distance) { update(t, W_INS, score, diag, ins); ADJUST(diag); }
            }
            MPos del = del_predecessor(t, leaf, score, diag, &edit);
            if (!del.empty) {
                successor_offset(t, &del);
                if (mp_distance(del, diag) >= t->min_distance) { update(t, W_DEL, score, diag, del); ADJUST(diag); }
                expand_if_necessary(t, del);
            }
            MPos subst = find_pos(t, W_MATCH, leaf, score - t->mismatch, diag, 1, 1);
            if (!subst.empty) { subst.seq++; successor_offset(t, &subst); expand_if_necessary(t, subst); }
            if (mp_less(subst, ins)) subst = ins;
            if (mp_less(subst, del)) subst = del;
            if (!subst.empty) {
                const TNode* node = &t->nodes[subst.cur];
                if (subst.off == node->target_offset) {
                    const uint32_t gap_length = t->L - subst.seq;
                    const int32_t gap_score = gap_length > 0 ? gap_penalty(t, gap_length) : 0;
                    if (score + gap_score < t->cand.score) { t->cand = (Point){ score + gap_score, diag, subst.seq, subst.off }; t->cand_node = subst.cur; }
                }
                if (mp_distance(subst, diag) >= t->min_distance) { update(t, W_MATCH, score, diag, subst); ADJUST(diag); }
            }
        }
        free(leaves);
    }
    #undef ADJUST
    PScore* p = ps_find(t, score);
    if (p) { p->min_d = alo; p->max_d = ahi; }
}

static int32_t alignment_score(const Tree* t, Point p, uint32_t final_insertion) {                /* :1380-1387 */
    const int32_t target_offset = (int32_t)p.seq - p.diag;
    return (t->match * ((int32_t)(p.seq + final_insertion) + target_offset) - p.score) / 2;
}

static void trim(Tree* t) {                                                                       /* :1849-1868 */
    t->cand = (Point){ 0, 0, 0, 0 }; t->cand_node = 0;
    int32_t best = 0; int have = 0;
    for (uint32_t node = 0; node < t->n_nodes; ++node) {
        const PMap* m = &t->nodes[node].wf[W_MATCH];
        for (uint32_t i = 0; i < m->cap; ++i) {
            if (!m->s[i].used) continue;
            const Point p = { m->s[i].score, m->s[i].diag, m->s[i].seq, m->s[i].off };
            const int32_t as = alignment_score(t, p, 0);
            const int better = as > best || (have && as == best && node == t->cand_node &&
                                             (p.score < t->cand.score || (p.score == t->cand.score && p.diag < t->cand.diag)));
            if (better) { t->cand = p; t->cand_node = node; best = as; have = 1; }
        }
    }
}

static int32_t evaluate(const vgk_wfa_event* e, uint32_t length) {                                /* gbwt_extender.hpp:371-373 */
    const int32_t v = (int32_t)(e->per_base * (double)length) + e->min;
    return v < e->max ? v : e->max;
}

static const vgk_wfa_error_model default_model = { { 0.03, 1, 6 }, { 0.05, 1, 10 }, { 0.1, 1, 20 }, { 0.1, 10, 200 } };

typedef struct { int32_t* path; uint32_t n_path, cap_path; uint32_t* edits; uint32_t n_edits, cap_edits; } Out;
static void out_append(Out* o, int edit, uint32_t length) {                                       /* WFAAlignment::append :850-859 */
    if (!length) return;
    if (o->n_edits && (o->edits[o->n_edits - 1] & 3u) == (uint32_t)edit) { o->edits[o->n_edits - 1] += length << 2; return; }
    if (o->n_edits == o->cap_edits) { o->cap_edits = o->cap_edits ? 2 * o->cap_edits : 16; o->edits = (uint32_t*)realloc(o->edits, sizeof(uint32_t) * o->cap_edits); }
    o->edits[o->n_edits++] = (length << 2) | (uint32_t)edit;
}
static void out_push_node(Out* o, int32_t node) {
    if (o->n_path == o->cap_path) { o->cap_path = o->cap_path ? 2 * o->cap_path : 16; o->path = (int32_t*)realloc(o->path, sizeof(int32_t) * o->cap_path); }
    o->path[o->n_path++] = node;
}
static int64_t final_offset(const vgk_haplo* h, const Out* o, uint32_t node_offset) {             /* :821-832 */
    int64_t f = node_offset;
    for (uint32_t i = 0; i < o->n_edits; ++i) if ((o->edits[i] & 3u) != E_INS) f += o->edits[i] >> 2;
    for (uint32_t i = 0; i + 1 < o->n_path; ++i) f -= h->len[o->path[i]];
    return f;
}

/* WFAExtender::connect :2052-2235 on a masked sequence.  Returns ok. */
static int wfa_connect(const vgk_scoring* sc, const vgk_haplo* h, const vgk_wfa_error_model* em, const char* seq, uint32_t L,
                       uint32_t from_node, uint32_t from_off, uint32_t to_node, uint32_t to_off, vgk_wfa_result* res, Out* out) {
    memset(res, 0, sizeof *res);
    out->n_path = out->n_edits = 0;
    if (from_node >= h->n_oriented) return 0;                                                     /* !has_node(id(from)) */
    Tree t; memset(&t, 0, sizeof t);
    t.h = h; t.seq = seq; t.L = L;
    t.no_to = to_node == VGK_WFA_NO_NODE; t.to_node = (int32_t)to_node; t.to_off = to_off;
    const int32_t match = sc->matrix[0], mism = -sc->matrix[1];
    t.match = match;
    t.mismatch = 2 * (match + mism);                                                              /* :1616-1618 */
    t.gap_open = 2 * ((int32_t)sc->gap_open - (int32_t)sc->gap_extend);
    t.gap_extend = 2 * (int32_t)sc->gap_extend + match;
    t.cand = (Point){ INT32_MAX, 0, 0, 0 };
    t.cap_nodes = 16; t.nodes = (TNode*)malloc(sizeof(TNode) * t.cap_nodes);
    t.cap_ps = 64; t.ps = (PScore*)malloc(sizeof(PScore) * t.cap_ps);
    SState root = { (int32_t)from_node, 0, (int32_t)h->count[from_node] - 1 };
    tnode_init(&t, &t.nodes[0], root, 0); t.n_nodes = 1;
    pm_put(&t.nodes[0].wf[W_MATCH], 0, 0, 0, from_off + 1);
    t.score_bound = evaluate(&em->mismatches, L) * t.mismatch + evaluate(&em->gaps, L) * t.gap_open + evaluate(&em->gap_length, L) * t.gap_extend;
    ps_insert(&t, (PScore){ 0, 0, 0, 0 });

    int32_t score = 0;
    for (;;) {
        extend(&t, score);
        const int32_t band = evaluate(&em->distance, L);
        if (band < t.max_distance) t.min_distance = t.max_distance - band;
        if (t.cand.score <= score) break;
        score = next_score(&t, score);
        if (score > t.score_bound) break;
        next(&t, score);
    }

    int ok = 1;
    uint32_t unaligned_tail = L - t.cand.seq;
    if (t.cand.score > t.score_bound) {
        unaligned_tail = 0;
        if (t.no_to) trim(&t); else ok = 0;
    }
    if (ok) {
        res->ok = 1;
        res->node_offset = from_off + 1; res->seq_offset = 0;
        res->length = t.cand.seq + unaligned_tail;
        res->score = alignment_score(&t, t.cand, unaligned_tail);
        for (uint32_t node = t.cand_node;; node = t.nodes[node].parent) {
            for (uint32_t i = t.nodes[node].path_len; i-- > 0;) out_push_node(out, t.nodes[node].path[i]);
            if (node == 0) break;
        }
        for (uint32_t i = 0, j = out->n_path; i + 1 < j; ++i) { --j; const int32_t x = out->path[i]; out->path[i] = out->path[j]; out->path[j] = x; }
        Point point = t.cand; uint32_t node = t.cand_node;
        if (unaligned_tail > 0) { out_append(out, E_INS, L - t.cand.seq); point.score -= gap_penalty(&t, unaligned_tail); }
        int edit = E_MATCH, lost = 0;
        while ((point.seq > 0 || point.diag != 0) && !lost) {
            MPos pred; int pe;
            switch (edit) {
            case E_MATCH:
                pred = match_predecessor(&t, node, point.score, point.diag, &pe);
                if (pred.empty && (point.score != 0 || point.diag != 0)) { lost = 1; break; }
                out_append(out, E_MATCH, point.seq - pred.seq);
                point.seq = pred.seq; point.off = pred.off;
                if (!pred.empty) node = pred.cur;
                edit = pe; break;
            case E_MISMATCH:
                out_append(out, E_MISMATCH, 1);
                point.seq--; predecessor_offset(&t, &node, &point.off);
                point.score -= t.mismatch; edit = E_MATCH; break;
            case E_INS:
                pred = ins_predecessor(&t, node, point.score, point.diag, &pe);
                if (pred.empty) { lost = 1; break; }
                out_append(out, E_INS, 1);
                point.seq--;
                point.score -= pe == E_INS ? t.gap_extend : t.gap_open + t.gap_extend;
                point.diag--; edit = pe; break;
            default:
                pred = del_predecessor(&t, node, point.score, point.diag, &pe);
                if (pred.empty) { lost = 1; break; }
                out_append(out, E_DEL, 1);
                predecessor_offset(&t, &node, &point.off);
                point.score -= pe == E_DEL ? t.gap_extend : t.gap_open + t.gap_extend;
                point.diag++; edit = pe; break;
            }
        }
        if (lost) {
            /* A candidate found by next() is recorded before the distance check (:1761-1776): when it lies behind min_distance
               neither it nor its gap point is stored, and the reference's backtrace (:2156-2202) then walks off the stored
               wavefronts with wrapping offsets and does not terminate.  Here: no alignment, status VGK_ENOBAND. */
            memset(res, 0, sizeof *res); res->status = VGK_ENOBAND; out->n_path = out->n_edits = 0; ok = 0;
        }
        for (uint32_t i = 0, j = out->n_edits; i + 1 < j; ++i) { --j; const uint32_t x = out->edits[i]; out->edits[i] = out->edits[j]; out->edits[j] = x; }
        if (out->n_path && res->node_offset >= h->len[out->path[0]]) {                          /* :2208-2211 */
            memmove(out->path, out->path + 1, sizeof(int32_t) * (out->n_path - 1)); --out->n_path; res->node_offset = 0;
        }
        int64_t fo = final_offset(h, out, res->node_offset);                                      /* :2217-2229 */
        while ((out->n_path == 1 && fo == (int64_t)res->node_offset) || (out->n_path > 1 && fo <= 0)) {
            --out->n_path;
            if (out->n_path) fo += h->len[out->path[out->n_path - 1]];
        }
    }
    for (uint32_t i = 0; i < t.n_nodes; ++i) { free(t.nodes[i].path); free(t.nodes[i].seq); for (int k = 0; k < 3; ++k) free(t.nodes[i].wf[k].s); }
    free(t.nodes); free(t.ps);
    return ok;
}

static char mask_base(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' ? c : 'X'; }  /* ReadMasker("ACGT") */
static char comp_base(char c) {                                                                   /* reverse_complement */
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}

/* One problem.  *path_out / *edits_out are malloc'ed by this call (the caller frees them). */
int vgo_wfa_one(const vgk_scoring* sc, const vgk_haplo* h, const vgk_wfa_error_model* model, const vgk_wfa_problem* p,
                vgk_wfa_result* res, int32_t** path_out, uint32_t** edits_out) {
    const vgk_wfa_error_model* em = model ? model : &default_model;
    memset(res, 0, sizeof *res); *path_out = NULL; *edits_out = NULL;
    if (!p || (!p->seq && p->seq_len) || p->mode > VGK_WFA_PREFIX) { res->status = VGK_EINVAL; return VGK_EINVAL; }
    const int32_t match = sc->matrix[0], mism = -sc->matrix[1];
    if (match < 0 || mism <= 0 || sc->gap_open < sc->gap_extend || sc->gap_extend == 0) { res->status = VGK_EUNSUPPORTED; return VGK_EUNSUPPORTED; }
    const uint32_t L = p->seq_len;
    char* seq = (char*)malloc(L + 1);
    Out out; memset(&out, 0, sizeof out);
    uint32_t from_node = p->from_node, from_off = p->from_offset, to_node = p->to_node, to_off = p->to_offset;
    if (p->mode == VGK_WFA_PREFIX) {                                                              /* :2248-2263 */
        if (p->to_node >= h->n_oriented || p->to_offset >= h->len[p->to_node]) { free(seq); res->status = VGK_EINVAL; return VGK_EINVAL; }
        from_node = p->to_node ^ 1u; from_off = (h->len[p->to_node] - 1) - p->to_offset;          /* reverse_base_pos, types.hpp:89 */
        to_node = VGK_WFA_NO_NODE; to_off = 0;
        for (uint32_t i = 0; i < L; ++i) seq[i] = mask_base(comp_base(p->seq[L - 1 - i]));
    } else {
        if (p->mode == VGK_WFA_SUFFIX) { to_node = VGK_WFA_NO_NODE; to_off = 0; }
        if ((from_node < h->n_oriented && from_off >= h->len[from_node]) || (to_node < h->n_oriented && to_off >= h->len[to_node])) {
            free(seq); res->status = VGK_EINVAL; return VGK_EINVAL;                               /* a position past its node */
        }
        for (uint32_t i = 0; i < L; ++i) seq[i] = mask_base(p->seq[i]);
    }
    const int ok = wfa_connect(sc, h, em, seq, L, from_node, from_off, to_node, to_off, res, &out);
    free(seq);
    if (ok && p->mode == VGK_WFA_PREFIX) {                                                        /* WFAAlignment::flip :834-848 */
        res->seq_offset = L - res->seq_offset - res->length;
        if (out.n_path) {
            res->node_offset = (uint32_t)((int64_t)h->len[out.path[out.n_path - 1]] - final_offset(h, &out, res->node_offset));
            for (uint32_t i = 0, j = out.n_path; i < j; ++i) { --j; const int32_t x = out.path[i] ^ 1, y = out.path[j] ^ 1; out.path[i] = y; out.path[j] = x; if (i == j) out.path[i] = x; }
            for (uint32_t i = 0, j = out.n_edits; i + 1 < j; ++i) { --j; const uint32_t x = out.edits[i]; out.edits[i] = out.edits[j]; out.edits[j] = x; }
        }
    }
    if (ok && p->mode != VGK_WFA_CONNECT && out.n_edits && res->length == L) {                    /* :2240-2243, :2258-2260 */
        const uint32_t e = (p->mode == VGK_WFA_SUFFIX ? out.edits[out.n_edits - 1] : out.edits[0]) & 3u;
        if (e == E_MATCH || e == E_MISMATCH) res->score += sc->full_length_bonus;
    }
    res->path_len = out.n_path; res->n_edits = out.n_edits;
    *path_out = out.path; *edits_out = out.edits;
    return VGK_OK;
}
