// wfa_device.hpp — haplotype-consistent wavefront alignment, one thread per problem.
//
// Follows WFAExtender::connect (reference src/gbwt_extender.cpp:2052-2235) over its WFATree (:1567-2046) and WFANode
// (:1434-1557).  The reference keeps, per trie node, three hash maps (score, diagonal) -> (sequence offset, node offset)
// and grows std::vectors; here a problem owns one fixed scratch slab:
//   * ONE open-addressing table for all wavefront points of the problem, keyed by (trie node, kind, penalty, diagonal) in
//     32 bits with the two offsets packed in the other 32 — a probe is one 8-byte load.  The slab's table is all-zero between
//     problems: every insertion is logged and the log is replayed to clear exactly the touched slots.
//   * trie nodes as fixed records; the graph nodes along their paths are chained through one pool with their starting offsets,
//     so the bases of a trie node are read straight from the index (eight per compare) instead of being copied into a string.
//   * LAZY trie nodes.  WFANode's constructor walks a non-branching path for up to 1024 bases (:1470-1487) although a tail of
//     100 bases needs four nodes of it; every step is a dependent record fetch from HBM.  Here a trie node grows one graph
//     node at a time, exactly when a position asks whether it is past the node's end (w_past_end): the walk rule, the
//     1024-base stop, the target and the dead-end flag are evaluated by the same code, just later, and nothing observes a
//     node's length except through that question — so the results are those of the eager walk (the oracle walks eagerly).
//   * possible penalties direct-mapped by value (the reference's std::map), diagonal range and "reachable with a gap" each.
// A position (MatchPos :1276-1365) is (sequence offset, node offset, current trie node, the trie node the lookup started
// at): its stack of trie offsets is the tree path between the two.  Children of a trie node are consecutive (they are
// created together, :1999-2004).  The recursion of extend_over (:1928) is an explicit stack of child ranges.
//   * The work per problem is heavily skewed (an exact match costs a handful of table probes, the median 7, the 99th percentile
//     270, the maximum beyond 4000) and a wavefront runs as long as its slowest lane, so problems are handed out one at a time
//     from a device counter instead of in fixed strides: a thread stuck on a slow problem holds nobody else's.  (Measured and
//     dropped: budgeted passes that defer slow problems to a later, compacted launch — every wavefront still waits for a lane
//     that runs into the budget, and the deferred problems run twice.)
// Anything that outgrows the slab ends the problem with VGK_ETOOBIG.
#pragma once
#include <stdint.h>
#include "../../include/vgk.h"
#include "gapless_device.hpp"

namespace vgk {

constexpr int W_NODES  = 32;         // trie nodes per problem (a 32-bit mask holds the leaves)
constexpr int W_PATH   = 256;        // graph nodes over all trie nodes
constexpr int W_SLOTS  = 2048;       // wavefront table (power of two)
constexpr int W_POINTS = 1024;       // stored points per problem
constexpr int W_SCORES = 512;        // penalties 0 .. W_SCORES - 1 (the host checks the score bound against this)
constexpr int W_EDITS  = 160;        // runs of edits per alignment
constexpr uint32_t W_TARGET_LENGTH = 1024;   // WFANode::TARGET_LENGTH
constexpr uint32_t W_NO_OFFSET = 0xffffffffu;
enum { WK_MATCH = 0, WK_INS = 1, WK_DEL = 2 };

// The sequence of problem i as the kernels read it, made on the device from the caller's bytes (wfa_api.cpp: the caller's sequences in one
// stretch of memory, uploaded as it is): ReadMasker's bytes (reference src/gbwt_extender.cpp:160-170: anything but ACGT never matches), a PREFIX
// problem's reverse complement (:2248-2255).  Lane `lane` of `lanes` takes every lanes-th base.
VGK_HD char wfa_mask_base(char c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'X'; }
VGK_HD char wfa_complement_base(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'X'; }
struct WProb;
VGK_HD void wfa_mask_one(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, uint32_t i, uint32_t lane, uint32_t lanes);

struct WProb {                        // packed by the host
    uint32_t seq_off, seq_len;
    uint32_t mode;
    uint32_t from_node, from_off, to_node, to_off;
    int32_t  score_bound, distance_band;
    int32_t  status;                  // problems the host refused keep their status
};
VGK_HD void wfa_mask_one(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, uint32_t i, uint32_t lane, uint32_t lanes) {
    const WProb w = probs[i];
    const char* src = raw + src_off[i]; char* dst = seqs + w.seq_off;
    if (w.mode == 2u /* VGK_WFA_PREFIX */) for (uint32_t k = lane; k < w.seq_len; k += lanes) dst[k] = wfa_complement_base(src[w.seq_len - 1 - k]);
    else for (uint32_t k = lane; k < w.seq_len; k += lanes) dst[k] = wfa_mask_base(src[k]);
}

struct WNode {
    int32_t  st_node, st_lo, st_hi;   // search state at the end of the (materialised) path
    uint32_t len;                     // bases along the materialised path
    uint32_t target_offset;
    uint16_t path_head, path_tail;    // chain through WScratch::path_next
    uint8_t  parent, first_child, n_children, dead_end;
    uint8_t  complete, pad[3];        // the walk has ended: branch, dead end, target, or 1024 bases
    uint32_t ancestors;               // bit per trie node on the way to the root, this node included
};
constexpr uint16_t W_NIL = 0xffffu;
struct WPScore { int16_t min_d, max_d; uint8_t flags; };             // flags: 1 = possible, 2 = reachable with a gap

struct WScratch {
    uint64_t slot[W_SLOTS];           // (key + 1) << 32 | seq << 16 | off ; 0 = free
    uint16_t log[W_POINTS];
    WNode    nodes[W_NODES];
    uint32_t node_end[W_NODES];       // nodes[i].len | complete << 31: what "is this offset past the node's end?" reads — the most frequent
                                      // question of all; the HIP kernel keeps these words in LDS instead (WCtx::end / end_stride)
    int32_t  path_node[W_PATH];
    uint16_t path_start[W_PATH];      // offset of the graph node inside its trie node
    uint16_t path_next[W_PATH];
    uint32_t ps_range[W_SCORES];      // possible penalties: min diagonal | max diagonal << 16 (int16 each) ...
    uint8_t  ps_flags[W_SCORES];      // ... and flags; only the flags are cleared per problem
    uint32_t edits[W_EDITS];
    uint8_t  chain[W_NODES];
    uint8_t  stack_cur[W_NODES], stack_end[W_NODES];
};

struct WfaParams {
    GIndex index;
    const WProb* probs; uint32_t n;
    const uint32_t* order;            // the order problems are handed out in (longest sequences first, neighbours in the graph together)
    const char* seqs;                 // masked: ACGT or X; 8 bytes of padding at either end
    int32_t match, mismatch, gap_open, gap_extend, bonus;      // match/bonus as scored; the other three are WFA penalties (:1616-1618)
    WScratch* scratch;                // one per resident thread
    vgk_wfa_result* results;
    uint32_t* paths; uint32_t* edits;
    unsigned long long* counters;     // [0] path entries, [1] edits handed out, [2] next problem
    unsigned long long caps[2];
    uint32_t max_points;              // stored wavefront points after which a connect is given up (<= W_POINTS; vgk_wfa_set_point_budget) ...
    uint32_t max_points_tail;         // ... and a prefix / suffix (vgk_wfa_set_point_budgets; a declined tail has no banded fallback between two anchors)
    // hybrid form: the thread kernel gives a problem up at `hand_over_points` stored points (when the caller's own budget lies above that)
    // and lists it for the wavefront kernel (wfa_wave_device.hpp) instead of reporting it
    uint32_t hand_over_points; uint32_t* handed_over; unsigned long long* n_handed_over;
    // ... and, when the two kernels run AT ONCE (Backend::run_wfa_hybrid): a counter every wavefront of the thread kernel bumps when it has
    // run dry — the wavefront kernel, polling the list, learns from it that nothing more will come.  Null when the launches follow each other.
    uint32_t* producers_done;
};

struct WPos { uint32_t seq, off; uint8_t cur, origin; bool empty; };
VGK_HD WPos w_none() { WPos p = { 0, 0, 0, 0, true }; return p; }
VGK_HD bool w_less(const WPos& a, const WPos& b) { if (a.empty) return !b.empty; if (b.empty) return false; return a.seq < b.seq; }    // (:1356-1364)
VGK_HD int32_t w_distance(const WPos& p, int32_t diag) { return 2 * (int32_t)p.seq - diag; }

struct WCtx {
    const WfaParams* P; WScratch* S;
    uint32_t* end; uint32_t end_stride;                           // node_end word of trie node i at end[i * end_stride]
    const char* seq; uint32_t L;
    int32_t to_node; uint32_t to_off; bool no_to;
    uint32_t n_nodes, n_path, n_points, max_points;
    uint32_t slot_mask;               // the part of the slab's table this problem uses: four slots per point it may store (a problem the hybrid form gives up at 16 points
                                      // keeps to 64 slots = four cache lines of its 16 KB table: a probe then hits a line the problem has touched before)
    uint32_t leaves;                  // trie nodes without children (WFANode::is_leaf: a dead end has none either)
    int32_t cand_score, cand_diag; uint32_t cand_seq, cand_off, cand_node;
    int32_t max_distance, min_distance;
    bool overflow; int why;         // why: 1 points, 2 trie nodes, 3 path pool, 4 edits, 5 node length
};

// ---- possible penalties (the reference's possible_scores map, :1596-1606), direct-mapped by value ----
VGK_HD WPScore w_ps(const WCtx& c, int32_t score) {
    WPScore p; p.flags = c.S->ps_flags[score];
    const uint32_t r = c.S->ps_range[score];
    p.min_d = (int16_t)(r & 0xffffu); p.max_d = (int16_t)(r >> 16);
    return p;
}
VGK_HD void w_ps_set_range(WCtx& c, int32_t score, int32_t lo, int32_t hi) { c.S->ps_range[score] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// ---- the wavefront table ----
// Hashed by (kind, penalty, diagonal) WITHOUT the trie node, so the points every trie node holds for one wavefront cell sit in one
// probe sequence: "the deepest ancestor of this leaf that has the cell" (WFATree::find_pos walks leaf -> root, one hash_map each)
// is a single scan to the first free slot, picking the largest trie node whose bit is set in the leaf's ancestor mask — trie
// nodes are numbered in creation order, so a deeper ancestor has the larger number.
VGK_HD uint32_t w_key(uint32_t node, int kind, int32_t score, int32_t diag) { return 1u + (node | ((uint32_t)kind << 5) | ((uint32_t)score << 7) | ((uint32_t)(diag + 512) << 17)); }
VGK_HD uint32_t w_hash(uint32_t key) { return (((key - 1u) >> 5) * 2654435761u) >> 21; }         // 11 bits = W_SLOTS; the node bits stay out
VGK_HD bool w_lookup(WCtx& c, uint32_t ancestors, int kind, int32_t score, int32_t diag, uint32_t& node, uint32_t& seq, uint32_t& off) {
    const uint32_t cell = (w_key(0, kind, score, diag) - 1u) >> 5;
    bool found = false; uint32_t best = 0;
    for (uint32_t i = w_hash(w_key(0, kind, score, diag)) & c.slot_mask;; i = (i + 1) & c.slot_mask) {
        const uint64_t s = c.S->slot[i];
        if (!s) break;
        const uint32_t key = (uint32_t)(s >> 32) - 1u, holder = key & 31u;
        if ((key >> 5) == cell && ((ancestors >> holder) & 1u) && (!found || holder > best)) {
            found = true; best = holder; seq = (uint32_t)(s >> 16) & 0xffffu; off = (uint32_t)s & 0xffffu;
        }
    }
    node = best;
    return found;
}
VGK_HD void w_store(WCtx& c, uint32_t node, int kind, int32_t score, int32_t diag, uint32_t seq, uint32_t off) {                     // WFANode::update (:1517-1530)
    const uint32_t key = w_key(node, kind, score, diag);
    const uint64_t v = ((uint64_t)key << 32) | ((uint64_t)(seq & 0xffffu) << 16) | (off & 0xffffu);
    for (uint32_t i = w_hash(key) & c.slot_mask;; i = (i + 1) & c.slot_mask) {
        const uint64_t s = c.S->slot[i];
        if (!s) {
            if (c.n_points >= c.max_points) { c.overflow = true; c.why = 1; return; }
            c.S->log[c.n_points++] = (uint16_t)i; c.S->slot[i] = v; return;
        }
        if ((uint32_t)(s >> 32) == key) { c.S->slot[i] = v; return; }
    }
}

// ---- search states: the non-empty one-node extensions in the order of the record's edges (follow_paths) ----
struct WState { int32_t node, lo, hi; };
// visits the non-empty extensions in order, copies number `want` into `out`, stops after `stop_at` of them; returns how many it saw
VGK_HD uint32_t w_follow_rec(const uint32_t* rec, const WState& s, uint32_t want, WState& out, uint32_t stop_at, uint32_t* edge_out);
VGK_HD uint32_t w_follow(const GIndex& h, const WState& s, uint32_t want, WState& out, uint32_t stop_at) {
    if (s.lo > s.hi) return 0;
    uint32_t edge = 0;
    return w_follow_rec(g_rec(h, (uint32_t)s.node), s, want, out, stop_at, &edge);
}
// the same over the node's record at hand; *edge_out: the edge extension number `want` leaves through
VGK_HD uint32_t w_follow_rec(const uint32_t* rec, const WState& s, uint32_t want, WState& out, uint32_t stop_at, uint32_t* edge_out) {
    if (s.lo > s.hi) return 0;
    const uint32_t ne = g_ne(rec);
    const uint32_t* body = g_visits(rec);
    const bool few = ne <= 4;
    const bool rle = g_rle(rec);
    const GCounts cn = few ? g_counts(rec, s.lo, s.hi) : GCounts{0, 0};     // one pass over the visits serves every edge
    uint32_t k = 0;
    for (uint32_t e = 0; e < ne && k < stop_at; ++e) {
        const int32_t to = ge_to(rec, e);
        if (to < 0) continue;
        int32_t before = 0, inside = 0;
        if (few) { before = (int32_t)g_count_of(cn.before, e); inside = (int32_t)g_count_of(cn.inside, e); }
        else if (rle) {
            int32_t pos = 0;
            for (uint32_t k = 0; pos <= s.hi; ++k) {
                const uint32_t run = body[k]; const int32_t end = pos + (int32_t)(run >> 8);
                if ((run & 0xffu) == e) {
                    const int32_t nb = (end < s.lo ? end : s.lo) - pos, last = end - 1 < s.hi ? end - 1 : s.hi, first = pos > s.lo ? pos : s.lo;
                    if (nb > 0) before += nb;
                    if (last >= first) inside += last - first + 1;
                }
                pos = end;
            }
        }
        else for (int32_t i = 0; i <= s.hi; ++i) if (g_body(body, (uint32_t)i) == e) { if (i < s.lo) ++before; else ++inside; }
        if (!inside) continue;
        if (k == want) { out.node = to; out.lo = (int32_t)ge_base(rec, e) + before; out.hi = out.lo + inside - 1; *edge_out = e; }
        ++k;
    }
    return k;
}

// ---- trie nodes ----
VGK_HD bool w_append_node(WCtx& c, WNode& n, const WState& next) {                                // (:1546-1556)
    n.st_node = next.node; n.st_lo = next.lo; n.st_hi = next.hi;
    if (c.n_path >= (uint32_t)W_PATH) { c.overflow = true; c.why = 3; return true; }
    const uint16_t at = (uint16_t)c.n_path++;
    c.S->path_node[at] = next.node; c.S->path_start[at] = (uint16_t)n.len; c.S->path_next[at] = W_NIL;
    if (n.path_head == W_NIL) n.path_head = at; else c.S->path_next[n.path_tail] = at;
    n.path_tail = at;
    const uint32_t nl = g_len(c.P->index, next.node);
    n.len += nl;
    if (n.len > 0xfff0u) { c.overflow = true; c.why = 5; return true; }
    if (!c.no_to && c.to_node == next.node) { n.target_offset = n.len - (nl - c.to_off); return true; }
    return false;
}
VGK_HD void w_node_init(WCtx& c, uint32_t id, const WState& state, uint32_t parent) {             // (:1463-1468); the walk itself is w_grow
    WNode n;
    n.len = 0; n.target_offset = W_NO_OFFSET; n.path_head = n.path_tail = W_NIL;
    n.parent = (uint8_t)parent; n.first_child = 0; n.n_children = 0; n.dead_end = 0; n.pad[0] = n.pad[1] = n.pad[2] = 0;
    n.ancestors = (id ? c.S->nodes[parent].ancestors : 0u) | (1u << id);
    c.leaves |= 1u << id;
    n.complete = w_append_node(c, n, state) ? 1 : 0;
    c.S->nodes[id] = n; c.end[id * c.end_stride] = n.len | ((uint32_t)n.complete << 31);
}
// one turn of the constructor's loop (:1470-1487)
VGK_HD void w_grow(WCtx& c, uint32_t id) {
    WNode n = c.S->nodes[id];
    if (n.len >= W_TARGET_LENGTH) n.complete = 1;
    else {
        const WState cur = { n.st_node, n.st_lo, n.st_hi }; WState next = { 0, 0, -1 };
        const uint32_t successors = w_follow(c.P->index, cur, 0, next, 2);
        if (successors == 0) { n.dead_end = 1; n.complete = 1; }
        else if (successors > 1) n.complete = 1;
        else if (w_append_node(c, n, next)) n.complete = 1;
    }
    c.S->nodes[id] = n; c.end[id * c.end_stride] = n.len | ((uint32_t)n.complete << 31);
}
// is `off` at or past the end of the trie node?  Grows the node until that is known.
VGK_HD bool w_past_end(WCtx& c, uint32_t id, uint32_t off) {
    uint32_t e = c.end[id * c.end_stride];
    while (!(e >> 31) && (e & 0x7fffffffu) <= off) { w_grow(c, id); e = c.end[id * c.end_stride]; }
    return off >= (e & 0x7fffffffu);
}

VGK_HD void w_pop(const WCtx& c, WPos& p) {                        // one step down the tree path towards the origin:
    const uint32_t below = c.S->nodes[p.origin].ancestors & ~((2u << p.cur) - 1u);     // the origin's ancestors numbered above the current node;
    p.cur = (uint8_t)__builtin_ctz(below);                         // trie nodes are numbered in creation order, so the nearest one is the child
}
VGK_HD bool w_at_dead_end(WCtx& c, const WPos& p) { return w_past_end(c, p.cur, p.off) && c.S->nodes[p.cur].dead_end; }   // (:2043)

// A source wavefront: one penalty and the diagonal range its points were stored in.  No point of a penalty was ever stored
// outside the range its wavefront ended up with (next :1780-1785, extend :1662), so a lookup outside it needs no probe.
struct WSrc { int32_t score; int32_t lo, hi; };                   // lo > hi: nothing stored
VGK_HD WSrc w_src(const WCtx& c, int32_t score) {
    WSrc s = { score, 1, 0 };
    if (score < 0) return s;
    const WPScore ps = w_ps(c, score);
    if (ps.flags & 1) { s.lo = ps.min_d; s.hi = ps.max_d; }
    return s;
}
// WFATree::find_pos (:2015-2040) from a trie node with the given ancestor mask
VGK_HD WPos w_find_in(WCtx& c, int kind, const WSrc& src, uint32_t ancestors, uint32_t origin, int32_t diag, bool ext_seq, bool ext_graph) {
    if (diag < src.lo || diag > src.hi) return w_none();
    uint32_t holder = 0, seq = 0, off = 0;
    if (!w_lookup(c, ancestors, kind, src.score, diag, holder, seq, off)) return w_none();
    WPos p = { seq, off, (uint8_t)holder, (uint8_t)origin, false };
    if (ext_seq && p.seq >= c.L) return w_none();
    if (ext_graph && w_at_dead_end(c, p)) return w_none();
    return p;
}
VGK_HD WPos w_find_pos(WCtx& c, int kind, uint32_t node, int32_t score, int32_t diag, bool ext_seq, bool ext_graph) {
    return w_find_in(c, kind, w_src(c, score), c.S->nodes[node].ancestors, node, diag, ext_seq, ext_graph);
}
VGK_HD void w_update(WCtx& c, int kind, int32_t score, int32_t diag, const WPos& p) { w_store(c, p.cur, kind, score, diag, p.seq, p.off); }

VGK_HD WPos w_ins_predecessor(WCtx& c, uint32_t node, int32_t score, int32_t diag, int& edit) {                               // (:1791-1795)
    const WPos open = w_find_pos(c, WK_MATCH, node, score - c.P->gap_open - c.P->gap_extend, diag - 1, true, false);
    const WPos ext = w_find_pos(c, WK_INS, node, score - c.P->gap_extend, diag - 1, true, false);
    if (w_less(open, ext)) { edit = VGK_WFA_INSERTION; return ext; }
    edit = VGK_WFA_MATCH; return open;
}
VGK_HD WPos w_del_predecessor(WCtx& c, uint32_t node, int32_t score, int32_t diag, int& edit) {                               // (:1800-1804)
    const WPos open = w_find_pos(c, WK_MATCH, node, score - c.P->gap_open - c.P->gap_extend, diag + 1, false, true);
    const WPos ext = w_find_pos(c, WK_DEL, node, score - c.P->gap_extend, diag + 1, false, true);
    if (w_less(open, ext)) { edit = VGK_WFA_DELETION; return ext; }
    edit = VGK_WFA_MATCH; return open;
}
// the same two with the source wavefronts and the leaf's ancestor mask already loaded (they do not change inside next())
VGK_HD WPos w_ins_source(WCtx& c, const WSrc& open_src, const WSrc& ext_src, uint32_t anc, uint32_t leaf, int32_t diag) {
    const WPos open = w_find_in(c, WK_MATCH, open_src, anc, leaf, diag - 1, true, false);
    const WPos ext = w_find_in(c, WK_INS, ext_src, anc, leaf, diag - 1, true, false);
    return w_less(open, ext) ? ext : open;
}
VGK_HD WPos w_del_source(WCtx& c, const WSrc& open_src, const WSrc& ext_src, uint32_t anc, uint32_t leaf, int32_t diag) {
    const WPos open = w_find_in(c, WK_MATCH, open_src, anc, leaf, diag + 1, false, true);
    const WPos ext = w_find_in(c, WK_DEL, ext_src, anc, leaf, diag + 1, false, true);
    return w_less(open, ext) ? ext : open;
}
VGK_HD WPos w_match_predecessor(WCtx& c, uint32_t node, int32_t score, int32_t diag, int& edit) {                             // (:1809-1823)
    const WPos ins = w_find_pos(c, WK_INS, node, score, diag, false, false);
    const WPos del = w_find_pos(c, WK_DEL, node, score, diag, false, false);
    WPos subst = w_find_pos(c, WK_MATCH, node, score - c.P->mismatch, diag, false, false);
    if (!subst.empty) { subst.seq++; subst.off++; }
    if (w_less(ins, del)) {
        if (w_less(del, subst)) { edit = VGK_WFA_MISMATCH; return subst; }
        edit = VGK_WFA_DELETION; return del;
    }
    if (w_less(ins, subst)) { edit = VGK_WFA_MISMATCH; return subst; }
    edit = VGK_WFA_INSERTION; return ins;
}
VGK_HD void w_successor_offset(WCtx& c, WPos& p) {                                                // (:1827-1832)
    if (w_past_end(c, p.cur, p.off)) { w_pop(c, p); p.off = 0; }
    p.off++;
}
VGK_HD void w_predecessor_offset(const WCtx& c, uint32_t& node, uint32_t& off) {                  // (:1835-1842)
    if (off > 0) --off;
    else { node = c.S->nodes[node].parent; off = (c.end[node * c.end_stride] & 0x7fffffffu) - 1; }
}

VGK_HD void w_expand_if_necessary(WCtx& c, const WPos& p) {                                       // (:1992-2008)
    const uint32_t node = p.cur;
    if (c.S->nodes[node].n_children || !w_past_end(c, node, p.off) || c.S->nodes[node].dead_end) return;
    const WState st = { c.S->nodes[node].st_node, c.S->nodes[node].st_lo, c.S->nodes[node].st_hi };
    WState next = { 0, 0, -1 };
    const uint32_t k = w_follow(c.P->index, st, 0, next, 0xffffffffu);
    if (!k) { c.S->nodes[node].dead_end = 1; return; }
    if (c.n_nodes + k > (uint32_t)W_NODES) { c.overflow = true; c.why = 2; return; }
    c.S->nodes[node].first_child = (uint8_t)c.n_nodes; c.S->nodes[node].n_children = (uint8_t)k;
    c.leaves &= ~(1u << node);
    for (uint32_t i = 0; i < k; ++i) {
        if (i) w_follow(c.P->index, st, i, next, i + 1);
        w_node_init(c, c.n_nodes, next, node); ++c.n_nodes;
        if (c.overflow) return;
    }
}

VGK_HD int32_t w_gap_penalty(const WCtx& c, uint32_t length) { return c.P->gap_open + (int32_t)length * c.P->gap_extend; }          // (:1649)

// WFANode::match_forward (:1533-1542) on the bases of the trie node's path, eight per compare; the node grows as the match runs into its end
VGK_HD void w_match_forward(WCtx& c, WPos& p) {
    if (p.seq >= c.L || w_past_end(c, p.cur, p.off)) return;
    uint32_t k = c.S->nodes[p.cur].path_head;
    while (c.S->path_next[k] != W_NIL && c.S->path_start[c.S->path_next[k]] <= p.off) k = c.S->path_next[k];
    for (;;) {
        const int32_t gn = c.S->path_node[k];
        const uint32_t start = c.S->path_start[k], gl = g_len(c.P->index, gn);
        const char* g = c.P->index.seq + g_seq_off(c.P->index, (uint32_t)gn) + (p.off - start);
        const char* r = c.seq + p.seq;
        uint32_t left = start + gl - p.off; if (c.L - p.seq < left) left = c.L - p.seq;
        uint32_t m = 0;
        while (m < left) {
            const uint64_t x = g_load8(g + m) ^ g_load8(r + m);
            if (x) { m += (uint32_t)(__builtin_ctzll(x) >> 3); break; }
            m += 8;
        }
        const bool differs = m < left;
        if (!differs) m = left;
        p.seq += m; p.off += m;
        if (differs || p.seq >= c.L || w_past_end(c, p.cur, p.off) || c.overflow) return;
        k = c.S->path_next[k];
    }
}

VGK_HD void w_candidate(WCtx& c, int32_t score, int32_t diag, uint32_t seq, uint32_t off, uint32_t node) {
    if (score < c.cand_score) { c.cand_score = score; c.cand_diag = diag; c.cand_seq = seq; c.cand_off = off; c.cand_node = node; }
}

VGK_HD void w_extend(WCtx& c, int32_t score) {                                                    // (:1656-1666, :1874-1935)
    const WPScore ps = w_ps(c, score);
    if (!(ps.flags & 1)) return;
    const WSrc here = { score, ps.min_d, ps.max_d };
    for (int32_t diag = ps.min_d; diag <= ps.max_d && !c.overflow; ++diag) {
        const uint32_t leaves = c.leaves;
        for (uint32_t top = 0; top < c.n_nodes && !c.overflow; ++top) {
            if (!(leaves >> top & 1)) continue;
            uint32_t sp = 0;
            c.S->stack_cur[0] = (uint8_t)top; c.S->stack_end[0] = (uint8_t)(top + 1); sp = 1;
            while (sp && !c.overflow) {
                if (c.S->stack_cur[sp - 1] == c.S->stack_end[sp - 1]) { --sp; continue; }
                const uint32_t leaf = c.S->stack_cur[sp - 1]++;
                WPos pos = w_find_in(c, WK_MATCH, here, c.S->nodes[leaf].ancestors, leaf, diag, false, false);
                if (pos.empty) continue;
                for (;;) {
                    const uint32_t off_before = pos.off;
                    w_match_forward(c, pos);
                    const bool at_end = w_past_end(c, pos.cur, pos.off);                            // the node is as long as this position needs from here on
                    const uint32_t target_offset = c.no_to ? W_NO_OFFSET : c.S->nodes[pos.cur].target_offset;
                    const bool may_reach_target = target_offset != W_NO_OFFSET && target_offset >= off_before;      // a set target lies inside the node
                    if ((may_reach_target && pos.off >= target_offset) || (c.no_to && pos.seq >= c.L)) {
                        const uint32_t overshoot = c.no_to ? 0 : pos.off - target_offset;
                        const uint32_t gap_length = (c.L - pos.seq) + overshoot;
                        w_candidate(c, score + (gap_length ? w_gap_penalty(c, gap_length) : 0), diag, pos.seq - overshoot, target_offset, pos.cur);
                    }
                    if (w_distance(pos, diag) > c.max_distance) c.max_distance = w_distance(pos, diag);
                    w_update(c, WK_MATCH, score, diag, pos);
                    if (!at_end) break;
                    w_expand_if_necessary(c, pos);
                    if (c.overflow) break;
                    if (pos.cur == pos.origin) {
                        const WNode& now = c.S->nodes[pos.cur];
                        if (now.n_children) { c.S->stack_cur[sp] = now.first_child; c.S->stack_end[sp] = (uint8_t)(now.first_child + now.n_children); ++sp; }
                        break;
                    }
                    w_pop(c, pos); pos.off = 0;
                }
            }
        }
    }
}

VGK_HD void w_mark(WCtx& c, int32_t score, bool gap) {                                            // possible_scores[...] of next_score
    const uint8_t f = c.S->ps_flags[score];
    if (!(f & 1)) { w_ps_set_range(c, score, 0, 0); c.S->ps_flags[score] = (uint8_t)(1 | (gap ? 2 : 0)); }
    else if (gap && !(f & 2)) c.S->ps_flags[score] = (uint8_t)(f | 2);
}
VGK_HD int32_t w_next_score(WCtx& c, int32_t match_score) {                                       // (:1672-1704)
    w_mark(c, match_score + c.P->mismatch, false);
    if (c.S->ps_flags[match_score] & 2) w_mark(c, match_score + c.P->gap_extend, true);
    w_mark(c, match_score + c.P->gap_open + c.P->gap_extend, true);
    int32_t s = match_score + 1;
    while (!(c.S->ps_flags[s] & 1)) ++s;
    return s;
}
VGK_HD void w_range(const WCtx& c, int32_t& lo, int32_t& hi, int32_t score) {                     // (:1956-1968)
    if (score < 0) return;
    const WPScore p = w_ps(c, score);
    if (!(p.flags & 1)) return;
    if (p.min_d < lo) lo = p.min_d;
    if (p.max_d > hi) hi = p.max_d;
}

VGK_HD void w_next(WCtx& c, int32_t score) {                                                      // (:1709-1786)
    int32_t lo = 32767, hi = -32768;                                                              // get_diagonals (:1977-1988)
    w_range(c, lo, hi, score - c.P->mismatch);
    w_range(c, lo, hi, score - c.P->gap_open - c.P->gap_extend);
    w_range(c, lo, hi, score - c.P->gap_extend);
    if (lo <= hi) { --lo; ++hi; }
    int32_t alo = 32767, ahi = -32768;
    const WSrc src_mismatch = w_src(c, score - c.P->mismatch), src_open = w_src(c, score - c.P->gap_open - c.P->gap_extend), src_extend = w_src(c, score - c.P->gap_extend);
    for (int32_t diag = lo; diag <= hi && !c.overflow; ++diag) {
        const uint32_t leaves = c.leaves;
        for (uint32_t leaf = 0; leaf < (uint32_t)W_NODES && (leaves >> leaf) && !c.overflow; ++leaf) {
            if (!(leaves >> leaf & 1)) continue;
            const uint32_t anc = c.S->nodes[leaf].ancestors;
            WPos ins = w_ins_source(c, src_open, src_extend, anc, leaf, diag);
            if (!ins.empty) {
                ins.seq++;
                if (w_distance(ins, diag) >= c.min_distance) { w_update(c, WK_INS, score, diag, ins); if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
            }
            WPos del = w_del_source(c, src_open, src_extend, anc, leaf, diag);
            if (!del.empty) {
                w_successor_offset(c, del);
                if (w_distance(del, diag) >= c.min_distance) { w_update(c, WK_DEL, score, diag, del); if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
                w_expand_if_necessary(c, del);
            }
            WPos subst = w_find_in(c, WK_MATCH, src_mismatch, anc, leaf, diag, true, true);
            if (!subst.empty) { subst.seq++; w_successor_offset(c, subst); w_expand_if_necessary(c, subst); }
            if (w_less(subst, ins)) subst = ins;
            if (w_less(subst, del)) subst = del;
            if (!subst.empty) {
                if (!c.no_to) w_past_end(c, subst.cur, subst.off);                                   // a target right at the materialised end shows itself
                if (!c.no_to && subst.off == c.S->nodes[subst.cur].target_offset) {
                    const uint32_t gap_length = c.L - subst.seq;
                    w_candidate(c, score + (gap_length ? w_gap_penalty(c, gap_length) : 0), diag, subst.seq, subst.off, subst.cur);
                }
                if (w_distance(subst, diag) >= c.min_distance) { w_update(c, WK_MATCH, score, diag, subst); if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
            }
        }
    }
    if (c.S->ps_flags[score] & 1) w_ps_set_range(c, score, alo, ahi);
}

VGK_HD int32_t w_alignment_score(const WCtx& c, int32_t score, int32_t diag, uint32_t seq, uint32_t final_insertion) {              // (:1380-1387)
    const int32_t target_offset = (int32_t)seq - diag;
    return (c.P->match * ((int32_t)(seq + final_insertion) + target_offset) - score) / 2;
}

// WFATree::trim (:1849-1868).  Among equally good points the reference keeps the first in hash-map order; here the one with
// the smallest (trie node, penalty, diagonal) — the order of the packed keys.
VGK_HD void w_trim(WCtx& c) {
    c.cand_score = 0; c.cand_diag = 0; c.cand_seq = 0; c.cand_off = 0; c.cand_node = 0;
    int32_t best = 0; uint32_t best_order = 0xffffffffu;
    for (uint32_t i = 0; i < c.n_points; ++i) {
        const uint64_t s = c.S->slot[c.S->log[i]];
        const uint32_t key = (uint32_t)(s >> 32) - 1;
        if (((key >> 5) & 3u) != (uint32_t)WK_MATCH) continue;
        const uint32_t node = key & 31u; const int32_t score = (int32_t)((key >> 7) & 1023u), diag = (int32_t)((key >> 17) & 1023u) - 512;
        const uint32_t seq = (uint32_t)(s >> 16) & 0xffffu, off = (uint32_t)s & 0xffffu;
        const int32_t as = w_alignment_score(c, score, diag, seq, 0);
        const uint32_t order = (node << 20) | ((uint32_t)score << 10) | (uint32_t)(diag + 512);
        if (as > best || (as == best && best_order != 0xffffffffu && order < best_order)) {
            best = as; best_order = order;
            c.cand_score = score; c.cand_diag = diag; c.cand_seq = seq; c.cand_off = off; c.cand_node = node;
        }
    }
}

VGK_HD void w_append_edit(WCtx& c, uint32_t& n_edits, int edit, uint32_t length) {                 // WFAAlignment::append (:850-859)
    if (!length) return;
    if (n_edits && (c.S->edits[n_edits - 1] & 3u) == (uint32_t)edit) { c.S->edits[n_edits - 1] += length << 2; return; }
    if (n_edits >= (uint32_t)W_EDITS) { c.overflow = true; c.why = 4; return; }
    c.S->edits[n_edits++] = (length << 2) | (uint32_t)edit;
}

// One problem (WFAExtender::connect :2052-2235; suffix :2237-2246; the strand flip of prefix :2248-2263 happens as the
// result is written out: the backtrace yields the edits last-to-first, which is the flipped order).
VGK_HD void wfa_extend_one(const WfaParams& P, uint32_t i, WScratch& S, uint32_t* end, uint32_t end_stride) {
    const WProb pb = P.probs[i];
    vgk_wfa_result out; out.status = pb.status; out.ok = 0; out.score = 0; out.node_offset = 0; out.seq_offset = 0; out.length = 0;
    out.path_begin = 0; out.path_len = 0; out.edit_begin = 0; out.n_edits = 0;
    if (pb.status != VGK_OK || pb.from_node >= P.index.n_oriented) { P.results[i] = out; return; }     // !has_node(id(from)) (:2059)
    WCtx c;
    c.P = &P; c.S = &S; c.end = end; c.end_stride = end_stride; c.seq = P.seqs + pb.seq_off; c.L = pb.seq_len;
    c.no_to = pb.to_node == VGK_WFA_NO_NODE; c.to_node = (int32_t)pb.to_node; c.to_off = pb.to_off;
    c.n_nodes = 0; c.n_path = 0; c.n_points = 0; c.leaves = 0; c.overflow = false; c.why = 0;
    c.max_points = c.no_to ? P.max_points_tail : P.max_points;
    const bool hands_over = P.handed_over && P.hand_over_points < c.max_points;
    if (hands_over) c.max_points = P.hand_over_points;
    { uint32_t slots = 64; while (slots < (uint32_t)W_SLOTS && slots < 4u * c.max_points) slots <<= 1; c.slot_mask = slots - 1u; }
    c.cand_score = 0x7fffffff; c.cand_diag = 0; c.cand_seq = 0; c.cand_off = 0; c.cand_node = 0;
    c.max_distance = 0; c.min_distance = 0;
    const int32_t top_score = pb.score_bound + P.gap_open + P.gap_extend + P.mismatch;               // the host keeps this below W_SCORES
    for (int32_t s = 0; s <= top_score && s < W_SCORES; s += 8) { uint64_t z = 0; __builtin_memcpy(S.ps_flags + s, &z, 8); }   // W_SCORES is a multiple of 8
    const WState root = { (int32_t)pb.from_node, 0, (int32_t)g_rec(P.index, pb.from_node)[0] - 1 };
    w_node_init(c, 0, root, 0); c.n_nodes = 1;
    w_store(c, 0, WK_MATCH, 0, 0, 0, pb.from_off + 1);
    w_mark(c, 0, false);

    int32_t score = 0;
    while (!c.overflow) {
        w_extend(c, score);
        if (pb.distance_band < c.max_distance) c.min_distance = c.max_distance - pb.distance_band;
        if (c.cand_score <= score) break;
        score = w_next_score(c, score);
        if (score > pb.score_bound) break;
        w_next(c, score);
    }

    bool ok = !c.overflow;
    uint32_t unaligned_tail = c.L - c.cand_seq;
    if (ok && c.cand_score > pb.score_bound) {
        unaligned_tail = 0;
        if (c.no_to) w_trim(c); else ok = false;
    }
    uint32_t n_edits = 0; bool lost = false;
    if (ok) {
        out.ok = 1; out.node_offset = pb.from_off + 1;
        out.length = c.cand_seq + unaligned_tail;
        out.score = w_alignment_score(c, c.cand_score, c.cand_diag, c.cand_seq, unaligned_tail);
        int32_t p_score = c.cand_score, p_diag = c.cand_diag; uint32_t p_seq = c.cand_seq, p_off = c.cand_off, node = c.cand_node;
        if (unaligned_tail > 0) { w_append_edit(c, n_edits, VGK_WFA_INSERTION, c.L - c.cand_seq); p_score -= w_gap_penalty(c, unaligned_tail); }
        int edit = VGK_WFA_MATCH;
        while ((p_seq > 0 || p_diag != 0) && !c.overflow && !lost) {                                 // (:2155-2202)
            int pe; WPos pred;
            switch (edit) {
            case VGK_WFA_MATCH:
                pred = w_match_predecessor(c, node, p_score, p_diag, pe);
                if (pred.empty && (p_score != 0 || p_diag != 0)) { lost = true; break; }
                w_append_edit(c, n_edits, VGK_WFA_MATCH, p_seq - pred.seq);
                p_seq = pred.seq; p_off = pred.off;
                if (!pred.empty) node = pred.cur;
                edit = pe; break;
            case VGK_WFA_MISMATCH:
                w_append_edit(c, n_edits, VGK_WFA_MISMATCH, 1);
                p_seq--; w_predecessor_offset(c, node, p_off);
                p_score -= P.mismatch; edit = VGK_WFA_MATCH; break;
            case VGK_WFA_INSERTION:
                pred = w_ins_predecessor(c, node, p_score, p_diag, pe);
                if (pred.empty) { lost = true; break; }
                w_append_edit(c, n_edits, VGK_WFA_INSERTION, 1);
                p_seq--;
                p_score -= pe == VGK_WFA_INSERTION ? P.gap_extend : P.gap_open + P.gap_extend;
                p_diag--; edit = pe; break;
            default:
                pred = w_del_predecessor(c, node, p_score, p_diag, pe);
                if (pred.empty) { lost = true; break; }
                w_append_edit(c, n_edits, VGK_WFA_DELETION, 1);
                w_predecessor_offset(c, node, p_off);
                p_score -= pe == VGK_WFA_DELETION ? P.gap_extend : P.gap_open + P.gap_extend;
                p_diag++; edit = pe; break;
            }
        }
        ok = !c.overflow && !lost;
    }
    if (lost) {
        // A candidate found by next() is recorded before the distance check (:1761-1776): behind min_distance neither it nor its
        // gap point is stored, and the reference's backtrace then leaves the stored wavefronts and does not terminate.
        out.status = VGK_ENOBAND; out.ok = 0; out.score = 0; out.node_offset = 0; out.length = 0;
    }
    if (ok) {
        // the path: the trie nodes from the root to the candidate; minus an exhausted first node (:2208-2211) and the trailing
        // nodes no edit reaches (:2217-2229)
        uint32_t n_chain = 0;
        for (uint32_t x = c.cand_node;; x = S.nodes[x].parent) { S.chain[n_chain++] = (uint8_t)x; if (x == 0) break; }
        uint32_t ref_len = 0;
        for (uint32_t e = 0; e < n_edits; ++e) if ((S.edits[e] & 3u) != (uint32_t)VGK_WFA_INSERTION) ref_len += S.edits[e] >> 2;
        const int32_t first_node = S.path_node[S.nodes[0].path_head];
        const uint32_t first_len = g_len(P.index, first_node);
        const bool drop_first = out.node_offset >= first_len;
        if (drop_first) out.node_offset = 0;
        const uint32_t used = out.node_offset + ref_len;              // end of the alignment, from the start of the first kept node
        uint32_t kept = 0, at = 0, last_start = 0, last_len = 0;       // `at` = start of the current node in the same coordinates
        for (uint32_t k = n_chain; k-- > 0;) {
            const WNode& n = S.nodes[S.chain[k]];
            for (uint32_t j = n.path_head; j != W_NIL; j = S.path_next[j]) {
                const uint32_t gl = g_len(P.index, S.path_node[j]);
                if (k == n_chain - 1 && j == n.path_head && drop_first) continue;
                if (kept == 0 || at < used) { ++kept; last_start = at; last_len = gl; }
                at += gl;
            }
        }
        if (kept == 1 && used == out.node_offset) kept = 0;
        const unsigned long long p0 = g_bump(P.counters + 0, kept), e0 = g_bump(P.counters + 1, n_edits);
        if (p0 + kept > P.caps[0] || e0 + n_edits > P.caps[1]) { out.status = VGK_EOPS; out.ok = 0; }
        else {
            const bool flip = pb.mode == VGK_WFA_PREFIX;
            uint32_t w = 0;
            for (uint32_t k = n_chain; k-- > 0 && w < kept;) {
                const WNode& n = S.nodes[S.chain[k]];
                for (uint32_t j = n.path_head; j != W_NIL && w < kept; j = S.path_next[j]) {
                    if (k == n_chain - 1 && j == n.path_head && drop_first) continue;
                    const uint32_t o = (uint32_t)S.path_node[j];
                    P.paths[p0 + (flip ? kept - 1 - w : w)] = flip ? (o ^ 1u) : o;
                    ++w;
                }
            }
            for (uint32_t e = 0; e < n_edits; ++e) P.edits[e0 + e] = S.edits[flip ? e : n_edits - 1 - e];
            out.path_begin = (uint32_t)p0; out.path_len = kept; out.edit_begin = (uint32_t)e0; out.n_edits = n_edits;
            if (pb.mode != VGK_WFA_CONNECT && n_edits && out.length == c.L) {                        // (:2240-2243, :2258-2260)
                const uint32_t last = S.edits[0] & 3u;                                               // the alignment's last edit
                if (last == (uint32_t)VGK_WFA_MATCH || last == (uint32_t)VGK_WFA_MISMATCH) out.score += P.bonus;
            }
            if (flip) {                                                                              // WFAAlignment::flip (:834-848)
                out.seq_offset = c.L - out.seq_offset - out.length;
                if (kept) out.node_offset = last_len - (used - last_start);
            }
        }
    } else if (c.overflow) { out.status = VGK_ETOOBIG; out.score = c.why; }
    for (uint32_t k = 0; k < c.n_points; ++k) S.slot[S.log[k]] = 0;                                  // leave the table clean
    if (c.overflow && c.why == 1 && hands_over) { P.handed_over[g_bump(P.n_handed_over, 1)] = i; return; }     // the wavefront kernel answers this one
    P.results[i] = out;
}
// one resident thread: problems are handed out one at a time.  `end` = W_NODES words for this thread's node ends, `end_stride`
// words apart (LDS in the HIP kernel, lane-interleaved; the slab's own array otherwise)
VGK_HD void wfa_thread(const WfaParams& P, uint32_t t, uint32_t* end, uint32_t end_stride) {
    for (;;) {
        const unsigned long long k = g_bump(P.counters + 2, 1);
        if (k >= P.n) break;
        wfa_extend_one(P, P.order[k], P.scratch[t], end ? end : P.scratch[t].node_end, end ? end_stride : 1u);
    }
}

}  // namespace vgk
