// tail_device.hpp — tail forests: the haplotype-consistent subgraphs giraffe aligns read tails to, walked and turned into a
// resident graph on the device (replaces MinimizerMapper::get_tail_forest, src/minimizer_mapper.cpp:5745-5860, and its
// dfs_gbwt, :5909-6013, for a batch of tails; the TreeSubgraph -> create_gssw_graph conversion that follows them in the
// reference, src/aligner.cpp:30-85, is the second half of this file).  DESIGN.md §17.
//
// Two kinds of lane code: (1) one lane per tail runs the depth-first walk over the haplotype index of gapless_device.hpp with
// an explicit stack in a per-lane slab — twice, first to size the forest, then to write it where a prefix sum over the sizes
// puts it; (2) one lane per tree node derives the tables of the window packer (gssw_pack_device.hpp: columns, column-info
// bytes, predecessor CSR, stored-node slots) for the whole forest as ONE graph, every tree a run of consecutive nodes.
//
// The same code runs on the CPU under tests/emu (test infrastructure only).
#pragma once
#include <stdint.h>
#include "gapless_device.hpp"
#include "gssw_device.hpp"
#include "../../include/vgk.h"

namespace vgk {

constexpr int T_STACK = 512;          // frames of a walk's stack: nodes on the current path + their untaken siblings
struct TFrame { int32_t node, lo, hi; uint32_t used; int32_t self; };      // self: T_FRESH before the first visit, then the tree node it became (-1: the skipped root)
constexpr int32_t T_FRESH = -2;
struct TScratch { TFrame stack[T_STACK]; int32_t from[T_STACK]; };          // from[k]: the tree node that pushed frame k (its parent in the tree; -1 = none)

struct TailParams {
    GIndex index;
    const vgk_tail_problem* probs; uint32_t n;
    vgk_tail_result* results;
    uint32_t* counts;                 // pass 1 out: tree nodes per problem ([n + 1], the last one 0) — the scan's input
    const uint32_t* first;            // pass 2 in: exclusive prefix sums of counts
    int32_t*  parent;                 // pass 2 out, per tree node: parent as an index in the forest, or -1
    uint32_t* node;                   //   the oriented node of the index
    uint32_t* len;                    //   its length in the forest graph
    uint32_t* trim;                   //   bases cut off its start (the root's cut)
    uint32_t* owner;                  //   (nullable) the problem it belongs to
    TScratch* scratch;                // one per resident lane
    int pass;
};

// the visits [lo, hi] of `rec` that leave through edge e: where they land in the successor's record
VGK_HD bool t_follow(const uint32_t* rec, int32_t lo, int32_t hi, uint32_t e, bool few, const GCounts& cn, int32_t& nlo, int32_t& nhi) {
    int32_t before = 0, inside = 0;
    if (few) { before = (int32_t)g_count_of(cn.before, e); inside = (int32_t)g_count_of(cn.inside, e); }
    else {
        const uint32_t* body = g_visits(rec);
        if (g_rle(rec)) {
            int32_t pos = 0;
            for (uint32_t k = 0; pos <= hi; ++k) {
                const uint32_t run = body[k]; const int32_t end = pos + (int32_t)(run >> 8);
                if ((run & 0xffu) == e) {
                    const int32_t b = (end < lo ? end : lo) - pos, last = end - 1 < hi ? end - 1 : hi, first = pos > lo ? pos : lo;
                    if (b > 0) before += b;
                    if (last >= first) inside += last - first + 1;
                }
                pos = end;
            }
        } else for (int32_t i = 0; i <= hi; ++i) if (g_body(body, (uint32_t)i) == e) { if (i < lo) ++before; else ++inside; }
    }
    if (inside <= 0) return false;
    nlo = (int32_t)ge_base(rec, e) + before; nhi = nlo + inside - 1;
    return true;
}

// One tail: dfs_gbwt (:5909-6013) with get_tail_forest's enter / exit handlers (:5823-5851) folded in.  The reference keeps a stack
// of parents beside the walk's stack; here every frame remembers which tree node pushed it.
VGK_HD void tail_walk_one(const TailParams& P, uint32_t i, TScratch& S) {
    const GIndex& h = P.index;
    const vgk_tail_problem pb = P.probs[i];
    if (P.pass == 2 && P.results[i].status != VGK_OK) return;                // declined in the sizing pass: it has no room in the forest, and its result stands
    vgk_tail_result out; out.status = VGK_OK; out.first_node = P.pass == 2 ? P.first[i] : 0u; out.n_nodes = 0; out.n_trees = 0; out.root_trim = 0; out.bases = 0;
    auto done = [&]() { if (P.pass == 1) P.counts[i] = out.status == VGK_OK ? out.n_nodes : 0u; if (out.status != VGK_OK) { out.n_nodes = 0; out.n_trees = 0; out.bases = 0; } P.results[i] = out; };
    if (pb.node >= h.n_oriented) { out.status = VGK_EINVAL; done(); return; }
    const uint32_t root_len = g_len(h, (int32_t)pb.node);
    if (pb.offset > root_len || pb.hi >= (int32_t)g_rec(h, pb.node)[0] || pb.lo < 0) { out.status = VGK_EINVAL; done(); return; }
    if (pb.lo > pb.hi) { done(); return; }                                    // no haplotype visits the first node (:5912-5915)
    const uint32_t remaining_root = root_len - pb.offset;                     // the cut is between bases: everything behind it (:5925)
    out.root_trim = remaining_root ? pb.offset : 0u;
    uint32_t sp = 0, count = 0;
    S.stack[0].node = (int32_t)pb.node; S.stack[0].lo = pb.lo; S.stack[0].hi = pb.hi; S.stack[0].used = 0; S.stack[0].self = T_FRESH; S.from[0] = -1; sp = 1;
    while (sp) {
        TFrame& f = S.stack[sp - 1];
        const bool is_root = sp == 1, hidden = is_root && remaining_root == 0;
        if (f.self == T_FRESH) {
            const uint32_t node_length = is_root ? remaining_root : g_len(h, f.node);
            if (!hidden) {                                                     // enter (:5823-5846)
                const int32_t par = S.from[sp - 1];
                f.self = (int32_t)count;
                if (par < 0) ++out.n_trees;                                    // nothing above it: the root of a tree
                if (P.pass == 2) {
                    const uint32_t at = out.first_node + count;
                    P.parent[at] = par < 0 ? -1 : (int32_t)(out.first_node + (uint32_t)par);
                    P.node[at] = (uint32_t)f.node; P.len[at] = node_length; P.trim[at] = is_root ? pb.offset : 0u;
                    if (P.owner) P.owner[at] = i;
                }
                ++count; out.bases += node_length;
            } else f.self = -1;
            f.used += node_length;
            if (f.used < pb.walk_distance) {                                   // follow_paths: the non-empty one-node extensions, in edge order (:5975-5983)
                const uint32_t* rec = g_rec(h, (uint32_t)f.node);
                const uint32_t ne = g_ne(rec);
                const bool few = ne <= 4;
                const GCounts cn = few ? g_counts(rec, f.lo, f.hi) : GCounts{0, 0};
                const int32_t self = f.self; const uint32_t used = f.used; const int32_t lo = f.lo, hi = f.hi;
                for (uint32_t e = 0; e < ne; ++e) {
                    const int32_t to = ge_to(rec, e); if (to < 0) continue;
                    int32_t nlo, nhi;
                    if (!t_follow(rec, lo, hi, e, few, cn, nlo, nhi)) continue;
                    if (sp >= (uint32_t)T_STACK) { out.status = VGK_ETOOBIG; done(); return; }
                    TFrame& c = S.stack[sp]; c.node = to; c.lo = nlo; c.hi = nhi; c.used = used; c.self = T_FRESH; S.from[sp] = self; ++sp;
                }
                continue;                                                      // the new top of the stack first; back here for the second visit
            }
        }
        --sp;                                                                  // second visit, or nothing to expand: exit and pop
    }
    out.n_nodes = count;
    done();
}

// ---- the forest as one resident graph (what vgk_graph_create computes on the host for a caller's graph) ---------------------------
struct ForestParams {
    GIndex index;
    uint32_t n_nodes;                 // tree nodes of the whole forest
    const int32_t* parent; const uint32_t* node; const uint32_t* len; const uint32_t* trim;
    uint32_t* has_pred;               // [n + 1] 1 when the node has a parent (the scan's input; [n] = 0)
    uint32_t* store;                  // [n + 1] 1 when a successor seeds its first column from this node's saved last column
    uint32_t* slow;                   // [n]
    // after the scans
    const uint32_t* col; const uint32_t* pred_off; uint32_t* pred_idx; uint8_t* info;
};
VGK_HD int t_ref_code(char ch) { switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; } }      // after nonATGCNtoN (src/aligner.cpp:39)
// stage 1, per node: the flags of window_api.cpp's vgk_graph_create — slow = the first column is seeded from saved last columns
// (anything but the plain chain link to the node before), store = some successor does that from this node
VGK_HD void forest_flags_one(const ForestParams& P, uint32_t v) {
    const int32_t par = P.parent[v];
    const bool chain = par >= 0 && (uint32_t)par + 1u == v;
    const bool slow = v > 0 && !chain;
    P.has_pred[v] = par >= 0 ? 1u : 0u;
    P.slow[v] = slow ? 1u : 0u;
    if (slow && par >= 0) P.store[par] = 1u;                                   // every writer writes 1
}
// stage 2, per node: its predecessor entry and the column-info bytes of its bases
VGK_HD void forest_emit_one(const ForestParams& P, uint32_t v) {
    const int32_t par = P.parent[v];
    if (par >= 0) P.pred_idx[P.pred_off[v]] = (uint32_t)par;
    const uint32_t len = P.len[v];
    const char* sq = P.index.seq + g_seq_off(P.index, P.node[v]) + P.trim[v];
    uint8_t* o = P.info + P.col[v];
    for (uint32_t k = 0; k < len; ++k) o[k] = (uint8_t)t_ref_code(sq[k]);
    o[0] |= (uint8_t)(CI_NODE_START | (P.slow[v] ? CI_SEED_SLOW : 0));
    if (P.store[v]) o[len - 1] |= (uint8_t)CI_STORE_END;
}

// ---- the tails of a batch of extension sets, on the device (vgk_tail_stage) -------------------------------------------------------
// What vg_amd/host/tail_stage.cpp does on host threads — which tails exist, their cuts and walk distances from the extensions' search
// states, the tails' bases, one window per tree, best tree per tail, totals (src/minimizer_mapper.cpp:5480-5535) — as lane code over the
// extension sets vgk_gapless_extend left in HBM.  Order of the tails: the right tails by extension, then the left tails, as there.
struct TMeta { uint32_t ext, read, begin, end, left, gap; };
struct TStageParams {
    GIndex index;
    uint32_t n_reads, n_ext;
    const GProb* probs; const char* reads;                        // the extension stage's inputs (masked reads, padded)
    const vgk_gapless_result* res; const vgk_extension* ext; const uint32_t* nodes;      // its sets, in problem order
    int32_t match, gap_open, gap_extend, bonus;
    const uint32_t* read_of;                                      // [n_ext]: the read an extension belongs to (the extension call's gather kernel wrote it)
    uint32_t* cnt_r; uint32_t* cnt_l; const uint32_t* off_r; const uint32_t* off_l; uint32_t total_r;      // [n_ext + 1] each
    vgk_tail_problem* problems; TMeta* meta; uint32_t n_tails;
    uint32_t* tail_len; const uint32_t* seq_off; char* seq;       // [n_tails + 1]; the tails' bases behind each other
    // trees -> windows
    const vgk_tail_result* tres; const int32_t* parent; const uint32_t* owner; uint32_t n_nodes;
    uint32_t* is_root; const uint32_t* root_off; uint32_t* root_pos; uint32_t n_trees;      // [n_nodes + 1]
    vgk_window_problem* windows; uint32_t* win_owner;
    // scores
    const vgk_result* wres; int32_t* tail_score; int32_t* ext_total; int32_t* read_score; unsigned long long* failed;
    // the winning tree's alignment per tail (vgk_tail_stage_aligned)
    unsigned long long* tail_best;                                 // [n_tails]: (score << 32) | (0xffffffff - window): the best tree, the first among equals; 0 = the soft clip
    const uint32_t* forest_node;                                   // [n_nodes]: oriented graph node of every tree node (TreeSubgraph::translate_down)
    const vgk_op* wops;                                            // the window batch's op slots (vgk_result::ops_begin indexes it)
    uint32_t* ops_cnt; const uint32_t* ops_off;                    // [n_tails + 1]
    vgk_tail_alignment* aligned; vgk_op* out_ops;
};
VGK_HD int64_t t_longest_gap(const TStageParams& P, int64_t read_length, int64_t read_pos) {     // EditAlignmentScorer::longest_detectable_gap (src/alignment_scorer.cpp:264-271)
    const int64_t overhang = read_pos < read_length - read_pos ? read_pos : read_length - read_pos;
    const int64_t gap = (P.match * overhang + P.bonus - P.gap_open) / P.gap_extend + 1;
    return (gap >= 0 && overhang > 0) ? gap : 0;
}
VGK_HD char t_comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; } }
VGK_HD bool tstage_open(const TStageParams& P, uint32_t e, const vgk_extension& x) {
    const vgk_gapless_result r = P.res[P.read_of[e]];
    return r.status == VGK_OK && !r.full_length && x.path_len;      // full-length sets are scored as they are (:5440)
}
// stage 1, per extension: does it have a right / a left tail (stage 0, the extension -> read table, is the extension call's: read_of)
VGK_HD void tstage_count_one(const TStageParams& P, uint32_t e) {
    const vgk_extension x = P.ext[e];
    const bool open = tstage_open(P, e, x);
    P.cnt_r[e] = open && !x.right_full ? 1u : 0u; P.cnt_l[e] = open && !x.left_full ? 1u : 0u;
    P.ext_total[e] = x.score;                                      // the total starts as the extension's own score
}
// stage 2, per extension: its tail problems
VGK_HD void tstage_tails_one(const TStageParams& P, uint32_t e) {
    const vgk_extension x = P.ext[e];
    if (!tstage_open(P, e, x)) return;
    const uint32_t r = P.read_of[e];
    const int64_t L = (int64_t)P.probs[r].read_len;
    if (!x.right_full) {                                           // look right from the end, forward state (:5768-5775)
        uint64_t before_last = 0;
        for (uint32_t k = 0; k + 1 < x.path_len; ++k) before_last += g_len(P.index, (int32_t)P.nodes[x.path_begin + k]);
        const int64_t tail = L - x.read_end, gap = t_longest_gap(P, L, tail);
        const uint32_t at = P.off_r[e];
        vgk_tail_problem p; p.node = x.state[0]; p.lo = (int32_t)x.state[1]; p.hi = (int32_t)x.state[2];
        p.offset = (uint32_t)(x.offset + (x.read_end - x.read_begin) - before_last); p.walk_distance = (uint32_t)(tail + gap);
        P.problems[at] = p;
        TMeta m; m.ext = e; m.read = r; m.begin = x.read_end; m.end = (uint32_t)L; m.left = 0; m.gap = (uint32_t)gap;
        P.meta[at] = m; P.tail_len[at] = m.end - m.begin;
    }
    if (!x.left_full) {                                            // look the other way from the start, backward state (:5756-5766)
        const uint32_t first = P.nodes[x.path_begin] ^ 1u;
        const int64_t tail = x.read_begin, gap = t_longest_gap(P, L, tail);
        const uint32_t at = P.total_r + P.off_l[e];
        vgk_tail_problem p; p.node = x.state[3]; p.lo = (int32_t)x.state[4]; p.hi = (int32_t)x.state[5];
        p.offset = g_len(P.index, (int32_t)first) - x.offset; p.walk_distance = (uint32_t)(tail + gap);
        P.problems[at] = p;
        TMeta m; m.ext = e; m.read = r; m.begin = 0; m.end = x.read_begin; m.left = 1; m.gap = (uint32_t)gap;
        P.meta[at] = m; P.tail_len[at] = m.end - m.begin;
    }
}
// stage 3, per tail: its bases — a right tail as it lies in the read, a left tail reverse-complemented (:5660)
VGK_HD void tstage_bases_one(const TStageParams& P, uint32_t t) {
    const TMeta m = P.meta[t];
    const char* rd = P.reads + P.probs[m.read].read_off; char* dst = P.seq + P.seq_off[t];
    const uint32_t len = m.end - m.begin;
    if (!m.left) for (uint32_t k = 0; k < len; ++k) dst[k] = rd[m.begin + k];
    else for (uint32_t k = 0; k < len; ++k) dst[k] = t_comp(rd[m.end - 1 - k]);
    P.tail_score[t] = 0;
}
// stage 4, per tree node: is it the root of a tree; stage 5, per root: where; stage 6, per tree: its window problem
VGK_HD void tstage_root_flag_one(const TStageParams& P, uint32_t v) { P.is_root[v] = P.parent[v] < 0 ? 1u : 0u; }
VGK_HD void tstage_root_pos_one(const TStageParams& P, uint32_t v) { if (P.parent[v] < 0) P.root_pos[P.root_off[v]] = v; }
VGK_HD void tstage_window_one(const TStageParams& P, uint32_t w) {
    const uint32_t v = P.root_pos[w], t = P.owner[v];
    const uint32_t end = P.tres[t].first_node + P.tres[t].n_nodes;
    const uint32_t nxt = w + 1 < P.n_trees ? P.root_pos[w + 1] : P.n_nodes;
    vgk_window_problem q; q.read_off = P.seq_off[t]; q.read_len = P.seq_off[t + 1] - P.seq_off[t]; q.flags = VGK_XDROP_PINNED | VGK_GSSW_TRACEBACK;
    q.first_node = v; q.n_nodes = (nxt < end ? nxt : end) - v; q.max_gap_length = P.meta[t].gap; q.reserved = 0;
    P.windows[w] = q; P.win_owner[w] = t;
}
// stage 7, per window: the best tree of a tail (nothing aligned = the soft clip, 0; :5632-5648); stage 8, per tail: into its extension's
// total; stage 9, per read: the best total
#if defined(__HIP_DEVICE_COMPILE__)
VGK_HD void t_atomic_max(int32_t* p, int32_t v) { atomicMax(p, v); }
VGK_HD void t_atomic_max(unsigned long long* p, unsigned long long v) { atomicMax(p, v); }
VGK_HD void t_atomic_add(int32_t* p, int32_t v) { atomicAdd(p, v); }
#else
VGK_HD void t_atomic_max(int32_t* p, int32_t v) { if (v > *p) *p = v; }
VGK_HD void t_atomic_max(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
VGK_HD void t_atomic_add(int32_t* p, int32_t v) { *p += v; }
#endif
VGK_HD void tstage_best_one(const TStageParams& P, uint32_t w) {
    const vgk_result r = P.wres[w];
    if (r.status != VGK_OK) { g_bump(P.failed, 1); return; }
    t_atomic_max(P.tail_score + P.win_owner[w], r.score);
    // which tree it was: a later tree replaces the best so far only with a strictly better score (equal scores: the reference asks
    // deterministic_beats, :5715 — a hash of the alignments; here the first tree), and the soft clip is kept unless a tree scores > 0
    if (P.tail_best && r.score > 0) t_atomic_max(P.tail_best + P.win_owner[w], ((unsigned long long)(uint32_t)r.score << 32) | (0xffffffffu - w));
}
// stages 10, 11 (vgk_tail_stage_aligned), per tail: the size of the winning alignment, then the alignment itself with its nodes
// translated from tree nodes to the graph's oriented nodes
VGK_HD bool tstage_winner(const TStageParams& P, uint32_t t, uint32_t& w) {
    const unsigned long long b = P.tail_best[t];
    if (!b) return false;
    w = 0xffffffffu - (uint32_t)(b & 0xffffffffull);
    return true;
}
VGK_HD void tstage_ops_count_one(const TStageParams& P, uint32_t t) {
    uint32_t w;
    P.ops_cnt[t] = tstage_winner(P, t, w) ? P.wres[w].n_ops : 0u;
}
VGK_HD void tstage_ops_copy_one(const TStageParams& P, uint32_t t) {
    const TMeta m = P.meta[t];
    vgk_tail_alignment a;
    a.ext = m.ext; a.left = m.left; a.read_begin = m.begin; a.read_end = m.end;
    a.score = 0; a.status = P.tres[t].status; a.ops_begin = P.ops_off[t]; a.n_ops = 0; a.first_offset = 0; a.n_trees = P.tres[t].n_trees;
    uint32_t w;
    if (tstage_winner(P, t, w)) {
        const vgk_result r = P.wres[w];
        const vgk_window_problem q = P.windows[w];
        a.score = r.score; a.n_ops = r.n_ops;
        vgk_op* dst = P.out_ops + P.ops_off[t];
        for (uint32_t k = 0; k < r.n_ops; ++k) {
            vgk_op o = P.wops[r.ops_begin + k];
            const uint32_t v = q.first_node + o.node;
            if (k == 0) a.first_offset = (uint32_t)r.first_offset + (P.parent[v] < 0 ? P.tres[t].root_trim : 0u);
            o.node = P.forest_node[v];
            dst[k] = o;
        }
    }
    P.aligned[t] = a;
}
VGK_HD void tstage_total_one(const TStageParams& P, uint32_t t) {
    if (P.tres[t].status != VGK_OK) g_bump(P.failed, 1);
    t_atomic_add(P.ext_total + P.meta[t].ext, P.tail_score[t]);
}
VGK_HD void tstage_read_one(const TStageParams& P, uint32_t i) {
    const vgk_gapless_result r = P.res[i];
    int32_t best = 0;
    for (uint32_t k = 0; k < r.n_ext; ++k) { const int32_t v = P.ext_total[r.ext_begin + k]; best = v > best ? v : best; }
    P.read_score[i] = best;
}
enum { TS_COUNT = 1, TS_TAILS, TS_BASES, TS_ROOT_FLAG, TS_ROOT_POS, TS_WINDOW, TS_BEST, TS_TOTAL, TS_READ, TS_OPS_COUNT, TS_OPS_COPY };
VGK_HD uint32_t tstage_items(const TStageParams& P, int what) {
    switch (what) { case TS_READ: return P.n_reads; case TS_COUNT: case TS_TAILS: return P.n_ext; case TS_BASES: case TS_TOTAL: case TS_OPS_COUNT: case TS_OPS_COPY: return P.n_tails;
                    case TS_ROOT_FLAG: case TS_ROOT_POS: return P.n_nodes; default: return P.n_trees; }
}
VGK_HD void tstage_one(const TStageParams& P, int what, uint32_t i) {
    switch (what) {
        case TS_COUNT: tstage_count_one(P, i); break; case TS_TAILS: tstage_tails_one(P, i); break;
        case TS_BASES: tstage_bases_one(P, i); break; case TS_ROOT_FLAG: tstage_root_flag_one(P, i); break; case TS_ROOT_POS: tstage_root_pos_one(P, i); break;
        case TS_WINDOW: tstage_window_one(P, i); break; case TS_BEST: tstage_best_one(P, i); break; case TS_TOTAL: tstage_total_one(P, i); break;
        case TS_OPS_COUNT: tstage_ops_count_one(P, i); break; case TS_OPS_COPY: tstage_ops_copy_one(P, i); break;
        default: tstage_read_one(P, i); break;
    }
}

}  // namespace vgk
