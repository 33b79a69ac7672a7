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
    const char* sq = P.index.seq + g_rec(P.index, P.node[v])[3] + P.trim[v];
    uint8_t* o = P.info + P.col[v];
    for (uint32_t k = 0; k < len; ++k) o[k] = (uint8_t)t_ref_code(sq[k]);
    o[0] |= (uint8_t)(CI_NODE_START | (P.slow[v] ? CI_SEED_SLOW : 0));
    if (P.store[v]) o[len - 1] |= (uint8_t)CI_STORE_END;
}

}  // namespace vgk
