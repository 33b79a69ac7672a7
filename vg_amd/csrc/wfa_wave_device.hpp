// wfa_wave_device.hpp — haplotype-consistent wavefront alignment, one WAVEFRONT per problem: the lanes are the diagonals.
//
// The algorithm is wfa_device.hpp's (WFAExtender::connect over a WFATree, reference src/gbwt_extender.cpp:1567-2235); what changes is
// who does the work.  In the reference — and in the one-thread kernel — a penalty step is two loops over (diagonal, leaf of the
// haplotype trie): extend() and next().  Nearly all of their body is pure: table lookups of wavefront points stored under smaller
// penalties, a comparison of bases, one store under a key that no other (diagonal, leaf) pair writes differently.  Here every lane
// takes one (diagonal, leaf) item of the step; the wavefront runs as long as its slowest ITEM instead of the sum of all of them, and
// a launch as long as its slowest problem's critical path instead of that problem's whole work.
//
// What is NOT pure is the trie itself, and results depend on it in one way only: trie nodes are numbered in creation order, and the
// number decides ties (which of two equally good candidates is met first, :1896; which equally good partial alignment trim keeps).
// The rules that keep the numbering — and with it every result — identical to the sequential order:
//   * an item's KEY is (diagonal, leaf) — the reference's iteration order (:1660-1663 per diagonal, get_leaves() in node order);
//   * trie nodes are never grown lazily here: a node is walked when it is created, to its end (branch, dead end, target, 1024
//     bases) or as far as any position of this problem can reach (sequence length + the deletions the score cap allows); a position
//     that still reaches the walked end of an unfinished node ends the problem with VGK_ETOOBIG instead of a wrong answer;
//   * children are created ("expansion", :1992-2008) by ONE lane at a time: an item that needs a node expanded blocks; when every
//     lane is done or blocked, the blocked lane with the smallest key creates the children — every item with a smaller key is
//     done by then, as in the sequential order — and it alone goes on as the reference's recursion does (extend_over on the new
//     children, :1919-1925); lanes that find a node expanded by someone else take its children as items of their own, in node
//     order behind every older leaf (which is where get_leaves() lists them for the following diagonals);
//   * next() never needs the children it asks for: its expansions are applied after the chunk of items, in key order;
//   * chunks of items hold whole diagonals, so a later chunk sees the trie exactly as the sequential loop would at its first diagonal;
//   * lanes that share an ancestor repeat its work and store the same point under the same key: stores are idempotent (64-bit,
//     compare-and-swap to claim a slot); nobody reads a point of the current penalty on another lane's diagonal.
// Candidates carry their key and are merged by (penalty, key): the first among equals in the sequential order.
//
// Tables.  The kernel comes in two sizes (template parameter SMALL).  The small one keeps EVERYTHING of a problem in LDS — wavefront table
// (256 points), its insertion log, the path pool, the edit runs of the backtrace, trie nodes, possible penalties, the lanes' work lists:
// 13 KB per wavefront, 12 wavefronts per CU — and runs every problem; a problem that outgrows it is collected and run again by the large
// one, whose table, log and pool are a slab in HBM per resident wavefront (wfa_api.cpp); only what outgrows that as well is reported
// VGK_ETOOBIG.  A path-pool entry carries what comparing bases needs (where the node's bases lie, how many), and the walk along a
// non-branching path takes a successor's length, bases and record from the edge it follows (the index stores them there): one
// dependent load per hop instead of three.
//
// Merged runs.  A trie node is a non-branching path, and walking one is the wavefront's serial part: one lane follows records hop by hop while
// 63 wait, three or four dependent loads per 32-base node.  When the index has its UNARY RUNS merged (gapless_api.cpp merge_unary_runs: consecutive
// nodes that every haplotype crosses together, one record and one base span per run) the walk hops run by run: WwParams::index is then the merged
// index, a problem's from / to positions are taken onto it at its start, and the path goes back out in the original nodes.  The trie — node for
// node, base for base — stays the original's: WFANode's walk ends a node after the ORIGINAL graph node that holds the target or brings the node
// to 1 024 bases (:1470-1487), so a run that such a node lies in is CUT there (ww_append_piece looks at the run's original lengths, no record
// fetched) and the rest of the run is the only child's first piece.  Offsets, stored points, node numbers and with them every tie are those of the
// node-by-node walk; the oracle (which knows no runs) is the check.
#pragma once
#include "wfa_device.hpp"

namespace vgk {

// VGK_WW_OCC = wavefronts per SIMD the kernel is built for: 3 (shipped: 168 VGPRs, 12.8 KB of LDS) or 4 (an experiment: 128 VGPRs and
// smaller small-size tables so that 16 wavefronts fit a CU's LDS — DESIGN.md §21)
#ifndef VGK_WW_OCC
#define VGK_WW_OCC 3
#endif
constexpr int WW_QUEUE_SMALL = 8, WW_QUEUE_LARGE = VGK_WW_OCC >= 4 ? 20 : 32;     // work-list entries per lane (trie nodes: each is queued at most once per item)
constexpr int WW_SMALL_SLOTS = VGK_WW_OCC >= 4 ? 256 : 512, WW_SMALL_POINTS = VGK_WW_OCC >= 4 ? 128 : 256, WW_SMALL_PATH = VGK_WW_OCC >= 4 ? 96 : 128;

struct WwNode {                       // WNode of wfa_device.hpp + where the record of the path's last graph node lies
    int32_t  st_node, st_lo, st_hi; uint32_t st_rec;
    uint32_t len, target_offset;
    uint16_t path_head, path_tail;
    uint8_t  parent, first_child, n_children, dead_end;
    uint8_t  complete, cut, pad[2];   // cut: the walk ended INSIDE a merged run (after the original node that ends this trie node); the rest of the run is its only child's
    uint32_t ancestors;
};
struct WwPath { int32_t node; uint32_t seq_off; uint16_t start, len, next, inner; };   // a graph node on a trie node's path (or a piece of a merged run: from base `inner` of the run on): its bases at index.seq + seq_off

constexpr uint32_t WW_STAT_WORDS = 12;      // per problem, when WwParams::stats is set (a debugging aid: wfa_api.cpp prints them)
struct WwParams {
    WfaParams base;                   // index, problems, sequences, scoring, outputs, counters[2] = next problem to hand out
    GIndex index; GMerge merge;       // the index this kernel walks: base.index, or its merged-run form (merge.on) with the tables between the two
    const uint32_t* todo; uint32_t n_todo;      // the problems of this launch, in hand-out order
    const unsigned long long* n_todo_dev;        // their number when only the device knows it (the thread kernel's hand-over list); else null
    // The list is still GROWING (the thread kernel runs beside this one): entries not yet written read 0xffffffff, `producers_done` counts
    // the thread kernel's wavefronts that have finished, `n_producers` of them in all.  Null: the list is complete when the launch starts.
    const uint32_t* producers_done; uint32_t n_producers;
    unsigned long long* hand_out;                // this launch's own "next entry" counter (null: base.counters + 2, which the thread kernel uses when it runs beside)
    uint32_t small_points;            // the small size's own point limit when below WW_SMALL_POINTS (0 = that; a test hook: more problems for the large size)
    // the large size: per resident wavefront ...
    unsigned long long* slots; uint32_t n_slots;   // ... n_slots (a power of two) table slots ...
    uint32_t* logs; uint32_t max_points;            // ... max_points log entries (= points a problem may store) ...
    WwPath* paths; uint32_t path_cap;               // ... path_cap pool entries ...
    uint32_t* node_masks; uint32_t mask_width;      // ... WW_MASK_ROWS x 3 x mask_width words: which trie nodes hold a point at (penalty mod WW_MASK_ROWS, kind, diagonal) — null: no item filter
    uint32_t* edit_runs;                            // ... and W_EDITS edit runs for the backtrace
    unsigned long long* n_declined;   // counts the problems the large size took over (nullable)
    uint32_t* stats;                  // nullable (VGAMD_WFA_STATS): per problem 8 words — stored points, penalty steps, chunks of items, trie nodes | items << 8,
                                      // and microseconds (xl.clock_us) in extend(), next(), the penalty bookkeeping between them, everything after the loop
};

#ifndef VGK_WW_FINE_STATS
#define VGK_WW_FINE_STATS 0
#endif
constexpr int WW_LARGE_PATH_LDS = 64;
constexpr int32_t WW_NO_MASKS = 0x7fffffff;
constexpr int WW_MASK_ROWS = 32;      // penalties whose node masks are kept at a time (a ring): more than any source wavefront lies back

template <bool SMALL> struct WwTables {                                       // the large size: its tables are a slab in HBM; in LDS only ...
    uint32_t sv[64];                  // ... a chunk in the making: per diagonal looked at, the leaves that have something to do there
    uint32_t src_mask[64][5];         // ... next(): per diagonal looked at, the node sets of its five source cells (an item probes only the cells that can answer)
    uint32_t filter_off;              // ... and whether some diagonal has left the masks' width (then every item is looked at again, as in the small size)
    int32_t  masks_from;              // ... and the first penalty whose sets are kept: the one after the trie's first expansion (WW_NO_MASKS until then — a trie of one node
                                      //     is never filtered, and most links that outgrow the small tables keep theirs to the end: their sets were 40 % of the kernel's writes)
    WwPath   path_first[WW_LARGE_PATH_LDS];   // ... and the first entries of the path pool (a link that outgrew the small tables did so by its points: its trie is a node or
                                      //     two, and every item of every chunk reads its entry — from the slab that was a trip to HBM each)
};
template <> struct WwTables<true> {   // the small size keeps its tables in LDS
    unsigned long long slot[WW_SMALL_SLOTS]; uint16_t log[WW_SMALL_POINTS]; WwPath path[WW_SMALL_PATH]; uint32_t runs[W_EDITS];
};
template <bool SMALL> struct WwShared : WwTables<SMALL> {                     // per wavefront, in LDS
    static constexpr int QUEUE = SMALL ? WW_QUEUE_SMALL : WW_QUEUE_LARGE;
    WwNode   nodes[W_NODES];
    uint32_t ps_range[W_SCORES]; uint8_t ps_flags[W_SCORES];
    uint8_t  queue[64][QUEUE];        // per lane: origins found expanded by someone else, in node order (FIFO)
    uint8_t  stack_cur[64][QUEUE], stack_end[64][QUEUE];      // per lane: child ranges of the expansions it made itself (the recursion)
    int32_t  expanded_at[W_NODES];    // next(): the diagonal of the item whose request expanded the node in the current chunk
    uint32_t n_nodes, n_path, n_points, leaves;
};

enum { WX_FETCH = 0, WX_MATCH = 1, WX_AFTER = 2, WX_BLOCKED = 3, WX_DONE = 4 };

template <class XL, bool SMALL> struct WwCtx {
    const WwParams* P; WwShared<SMALL>* sh; XL* xl; uint32_t lane;
    unsigned long long* slot; uint32_t* log; WwPath* path; uint32_t* runs;      // the large size's slab (unused by the small one)
    uint32_t mask, max_points, path_cap;
    uint32_t* masks; uint32_t mask_width;                                       // the item filter's node masks (large size; null = off)
    const char* seq; uint32_t L;
    int32_t to_node; uint32_t to_off; bool no_to;      // the target in the index walked: (merged) node, offset there ...
    uint32_t to_inner, to_orig_len;   // ... and where in that node the ORIGINAL target node starts, and its length (merged runs; 0, the node's length otherwise)
    uint32_t grow_cap;                // bases a trie node is walked at its creation unless it ends earlier
    int32_t min_distance;
    // this lane's findings, merged after every phase
    int32_t cand_score, cand_diag; uint32_t cand_seq, cand_off, cand_node, cand_leaf;
    int32_t max_distance;
    uint32_t n_chunks, n_steps, n_items;      // (statistics)
    uint32_t us_extend, us_next, us_score;
#if VGK_WW_FINE_STATS
    uint32_t us_fine[4], t_fine;      // (an experiment's build: where next() spends its time — filter + chunk, lookups, stores, expansions + merge)
#endif
    bool overflow; int why;           // why: 1 points, 2 trie nodes, 3 path pool, 4 edits, 5 node length, 6 walked end reached, 7 work list,
                                      // 8 table without a free slot, 9 broken path chain, 10 a loop ran past its bound (8-10: cannot happen; never hang)
    VGK_HD unsigned long long* tbl(uint32_t i) { if constexpr (SMALL) return sh->slot + i; else return slot + i; }
    VGK_HD void log_put(uint32_t at, uint32_t i) { if constexpr (SMALL) sh->log[at] = (uint16_t)i; else log[at] = i; }
    VGK_HD uint32_t log_at(uint32_t k) const { if constexpr (SMALL) return sh->log[k]; else return log[k]; }
    VGK_HD WwPath pth(uint32_t k) const { if constexpr (SMALL) return sh->path[k]; else { if (k < (uint32_t)WW_LARGE_PATH_LDS) return sh->path_first[k]; return path[k]; } }
    VGK_HD void pth_put(uint32_t k, const WwPath& e) { if constexpr (SMALL) sh->path[k] = e; else { if (k < (uint32_t)WW_LARGE_PATH_LDS) sh->path_first[k] = e; else path[k] = e; } }
    VGK_HD void pth_link(uint32_t k, uint16_t next) { if constexpr (SMALL) sh->path[k].next = next; else { if (k < (uint32_t)WW_LARGE_PATH_LDS) sh->path_first[k].next = next; else path[k].next = next; } }
    VGK_HD uint32_t* run_buf() { if constexpr (SMALL) return sh->runs; else return runs; }
    // the word that names the trie nodes holding a point of (kind, score, diag); null when the diagonal lies outside the masks
    VGK_HD uint32_t* mask_word(int kind, int32_t score, int32_t diag) const {
        const int32_t at = diag + (int32_t)(mask_width >> 1);
        if (at < 0 || at >= (int32_t)mask_width) return nullptr;
        return masks + ((size_t)((uint32_t)score % (uint32_t)WW_MASK_ROWS) * 3u + (uint32_t)kind) * mask_width + (uint32_t)at;
    }
};

// ---- possible penalties ----
template <class XL, bool SMALL> VGK_HD WPScore ww_ps(const WwCtx<XL, SMALL>& c, int32_t score) {
    WPScore p; p.flags = c.sh->ps_flags[score];
    const uint32_t r = c.sh->ps_range[score];
    p.min_d = (int16_t)(r & 0xffffu); p.max_d = (int16_t)(r >> 16);
    return p;
}
template <class XL, bool SMALL> VGK_HD WSrc ww_src(const WwCtx<XL, SMALL>& c, int32_t score) {
    WSrc s = { score, 1, 0 };
    if (score < 0) return s;
    const WPScore ps = ww_ps(c, score);
    if (ps.flags & 1) { s.lo = ps.min_d; s.hi = ps.max_d; }
    return s;
}

// ---- the wavefront table: w_key / the node-free hash of wfa_device.hpp over this launch's slot count ----
// The small size (LDS, up to half full) scatters cells; the large size (an HBM slab of 4 MB per wavefront, a few percent full) keeps eight neighbouring
// diagonals of one (kind, penalty) in ONE 128-byte line — 16 slots, a diagonal's home at every second one, so that the free slot that ends its lookup lies in
// the line too: the lanes of a chunk are neighbouring diagonals, and with scattered cells every probe, every new point and every slot wiped at the end was a
// line of its own fetched and written back for 8 bytes (the kernel's HBM traffic was 52 x its algorithmic bytes, profiles/r06/NOTES.md §4).
template <class XL, bool SMALL> VGK_HD uint32_t ww_hash(const WwCtx<XL, SMALL>& c, uint32_t key) {
    const uint32_t cell = (key - 1u) >> 5;                                      // kind | penalty << 2 | (diagonal + 512) << 12
    if constexpr (SMALL) return ((cell * 2654435761u) >> 8) & c.mask;
    else {
        const uint32_t group = (cell & 0xfffu) | ((cell >> 15) << 12);
        return (((((group * 2654435761u) >> 8) << 4) | (((cell >> 12) & 7u) << 1))) & c.mask;
    }
}
template <class XL, bool SMALL> VGK_HD bool ww_lookup(WwCtx<XL, SMALL>& c, uint32_t ancestors, int kind, int32_t score, int32_t diag, uint32_t& node, uint32_t& seq, uint32_t& off) {
    const uint32_t cell = (w_key(0, kind, score, diag) - 1u) >> 5;
    bool found = false; uint32_t best = 0;
    uint32_t probes = 0;
    for (uint32_t i = ww_hash(c, w_key(0, kind, score, diag));; i = (i + 1) & c.mask) {
        if (probes++ > c.mask) { c.overflow = true; c.why = 8; break; }         // a table without a free slot cannot be: say so instead of probing for ever
        const unsigned long long s = c.xl->load64(c.tbl(i));
        if (!s) break;
        const uint32_t key = (uint32_t)(s >> 32) - 1u, holder = key & 31u;
        if ((key >> 5) == cell && ((ancestors >> holder) & 1u) && (!found || holder > best)) {
            found = true; best = holder; seq = (uint32_t)(s >> 16) & 0xffffu; off = (uint32_t)s & 0xffffu;
        }
    }
    node = best;
    return found;
}
// The same lookup with its first two slots read ahead (WwProbe): next() reads the first slots of all five source cells of an item TOGETHER — ten
// loads in flight, then five walks that mostly end within what is already there (a cell that is found takes two probes: the point and the free slot
// behind it) — instead of five chains of dependent loads from the HBM slab one after the other.  Scalars, not arrays: arrays of probe states went
// to scratch memory when this was tried with loops (profiles/r03).
struct WwProbe { uint32_t i; unsigned long long a, b; bool on; };
template <class XL, bool SMALL> VGK_HD bool ww_at_dead_end(WwCtx<XL, SMALL>& c, const WPos& p);
template <class XL, bool SMALL> VGK_HD WwProbe ww_probe(WwCtx<XL, SMALL>& c, bool on, int kind, const WSrc& src, int32_t diag) {
    WwProbe p; p.on = on && !(diag < src.lo || diag > src.hi); p.i = 0; p.a = 0; p.b = 0;
    if (p.on) { p.i = ww_hash(c, w_key(0, kind, src.score, diag)); p.a = c.xl->load64(c.tbl(p.i)); p.b = c.xl->load64(c.tbl((p.i + 1) & c.mask)); }
    return p;
}
template <class XL, bool SMALL> VGK_HD WPos ww_find_probed(WwCtx<XL, SMALL>& c, const WwProbe& pr, int kind, const WSrc& src, uint32_t ancestors, uint32_t origin, int32_t diag, bool ext_seq, bool ext_graph) {
    if (!pr.on) return w_none();
    const uint32_t cell = (w_key(0, kind, src.score, diag) - 1u) >> 5;
    bool found = false; uint32_t best = 0, seq = 0, off = 0, probes = 0;
    for (uint32_t i = pr.i;; i = (i + 1) & c.mask) {
        if (probes > c.mask) { c.overflow = true; c.why = 8; break; }
        const unsigned long long s = probes == 0 ? pr.a : probes == 1 ? pr.b : c.xl->load64(c.tbl(i));
        ++probes;
        if (!s) break;
        const uint32_t key = (uint32_t)(s >> 32) - 1u, holder = key & 31u;
        if ((key >> 5) == cell && ((ancestors >> holder) & 1u) && (!found || holder > best)) {
            found = true; best = holder; seq = (uint32_t)(s >> 16) & 0xffffu; off = (uint32_t)s & 0xffffu;
        }
    }
    if (!found) return w_none();
    WPos p = { seq, off, (uint8_t)best, (uint8_t)origin, false };
    if (ext_seq && p.seq >= c.L) return w_none();
    if (ext_graph && ww_at_dead_end(c, p)) return w_none();
    return p;
}
// A point is stored in two halves — the trip to its home slot (a CAS against "free"), then what that found — so that next() can have an item's three
// stores (insertion, deletion, match) on their way together instead of three trips one after the other.
struct WwPut { uint32_t i, key; unsigned long long v, old; bool on; };
template <class XL, bool SMALL> VGK_HD WwPut ww_put_begin(WwCtx<XL, SMALL>& c, bool on, uint32_t node, int kind, int32_t score, int32_t diag, uint32_t seq, uint32_t off) {
    WwPut p; p.on = on; p.i = 0; p.key = 0; p.v = 0; p.old = 0;
    if (!on) return p;
    if constexpr (!SMALL) {
        if (c.masks && score >= c.sh->masks_from) {
            uint32_t* w = c.mask_word(kind, score, diag);
            if (w) c.xl->or32(w, 1u << node); else c.sh->filter_off = 1u;       // (read by everyone after the phase's fence)
        }
    }
    p.key = w_key(node, kind, score, diag);
    p.v = ((unsigned long long)p.key << 32) | ((unsigned long long)(seq & 0xffffu) << 16) | (off & 0xffffu);
    p.i = ww_hash(c, p.key);
    if (c.sh->n_points >= c.max_points + 64u) { c.overflow = true; c.why = 1; p.on = false; return p; }     // (someone's table has run over already: do not pile on)
    p.old = c.xl->cas64(c.tbl(p.i), 0ull, p.v);                               // (no load first: a new point is one trip to its slot, not two)
    return p;
}
template <class XL, bool SMALL> VGK_HD void ww_put_end(WwCtx<XL, SMALL>& c, const WwPut& p) {
    if (!p.on) return;
    uint32_t i = p.i, probes = 0;
    for (unsigned long long s = p.old;;) {
        if (!s) {                                                              // the slot is ours: a new point
            const uint32_t at = c.xl->add32(&c.sh->n_points, 1u);
            if (at >= c.max_points) { c.overflow = true; c.why = 1; return; }         // (the table is wiped whole after an overflow)
            c.log_put(at, i);
            return;
        }
        if ((uint32_t)(s >> 32) == p.key) { c.xl->store64(c.tbl(i), p.v); return; }
        i = (i + 1) & c.mask;
        if (probes++ > c.mask) { c.overflow = true; c.why = 8; return; }
        if (c.sh->n_points >= c.max_points + 64u) { c.overflow = true; c.why = 1; return; }
        s = c.xl->cas64(c.tbl(i), 0ull, p.v);
    }
}
template <class XL, bool SMALL> VGK_HD void ww_store(WwCtx<XL, SMALL>& c, uint32_t node, int kind, int32_t score, int32_t diag, uint32_t seq, uint32_t off) {
    const WwPut p = ww_put_begin(c, true, node, kind, score, diag, seq, off);
    ww_put_end(c, p);
}

// ---- trie nodes (all of these read only; the trie changes in ww_node_create / ww_expand, one lane at a time) ----
template <class XL, bool SMALL> VGK_HD bool ww_past_end(WwCtx<XL, SMALL>& c, uint32_t id, uint32_t off) {
    const WwNode& n = c.sh->nodes[id];
    if (off < n.len) return false;
    if (!n.complete) { c.overflow = true; c.why = 6; }                       // beyond what was walked: never guess
    return true;
}
template <class XL, bool SMALL> VGK_HD void ww_pop(const WwCtx<XL, SMALL>& c, WPos& p) {
    const uint32_t below = c.sh->nodes[p.origin].ancestors & ~((2u << p.cur) - 1u);
    p.cur = (uint8_t)__builtin_ctz(below);
}
template <class XL, bool SMALL> VGK_HD bool ww_at_dead_end(WwCtx<XL, SMALL>& c, const WPos& p) { return ww_past_end(c, p.cur, p.off) && c.sh->nodes[p.cur].dead_end; }
template <class XL, bool SMALL> VGK_HD WPos ww_find_in(WwCtx<XL, SMALL>& c, int kind, const WSrc& src, uint32_t ancestors, uint32_t origin, int32_t diag, bool ext_seq, bool ext_graph) {
    if (diag < src.lo || diag > src.hi) return w_none();
    uint32_t holder = 0, seq = 0, off = 0;
    if (!ww_lookup(c, ancestors, kind, src.score, diag, holder, seq, off)) return w_none();
    WPos p = { seq, off, (uint8_t)holder, (uint8_t)origin, false };
    if (ext_seq && p.seq >= c.L) return w_none();
    if (ext_graph && ww_at_dead_end(c, p)) return w_none();
    return p;
}
template <class XL, bool SMALL> VGK_HD WPos ww_find_pos(WwCtx<XL, SMALL>& c, int kind, uint32_t node, int32_t score, int32_t diag, bool ext_seq, bool ext_graph) {
    return ww_find_in(c, kind, ww_src(c, score), c.sh->nodes[node].ancestors, node, diag, ext_seq, ext_graph);
}
template <class XL, bool SMALL> VGK_HD void ww_update(WwCtx<XL, SMALL>& c, int kind, int32_t score, int32_t diag, const WPos& p) { ww_store(c, p.cur, kind, score, diag, p.seq, p.off); }
template <class XL, bool SMALL> VGK_HD void ww_successor_offset(WwCtx<XL, SMALL>& c, WPos& p) {
    if (ww_past_end(c, p.cur, p.off)) { ww_pop(c, p); p.off = 0; }
    p.off++;
}
template <class XL, bool SMALL> VGK_HD void ww_predecessor_offset(const WwCtx<XL, SMALL>& c, uint32_t& node, uint32_t& off) {
    if (off > 0) --off;
    else { node = c.sh->nodes[node].parent; off = c.sh->nodes[node].len - 1; }
}
template <class XL, bool SMALL> VGK_HD int32_t ww_gap_penalty(const WwCtx<XL, SMALL>& c, uint32_t length) { return c.P->base.gap_open + (int32_t)length * c.P->base.gap_extend; }
template <class XL, bool SMALL> VGK_HD void ww_candidate(WwCtx<XL, SMALL>& c, int32_t score, int32_t diag, uint32_t seq, uint32_t off, uint32_t node, uint32_t leaf) {
    if (score < c.cand_score) { c.cand_score = score; c.cand_diag = diag; c.cand_seq = seq; c.cand_off = off; c.cand_node = node; c.cand_leaf = leaf; }
}
template <class XL, bool SMALL> VGK_HD bool ww_wants_expansion(WwCtx<XL, SMALL>& c, const WPos& p) {                 // the test of expand_if_necessary (:1995-1997)
    const WwNode& n = c.sh->nodes[p.cur];
    return !n.n_children && !n.dead_end && ww_past_end(c, p.cur, p.off);
}

template <class XL, bool SMALL> VGK_HD void ww_match_forward(WwCtx<XL, SMALL>& c, WPos& p) {
    if (p.seq >= c.L || ww_past_end(c, p.cur, p.off)) return;
    const GIndex& h = c.P->index;
    // the entry that holds offset p.off: along the node's chain from its head (the entries of sibling nodes interleave in the pool — their walks run
    // on as many lanes at once —, so no search by index; over merged runs a node has two or three entries, a position is one or two hops down)
    uint32_t k = c.sh->nodes[p.cur].path_head;
    uint32_t hops = 0;
    for (;;) {
        if (hops++ > 2 * c.path_cap || k >= c.path_cap) { c.overflow = true; c.why = 9; return; }       // a broken chain: never walk it for ever
        const WwPath e = c.pth(k);                                             // one entry: where the node's bases lie and how many
        if (p.off >= (uint32_t)e.start + e.len) { k = e.next; continue; }      // (the entries before the offset's: a node over merged runs has two or three in all)
        if (p.off < e.start) { c.overflow = true; c.why = 9; return; }        // (not the entry of this offset: cannot happen)
        const char* g = h.seq + e.seq_off + (p.off - e.start);
        const char* r = c.seq + p.seq;
        uint32_t left = (uint32_t)e.start + e.len - p.off; if (c.L - p.seq < left) left = c.L - p.seq;
        uint32_t m = 0;
        while (m < left) {                                                     // 32 bases per round: the eight loads are in flight together (an item's match is ONE lane's chain: nothing else hides their latency)
            const uint64_t x0 = g_load8(g + m) ^ g_load8(r + m);
            const uint64_t x1 = m + 8 < left ? g_load8(g + m + 8) ^ g_load8(r + m + 8) : 0ull, x2 = m + 16 < left ? g_load8(g + m + 16) ^ g_load8(r + m + 16) : 0ull,
                           x3 = m + 24 < left ? g_load8(g + m + 24) ^ g_load8(r + m + 24) : 0ull;
            if (x0) { m += (uint32_t)(__builtin_ctzll(x0) >> 3); break; }
            if (x1) { m += 8u + (uint32_t)(__builtin_ctzll(x1) >> 3); break; }
            if (x2) { m += 16u + (uint32_t)(__builtin_ctzll(x2) >> 3); break; }
            if (x3) { m += 24u + (uint32_t)(__builtin_ctzll(x3) >> 3); break; }
            m += 32;
        }
        const bool differs = m < left;
        if (!differs) m = left;
        p.seq += m; p.off += m;
        if (differs || p.seq >= c.L || ww_past_end(c, p.cur, p.off) || c.overflow) return;
        k = e.next;
    }
}

// ---- changes to the trie: one lane, everyone else waiting at a fence ----
// the graph node a walk steps to: its search state and, from the edge followed, its length, bases and record
// (with merged runs `len` bases of the run from its base `inner` on — a PIECE: the whole run, or what a cut left of it)
struct WwStep { WState state; uint32_t len, seq_off, rec, inner; };
// the original nodes of a run, in the order its strand reads them: node index and length of the one that starts `inner` bases into the run, then on
struct WwOriginals {
    const GMerge* M; uint32_t v, v_end; bool reverse; uint32_t at;          // at: bases of the run before original v
    VGK_HD uint32_t length() const { return M->ocol[v + 1] - M->ocol[v]; }
    VGK_HD uint32_t oriented() const { return 2u * v + (reverse ? 1u : 0u); }
    VGK_HD bool done() const { return v == v_end; }
    VGK_HD void step() { at += length(); v = reverse ? v - 1u : v + 1u; }
};
VGK_HD WwOriginals ww_originals(const GMerge& M, uint32_t merged_oriented, uint32_t inner) {
    const uint32_t m = merged_oriented >> 1, v0 = M.run_first[m], v1 = M.run_first[m + 1];
    WwOriginals it; it.M = &M; it.reverse = (merged_oriented & 1u) != 0; it.at = 0;
    it.v = it.reverse ? v1 - 1u : v0; it.v_end = it.reverse ? v0 - 1u : v1;
    while (!it.done() && it.at < inner) it.step();
    return it;
}
// Appends a piece to a trie node the way WFANode's walk appends its original nodes one by one (:1470-1487): the node ends after the original that
// holds the target, or that brings it to W_TARGET_LENGTH bases — inside a run, the piece is cut there (n.cut).  -> whether the node is complete.
template <class XL, bool SMALL> VGK_HD bool ww_append_piece(WwCtx<XL, SMALL>& c, WwNode& n, const WwStep& next) {
    n.st_node = next.state.node; n.st_lo = next.state.lo; n.st_hi = next.state.hi; n.st_rec = next.rec;
    uint32_t take = next.len; bool complete = false, at_target = false;
    const bool holds_target = !c.no_to && c.to_node == next.state.node && c.to_inner >= next.inner;
    if (holds_target) { take = c.to_inner + c.to_orig_len - next.inner; complete = true; at_target = true; }
    if (n.len + take >= W_TARGET_LENGTH && c.P->merge.on) {
        // the first original whose end brings the node to the target length (it may lie before the target's)
        WwOriginals it = ww_originals(c.P->merge, (uint32_t)next.state.node, next.inner);
        uint32_t got = 0;
        while (!it.done() && got < take) { got += it.length(); it.step(); if (n.len + got >= W_TARGET_LENGTH) break; }
        if (got < take) { take = got; complete = true; at_target = false; }
    }
    const uint32_t slot_at = c.xl->add32(&c.sh->n_path, 1u);                   // (the children of an expansion are walked by as many lanes at once: ww_expand_wave)
    if (slot_at >= c.path_cap) { c.overflow = true; c.why = 3; return true; }
    const uint16_t at = (uint16_t)slot_at;
    WwPath e; e.node = next.state.node; e.seq_off = next.seq_off; e.start = (uint16_t)n.len; e.len = (uint16_t)take; e.next = W_NIL; e.inner = (uint16_t)next.inner;
    c.pth_put(at, e);
    if (n.path_head == W_NIL) n.path_head = at; else c.pth_link(n.path_tail, at);
    n.path_tail = at;
    n.len += take;
    n.cut = take < next.len ? 1 : 0;
    if (n.len > 0xfff0u) { c.overflow = true; c.why = 5; return true; }
    if (at_target) n.target_offset = n.len - (c.to_orig_len - c.to_off);
    return complete;
}
// w_follow over a record that is already at hand; the step's length / bases / record come from the edge (gapless_device.hpp: record layout)
template <class XL, bool SMALL> VGK_HD uint32_t ww_follow(WwCtx<XL, SMALL>& c, uint32_t rec_off, const WState& s, uint32_t want, WwStep& out, uint32_t stop_at) {
    const uint32_t* rec = c.P->index.rec + rec_off;
    uint32_t edge = 0;
    const uint32_t k = w_follow_rec(rec, s, want, out.state, stop_at, &edge);
    if (k > want) { out.len = ge_len(rec, edge); out.seq_off = ge_seq(rec, edge); out.rec = ge_rec(rec, edge); out.inner = 0; }
    return k;
}
// WFANode's constructor (:1463-1487), walked as far as this problem can look
template <class XL, bool SMALL> VGK_HD void ww_node_create(WwCtx<XL, SMALL>& c, uint32_t id, const WwStep& first, uint32_t parent, uint32_t reach) {
    WwNode n;
    n.len = 0; n.target_offset = W_NO_OFFSET; n.path_head = n.path_tail = W_NIL;
    n.parent = (uint8_t)parent; n.first_child = 0; n.n_children = 0; n.dead_end = 0; n.cut = 0; n.pad[0] = n.pad[1] = 0;
    n.ancestors = (id ? c.sh->nodes[parent].ancestors : 0u) | (1u << id);
    c.xl->or32(&c.sh->leaves, 1u << id);
    n.complete = ww_append_piece(c, n, first) ? 1 : 0;
    for (uint32_t hops = 0; !n.complete && !c.overflow; ++hops) {
        if (hops > c.path_cap) { c.overflow = true; c.why = 9; break; }
        if (n.len >= W_TARGET_LENGTH) { n.complete = 1; break; }
        if (n.len >= reach) break;                                           // nothing of this problem gets further; the node stays unfinished
        const WState cur = { n.st_node, n.st_lo, n.st_hi }; WwStep next; next.state.node = 0; next.state.lo = 0; next.state.hi = -1;
        const uint32_t successors = ww_follow(c, n.st_rec, cur, 0, next, 2);
        if (successors == 0) { n.dead_end = 1; n.complete = 1; }
        else if (successors > 1) n.complete = 1;
        else if (ww_append_piece(c, n, next)) n.complete = 1;
    }
    c.sh->nodes[id] = n;
}
// expand_if_necessary (:1992-2008) for a node known to be at its end; -> whether this call made the children
template <class XL, bool SMALL> VGK_HD bool ww_expand(WwCtx<XL, SMALL>& c, uint32_t node) {
    if (c.sh->nodes[node].n_children || c.sh->nodes[node].dead_end) return false;
    const WState st = { c.sh->nodes[node].st_node, c.sh->nodes[node].st_lo, c.sh->nodes[node].st_hi };
    const uint32_t rec = c.sh->nodes[node].st_rec;
    WwStep next; next.state.node = 0; next.state.lo = 0; next.state.hi = -1;
    if (c.sh->nodes[node].cut) {
        // the walk ended inside a run: the original's next graph node is the run's next original, reached by every haplotype of the state — one
        // child, whose first piece is the rest of the run (the same search state, the same record)
        if (c.sh->n_nodes + 1 > (uint32_t)W_NODES) { c.overflow = true; c.why = 2; return false; }
        const WwPath last = c.pth(c.sh->nodes[node].path_tail);
        uint32_t run_len = 0; const uint32_t run_seq = g_seq_of(c.P->index, (uint32_t)st.node, run_len);
        (void)run_seq;
        next.state = st; next.rec = rec; next.inner = (uint32_t)last.inner + last.len; next.seq_off = last.seq_off + last.len; next.len = run_len - next.inner;
        c.sh->nodes[node].first_child = (uint8_t)c.sh->n_nodes; c.sh->nodes[node].n_children = 1;
        c.sh->leaves &= ~(1u << node);
        ww_node_create(c, c.sh->n_nodes, next, node, c.grow_cap); ++c.sh->n_nodes;
        return true;
    }
    const uint32_t k = ww_follow(c, rec, st, 0, next, 0xffffffffu);
    if (!k) { c.sh->nodes[node].dead_end = 1; return false; }
    if (c.sh->n_nodes + k > (uint32_t)W_NODES) { c.overflow = true; c.why = 2; return false; }
    c.sh->nodes[node].first_child = (uint8_t)c.sh->n_nodes; c.sh->nodes[node].n_children = (uint8_t)k;
    c.sh->leaves &= ~(1u << node);
    for (uint32_t i = 0; i < k; ++i) {
        if (i) ww_follow(c, rec, st, i, next, i + 1);
        ww_node_create(c, c.sh->n_nodes, next, node, c.grow_cap); ++c.sh->n_nodes;
        if (c.overflow) return true;
    }
    return true;
}

// The same by the whole wavefront — EVERY lane calls, with the same node: the children are walked by as many lanes at once (a walk is a chain of
// dependent record loads; two children one after the other was half of extend()'s time in a link with a bubble every hundred bases).  Every lane
// reads the node's record (the same words: one fetch) and takes extension number `lane` of it; the numbers the children get are those of the
// loop — first_child + their position among the edges.  -> on lane `who` only: whether this call made the children.
template <class XL, bool SMALL> VGK_HD bool ww_expand_wave(WwCtx<XL, SMALL>& c, uint32_t node, uint32_t who) {
    const WwNode par = c.sh->nodes[node];
    if (par.n_children || par.dead_end) return false;
    if (par.cut) return c.lane == who ? ww_expand(c, node) : false;             // (one child, the rest of the run: nothing to share)
    const WState st = { par.st_node, par.st_lo, par.st_hi };
    WwStep next; next.state.node = 0; next.state.lo = 0; next.state.hi = -1;
    const uint32_t k = ww_follow(c, par.st_rec, st, c.lane, next, 0xffffffffu);
    const uint32_t base = c.sh->n_nodes;
    c.xl->fence_lds();                                                         // (everyone has read the node and the count before they change)
    if (!k) { if (c.lane == who) c.sh->nodes[node].dead_end = 1; return false; }
    if (base + k > (uint32_t)W_NODES) { c.overflow = true; c.why = 2; return false; }
    if (c.lane == who) {
        c.sh->nodes[node].first_child = (uint8_t)base; c.sh->nodes[node].n_children = (uint8_t)k;
        c.sh->leaves &= ~(1u << node);
        c.sh->n_nodes = base + k;
    }
    c.xl->fence_lds();
    if (c.lane < k) ww_node_create(c, base + c.lane, next, node, c.grow_cap);
    return c.lane == who;
}

// the lanes agree on whether anyone has failed; the reason of the lowest such lane is everyone's
template <class XL, bool SMALL> VGK_HD bool ww_any_overflow(WwCtx<XL, SMALL>& c) {
    const unsigned long long bad = c.xl->ballot(c.overflow);
    if (!bad) return false;
    const uint32_t who = (uint32_t)__builtin_ctzll(bad);
    c.why = (int)c.xl->bcast((uint32_t)c.why, who); c.overflow = true;
    return true;
}
VGK_HD uint32_t ww_nth_bit(uint32_t mask, uint32_t n) { for (; n; --n) mask &= mask - 1; return (uint32_t)__builtin_ctz(mask); }
VGK_HD uint32_t ww_popcount(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

// ---- which items a chunk holds ----
// The reference's step is two loops over a rectangle, (every diagonal of the wavefront) x (every leaf of the trie); with dozens of leaves
// nine items in ten find no point to start from — a point lies on ONE trie node and serves only the leaves below it.  The large size keeps,
// per (penalty, kind, diagonal), the set of trie nodes that hold a point (a 32-bit word, set by ww_store; WW_MASK_ROWS penalties at a time)
// and looks at an item only if some node on its leaf's way to the root is in the sets its lookups would probe.  Items keep their order
// (diagonal, then leaf), a chunk still holds whole diagonals — only more of them, up to 64 — and an item that is skipped is one the
// full loop would have found nothing for: nothing observes the difference.  `m` = this lane's diagonal's (diag0 + lane) union of sets.
template <class XL> VGK_HD uint32_t ww_chunk_of(WwCtx<XL, false>& c, int32_t diag0, int32_t hi, uint32_t m, int32_t& diag, uint32_t& leaf, bool& have) {
    uint32_t sv = 0;
    if (m) for (uint32_t rest = c.sh->leaves; rest; rest &= rest - 1) { const uint32_t l = (uint32_t)__builtin_ctz(rest); if (c.sh->nodes[l].ancestors & m) sv |= 1u << l; }
    c.sh->sv[c.lane] = sv;
    c.xl->fence_lds();
    uint32_t acc = 0, taken = 0;
    have = false;
    for (uint32_t j = 0; j < 64u && diag0 + (int32_t)j <= hi; ++j) {
        const uint32_t w = c.sh->sv[j], cnt = (uint32_t)__builtin_popcount(w);
        if (acc + cnt > 64u) break;                                            // (never the first diagonal: a trie has at most 32 leaves)
        if (c.lane >= acc && c.lane < acc + cnt) {
            have = true; diag = diag0 + (int32_t)j;
            uint32_t rest = w; for (uint32_t n = c.lane - acc; n; --n) rest &= rest - 1;
            leaf = (uint32_t)__builtin_ctz(rest);
        }
        acc += cnt; taken = j + 1;
    }
    c.xl->fence_lds();                                                         // (sv belongs to the next chunk from here)
    c.n_items += acc;
    return taken;
}
template <class XL, bool SMALL> VGK_HD uint32_t ww_mask_at(WwCtx<XL, SMALL>& c, int kind, const WSrc& src, int32_t diag) {
    if constexpr (SMALL) return 0xffffffffu;
    else {
        if (src.lo > src.hi || diag < src.lo || diag > src.hi) return 0u;
        uint32_t* w = c.mask_word(kind, src.score, diag);
        return w ? c.xl->load32(w) : 0xffffffffu;
    }
}
// is the filter on for the coming chunk?  `need` = the oldest penalty whose sets the chunk reads (the same answer on every lane: filter_off and masks_from
// change only between fences)
template <class XL, bool SMALL> VGK_HD bool ww_filtering(WwCtx<XL, SMALL>& c, int32_t need) {
    if constexpr (SMALL) return false;
    else { if (!c.masks) return false; c.xl->fence_lds(); return c.sh->filter_off == 0u && need >= c.sh->masks_from; }
}
template <class XL, bool SMALL> VGK_HD void ww_clear_masks(WwCtx<XL, SMALL>& c, int32_t score) {       // a penalty's sets, before its first point is stored
    if constexpr (!SMALL) {
        if (!c.masks || score < c.sh->masks_from) return;
        uint32_t* row = c.masks + (size_t)((uint32_t)score % (uint32_t)WW_MASK_ROWS) * 3u * c.mask_width;
        for (uint32_t j = c.lane; j < 3u * c.mask_width; j += 64u) row[j] = 0u;
        c.xl->fence();
    }
}
// the trie has grown beyond its root during the phase of penalty `score`: the sets are kept from the next penalty on (this one's points are partly stored already)
template <class XL, bool SMALL> VGK_HD void ww_masks_begin(WwCtx<XL, SMALL>& c, int32_t score) {
    if constexpr (!SMALL) {
        if (!c.masks) return;
        if (c.sh->masks_from == WW_NO_MASKS && c.sh->n_nodes > 1u) c.sh->masks_from = score + 1;      // (every lane the same value)
        c.xl->fence_lds();
    }
}

// ---- extend(): an item runs until it is done or needs a node expanded ----
struct WwItem { int st; uint32_t qh, qt, sp, key_leaf, blocked_on; WPos pos; bool creator; };

template <class XL, bool SMALL> VGK_HD void ww_extend_run(WwCtx<XL, SMALL>& c, const WSrc& here, int32_t score, int32_t diag, WwItem& it) {
    uint8_t* queue = c.sh->queue[c.lane]; uint8_t* scur = c.sh->stack_cur[c.lane]; uint8_t* send = c.sh->stack_end[c.lane];
    for (uint32_t turns = 0;; ++turns) {
        if (turns > 8u * W_NODES + 64u) { c.overflow = true; c.why = 10; }       // every trie node is visited a bounded number of times per item
        if (c.overflow) { it.st = WX_DONE; return; }
        if (it.st == WX_FETCH) {
            uint32_t leaf;
            while (it.sp && scur[it.sp - 1] == send[it.sp - 1]) --it.sp;
            if (it.sp) leaf = scur[it.sp - 1]++;                               // the recursion over children this item created itself
            else if (it.qh < it.qt) { leaf = queue[it.qh++]; it.key_leaf = leaf; }    // the next leaf of this item's own
            else { it.st = WX_DONE; return; }
            it.pos = ww_find_in(c, WK_MATCH, here, c.sh->nodes[leaf].ancestors, leaf, diag, false, false);
            if (it.pos.empty) continue;
            it.st = WX_MATCH;
        }
        if (it.st == WX_MATCH) {
            WPos& pos = it.pos;
            const uint32_t off_before = pos.off;
            ww_match_forward(c, pos);
            const bool at_end = ww_past_end(c, pos.cur, pos.off);
            const uint32_t target_offset = c.no_to ? W_NO_OFFSET : c.sh->nodes[pos.cur].target_offset;
            const bool may_reach_target = target_offset != W_NO_OFFSET && target_offset >= off_before;
            if ((may_reach_target && pos.off >= target_offset) || (c.no_to && pos.seq >= c.L)) {
                const uint32_t overshoot = c.no_to ? 0 : pos.off - target_offset;
                const uint32_t gap_length = (c.L - pos.seq) + overshoot;
                ww_candidate(c, score + (gap_length ? ww_gap_penalty(c, gap_length) : 0), diag, pos.seq - overshoot, target_offset, pos.cur, it.key_leaf);
            }
            if (w_distance(pos, diag) > c.max_distance) c.max_distance = w_distance(pos, diag);
            ww_update(c, WK_MATCH, score, diag, pos);
            if (c.overflow) { it.st = WX_DONE; return; }
            if (!at_end) { it.st = WX_FETCH; continue; }
            it.creator = false;
            if (!c.sh->nodes[pos.cur].n_children && !c.sh->nodes[pos.cur].dead_end) { it.st = WX_BLOCKED; it.blocked_on = pos.cur; return; }
            it.st = WX_AFTER;
        }
        if (it.st == WX_AFTER) {                                               // at the end of a node that is expanded (or a dead end) by now
            WPos& pos = it.pos;
            if (pos.cur == pos.origin) {
                const WwNode& now = c.sh->nodes[pos.cur];
                if (now.n_children) {
                    if (it.creator) {                                           // the reference's recursion: the new children at once, in order
                        if (it.sp >= (uint32_t)WwShared<SMALL>::QUEUE) { c.overflow = true; c.why = 7; it.st = WX_DONE; return; }
                        scur[it.sp] = now.first_child; send[it.sp] = (uint8_t)(now.first_child + now.n_children); ++it.sp;
                    } else {                                                    // someone with a smaller key made them: they are leaves of this diagonal, behind the older ones
                        for (uint32_t k = 0; k < now.n_children; ++k) {
                            if (it.qt >= (uint32_t)WwShared<SMALL>::QUEUE) { c.overflow = true; c.why = 7; it.st = WX_DONE; return; }
                            queue[it.qt++] = (uint8_t)(now.first_child + k);
                        }
                    }
                }
                it.st = WX_FETCH; continue;
            }
            ww_pop(c, pos); pos.off = 0;
            it.st = WX_MATCH; continue;
        }
        return;
    }
}

// merge the lanes' candidates into every lane's copy of the best: smallest penalty, then the smallest key
template <class XL, bool SMALL> VGK_HD void ww_merge_candidates(WwCtx<XL, SMALL>& c, int32_t& best_score, int32_t& best_diag, uint32_t& best_seq, uint32_t& best_off, uint32_t& best_node,
                                                    int32_t diag_base) {
    // (penalties are < 2^20 here or INT_MAX for "none"; diagonals within +-512 of diag_base)
    const unsigned long long none = ~0ull;
    unsigned long long key = none;
    if (c.cand_score != 0x7fffffff)
        key = ((unsigned long long)(uint32_t)c.cand_score << 32) | ((unsigned long long)(uint32_t)(c.cand_diag - diag_base + 1024) << 16) | ((unsigned long long)c.cand_leaf << 8) | c.lane;
    const unsigned long long win = c.xl->reduce_min_u64(key);
    if (win != none) {
        const uint32_t who = (uint32_t)(win & 0xffu);
        const int32_t s = (int32_t)c.xl->bcast((uint32_t)c.cand_score, who), d = (int32_t)c.xl->bcast((uint32_t)c.cand_diag, who);
        const uint32_t q = c.xl->bcast(c.cand_seq, who), o = c.xl->bcast(c.cand_off, who), n = c.xl->bcast(c.cand_node, who);
        if (s < best_score) { best_score = s; best_diag = d; best_seq = q; best_off = o; best_node = n; }
    }
    c.cand_score = 0x7fffffff;
}

template <class XL, bool SMALL> VGK_HD void ww_extend(WwCtx<XL, SMALL>& c, int32_t score, int32_t& best_score, int32_t& best_diag, uint32_t& best_seq, uint32_t& best_off, uint32_t& best_node) {
    const WPScore ps = ww_ps(c, score);
    if (!(ps.flags & 1)) return;
    const WSrc here = { score, ps.min_d, ps.max_d };
    for (int32_t diag0 = ps.min_d; diag0 <= ps.max_d;) {
        const uint32_t leaves = c.sh->leaves, n_leaves = ww_popcount(leaves);
        uint32_t n_diag;
        ++c.n_chunks;
        WwItem it; it.st = WX_DONE; it.qh = it.qt = 0; it.sp = 0; it.key_leaf = 0; it.blocked_on = 0; it.creator = false; it.pos = w_none();
        int32_t diag = diag0;
        bool have = false; uint32_t top = 0;
        if (n_leaves > 1u && ww_filtering(c, score)) {                          // (a trie of one leaf has an item on every diagonal: nothing to pack, and the sets' loads and the packing cost a chunk more than its lookups)
            if constexpr (!SMALL) n_diag = ww_chunk_of(c, diag0, ps.max_d, ww_mask_at(c, WK_MATCH, here, diag0 + (int32_t)c.lane), diag, top, have);
        } else {
            const uint32_t fit = 64u / n_leaves, left = (uint32_t)(ps.max_d - diag0 + 1);
            n_diag = fit < left ? fit : left;
            if (c.lane < n_diag * n_leaves) { have = true; diag = diag0 + (int32_t)(c.lane / n_leaves); top = ww_nth_bit(leaves, c.lane % n_leaves); }
        }
        if (have) { c.sh->queue[c.lane][0] = (uint8_t)top; it.qt = 1; it.st = WX_FETCH; }
        for (uint32_t rounds = 0;; ++rounds) {
            if (rounds > 64u * W_NODES) { c.overflow = true; c.why = 10; }
            if (it.st != WX_DONE && it.st != WX_BLOCKED) ww_extend_run(c, here, score, diag, it);
            if (ww_any_overflow(c)) return;
            const unsigned long long none = ~0ull;
            const unsigned long long key = it.st == WX_BLOCKED ? (((unsigned long long)(uint32_t)(diag - diag0) << 16) | ((unsigned long long)it.key_leaf << 8) | c.lane) : none;
            const unsigned long long first = c.xl->reduce_min_u64(key);
            if (first == none) break;                                          // everyone is done
            const uint32_t who = (uint32_t)(first & 0xffu);
            const uint32_t node = c.xl->bcast(it.blocked_on, who);
            const bool made = ww_expand_wave(c, node, who);
            c.xl->fence();
            ww_masks_begin(c, score);
            if (ww_any_overflow(c)) return;                                    // (a trie that ran out of nodes is half made: nobody may look at it)
            if (it.st == WX_BLOCKED && it.blocked_on == node) { it.st = WX_AFTER; it.creator = made; }
        }
        if (ww_any_overflow(c)) return;
        diag0 += (int32_t)n_diag;
    }
    const int32_t far = c.xl->reduce_max(c.max_distance);
    c.max_distance = far;
    ww_merge_candidates(c, best_score, best_diag, best_seq, best_off, best_node, ps.min_d);
}

// ---- next() (:1709-1786): one item per lane, nothing to wait for ----
template <class XL, bool SMALL> VGK_HD void ww_next(WwCtx<XL, SMALL>& c, int32_t score, int32_t& best_score, int32_t& best_diag, uint32_t& best_seq, uint32_t& best_off, uint32_t& best_node) {
    const WfaParams& B = c.P->base;
    int32_t lo = 32767, hi = -32768;
    auto widen = [&](int32_t s) { if (s < 0) return; const WPScore p = ww_ps(c, s); if (!(p.flags & 1)) return; if (p.min_d < lo) lo = p.min_d; if (p.max_d > hi) hi = p.max_d; };
    widen(score - B.mismatch); widen(score - B.gap_open - B.gap_extend); widen(score - B.gap_extend);
    if (lo <= hi) { --lo; ++hi; }
    int32_t alo = 32767, ahi = -32768;
    const WSrc src_mismatch = ww_src(c, score - B.mismatch), src_open = ww_src(c, score - B.gap_open - B.gap_extend), src_extend = ww_src(c, score - B.gap_extend);
    ww_clear_masks(c, score);
    for (int32_t diag0 = lo; diag0 <= hi;) {
        const uint32_t leaves = c.sh->leaves, n_leaves = ww_popcount(leaves);
        uint32_t n_diag;
        ++c.n_chunks;
        uint32_t want = W_NODES;                                               // the node this item asks to have expanded
#if VGK_WW_FINE_STATS
        c.t_fine = c.xl->clock_us();
#define WW_FINE(k) { const uint32_t t_ = c.xl->clock_us(); c.us_fine[k] += t_ - c.t_fine; c.t_fine = t_; }
#else
#define WW_FINE(k)
#endif
        int32_t diag = diag0; uint32_t leaf = 0; bool have_cand = false, have = false, filtered = false;
        if (n_leaves > 1u && ww_filtering(c, score - (B.mismatch > B.gap_open + B.gap_extend ? B.mismatch : B.gap_open + B.gap_extend))) {      // (one leaf: an item on every diagonal, nothing to pack)
            if constexpr (!SMALL) {
                const int32_t d = diag0 + (int32_t)c.lane;
                uint32_t mk[5] = {0u, 0u, 0u, 0u, 0u};
                if (d <= hi) {
                    mk[0] = ww_mask_at(c, WK_MATCH, src_mismatch, d); mk[1] = ww_mask_at(c, WK_MATCH, src_open, d - 1); mk[2] = ww_mask_at(c, WK_INS, src_extend, d - 1);
                    mk[3] = ww_mask_at(c, WK_MATCH, src_open, d + 1); mk[4] = ww_mask_at(c, WK_DEL, src_extend, d + 1);
                }
                for (int k = 0; k < 5; ++k) c.sh->src_mask[c.lane][k] = mk[k];
                n_diag = ww_chunk_of(c, diag0, hi, mk[0] | mk[1] | mk[2] | mk[3] | mk[4], diag, leaf, have);
                filtered = true;
            }
        } else {
            const uint32_t fit = 64u / n_leaves, left = (uint32_t)(hi - diag0 + 1);
            n_diag = fit < left ? fit : left;
            if (c.lane < n_diag * n_leaves) { have = true; diag = diag0 + (int32_t)(c.lane / n_leaves); leaf = ww_nth_bit(leaves, c.lane % n_leaves); }
        }
        WW_FINE(0)
        if (have) {
            const uint32_t anc = c.sh->nodes[leaf].ancestors;
            // which of the five source cells hold a point on this leaf's way to the root (all, when every item is looked at): a cell
            // that does not is not probed — the probe would walk its hash chain to the first free slot and find nothing
            uint32_t has = 31u;
            if constexpr (!SMALL) { if (filtered) { has = 0u; for (uint32_t k = 0; k < 5u; ++k) if (c.sh->src_mask[diag - diag0][k] & anc) has |= 1u << k; } }
            // (the first two slots of every source cell's probe sequence, read together: ww_probe)
            const WwProbe p_io = ww_probe(c, (has & 2u) != 0, WK_MATCH, src_open, diag - 1), p_ie = ww_probe(c, (has & 4u) != 0, WK_INS, src_extend, diag - 1),
                          p_do = ww_probe(c, (has & 8u) != 0, WK_MATCH, src_open, diag + 1), p_de = ww_probe(c, (has & 16u) != 0, WK_DEL, src_extend, diag + 1),
                          p_s = ww_probe(c, (has & 1u) != 0, WK_MATCH, src_mismatch, diag);
            WPos ins;
            { const WPos open = ww_find_probed(c, p_io, WK_MATCH, src_open, anc, leaf, diag - 1, true, false), ext = ww_find_probed(c, p_ie, WK_INS, src_extend, anc, leaf, diag - 1, true, false);
              ins = w_less(open, ext) ? ext : open; }
            bool put_ins = false, put_del = false, put_match = false;          // (the three stores go out together at the end: ww_put_begin)
            if (!ins.empty) {
                ins.seq++;
                if (w_distance(ins, diag) >= c.min_distance) { put_ins = true; if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
            }
            WPos del;
            { const WPos open = ww_find_probed(c, p_do, WK_MATCH, src_open, anc, leaf, diag + 1, false, true), ext = ww_find_probed(c, p_de, WK_DEL, src_extend, anc, leaf, diag + 1, false, true);
              del = w_less(open, ext) ? ext : open; }
            if (!del.empty) {
                ww_successor_offset(c, del);
                if (w_distance(del, diag) >= c.min_distance) { put_del = true; if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
                if (ww_wants_expansion(c, del)) want = del.cur;
            }
            WPos subst = ww_find_probed(c, p_s, WK_MATCH, src_mismatch, anc, leaf, diag, true, true);
            if (!subst.empty) { subst.seq++; ww_successor_offset(c, subst); if (want == (uint32_t)W_NODES && ww_wants_expansion(c, subst)) want = subst.cur; }
            if (w_less(subst, ins)) subst = ins;
            if (w_less(subst, del)) subst = del;
            if (!subst.empty && !c.overflow) {
                if (!c.no_to) ww_past_end(c, subst.cur, subst.off);
                if (!c.no_to && subst.off == c.sh->nodes[subst.cur].target_offset) {
                    const uint32_t gap_length = c.L - subst.seq;
                    const int32_t before = c.cand_score;
                    ww_candidate(c, score + (gap_length ? ww_gap_penalty(c, gap_length) : 0), diag, subst.seq, subst.off, subst.cur, leaf);
                    have_cand = c.cand_score != before;
                }
                if (w_distance(subst, diag) >= c.min_distance) { put_match = true; if (diag < alo) alo = diag; if (diag > ahi) ahi = diag; }
            }
            WW_FINE(1)
            const WwPut s_ins = ww_put_begin(c, put_ins, ins.cur, WK_INS, score, diag, ins.seq, ins.off), s_del = ww_put_begin(c, put_del, del.cur, WK_DEL, score, diag, del.seq, del.off),
                        s_match = ww_put_begin(c, put_match, subst.cur, WK_MATCH, score, diag, subst.seq, subst.off);
            ww_put_end(c, s_ins); ww_put_end(c, s_del); ww_put_end(c, s_match);
            WW_FINE(2)
        }
        if (ww_any_overflow(c)) return;
        // the expansions this chunk asked for, in key order (= lane order); a node is expanded for the first item that asks
        for (unsigned long long asking = c.xl->ballot(want != (uint32_t)W_NODES); asking; asking &= asking - 1) {
            const uint32_t who = (uint32_t)__builtin_ctzll(asking);
            const uint32_t node = c.xl->bcast(want, who);
            const int32_t at = (int32_t)c.xl->bcast((uint32_t)diag, who);
            if (ww_expand_wave(c, node, who)) c.sh->expanded_at[node] = at;
            c.xl->fence();
            ww_masks_begin(c, score);
            if (ww_any_overflow(c)) return;
        }
        // A candidate of an item whose leaf was expanded by an item of an EARLIER diagonal of this chunk: sequentially that diagonal's
        // items are the leaf's children (first the one made first), behind every older leaf.
        if (have_cand && c.sh->nodes[leaf].n_children && c.sh->expanded_at[leaf] < diag) c.cand_leaf = c.sh->nodes[leaf].first_child;
        diag0 += (int32_t)n_diag;
        // (a lane holds at most one candidate per chunk, and chunks come in key order: merge chunk by chunk)
        ww_merge_candidates(c, best_score, best_diag, best_seq, best_off, best_node, lo);
        WW_FINE(3)
    }
    const int32_t rlo = -c.xl->reduce_max(-alo), rhi = c.xl->reduce_max(ahi);
    if (c.lane == 0 && (c.sh->ps_flags[score] & 1)) c.sh->ps_range[score] = ((uint32_t)rlo & 0xffffu) | ((uint32_t)rhi << 16);
    c.xl->fence();
}

// ---- the sequential rest, on lane 0: penalties, trim, backtrace, output (the code of wfa_device.hpp over this kernel's tables) ----
template <class XL, bool SMALL> VGK_HD void ww_mark(WwCtx<XL, SMALL>& c, int32_t score, bool gap) {
    const uint8_t f = c.sh->ps_flags[score];
    if (!(f & 1)) { c.sh->ps_range[score] = 0; c.sh->ps_flags[score] = (uint8_t)(1 | (gap ? 2 : 0)); }
    else if (gap && !(f & 2)) c.sh->ps_flags[score] = (uint8_t)(f | 2);
}
template <class XL, bool SMALL> VGK_HD int32_t ww_next_score(WwCtx<XL, SMALL>& c, int32_t match_score) {
    const WfaParams& B = c.P->base;
    ww_mark(c, match_score + B.mismatch, false);
    if (c.sh->ps_flags[match_score] & 2) ww_mark(c, match_score + B.gap_extend, true);
    ww_mark(c, match_score + B.gap_open + B.gap_extend, true);
    int32_t s = match_score + 1;
    while (s < W_SCORES - 1 && !(c.sh->ps_flags[s] & 1)) ++s;
    return s;
}
template <class XL, bool SMALL> VGK_HD int32_t ww_alignment_score(const WwCtx<XL, SMALL>& c, int32_t score, int32_t diag, uint32_t seq, uint32_t final_insertion) {
    const int32_t target_offset = (int32_t)seq - diag;
    return (c.P->base.match * ((int32_t)(seq + final_insertion) + target_offset) - score) / 2;
}
template <class XL, bool SMALL> VGK_HD WPos ww_ins_predecessor(WwCtx<XL, SMALL>& c, uint32_t node, int32_t score, int32_t diag, int& edit) {
    const WfaParams& B = c.P->base;
    const WPos open = ww_find_pos(c, WK_MATCH, node, score - B.gap_open - B.gap_extend, diag - 1, true, false);
    const WPos ext = ww_find_pos(c, WK_INS, node, score - B.gap_extend, diag - 1, true, false);
    if (w_less(open, ext)) { edit = VGK_WFA_INSERTION; return ext; }
    edit = VGK_WFA_MATCH; return open;
}
template <class XL, bool SMALL> VGK_HD WPos ww_del_predecessor(WwCtx<XL, SMALL>& c, uint32_t node, int32_t score, int32_t diag, int& edit) {
    const WfaParams& B = c.P->base;
    const WPos open = ww_find_pos(c, WK_MATCH, node, score - B.gap_open - B.gap_extend, diag + 1, false, true);
    const WPos ext = ww_find_pos(c, WK_DEL, node, score - B.gap_extend, diag + 1, false, true);
    if (w_less(open, ext)) { edit = VGK_WFA_DELETION; return ext; }
    edit = VGK_WFA_MATCH; return open;
}
template <class XL, bool SMALL> VGK_HD WPos ww_match_predecessor(WwCtx<XL, SMALL>& c, uint32_t node, int32_t score, int32_t diag, int& edit) {
    const WPos ins = ww_find_pos(c, WK_INS, node, score, diag, false, false);
    const WPos del = ww_find_pos(c, WK_DEL, node, score, diag, false, false);
    WPos subst = ww_find_pos(c, WK_MATCH, node, score - c.P->base.mismatch, diag, false, false);
    if (!subst.empty) { subst.seq++; subst.off++; }
    if (w_less(ins, del)) {
        if (w_less(del, subst)) { edit = VGK_WFA_MISMATCH; return subst; }
        edit = VGK_WFA_DELETION; return del;
    }
    if (w_less(ins, subst)) { edit = VGK_WFA_MISMATCH; return subst; }
    edit = VGK_WFA_INSERTION; return ins;
}
template <class XL, bool SMALL> VGK_HD void ww_append_edit(WwCtx<XL, SMALL>& c, uint32_t* runs, uint32_t& n_edits, int edit, uint32_t length) {
    if (!length) return;
    if (n_edits && (runs[n_edits - 1] & 3u) == (uint32_t)edit) { runs[n_edits - 1] += length << 2; return; }
    if (n_edits >= (uint32_t)W_EDITS) { c.overflow = true; c.why = 4; return; }
    runs[n_edits++] = (length << 2) | (uint32_t)edit;
}

// One problem on one wavefront.  `slab` = this wavefront's number among the resident ones.
// -> true: the problem outgrew the small size's tables and the large size is to take it over (never from the large size itself)
template <class XL, bool SMALL> VGK_HD bool wfa_wave_problem(const WwParams& P, uint32_t i, uint32_t slab, uint32_t lane, WwShared<SMALL>& sh, XL& xl) {
    const WfaParams& B = P.base;
    const uint32_t t_begin = P.stats ? xl.clock_us() : 0u;
    const WProb pb = B.probs[i];
    vgk_wfa_result out; out.status = pb.status; out.ok = 0; out.score = 0; out.node_offset = 0; out.seq_offset = 0; out.length = 0;
    out.path_begin = 0; out.path_len = 0; out.edit_begin = 0; out.n_edits = 0;
    if (pb.status != VGK_OK || pb.from_node >= B.index.n_oriented) { if (lane == 0) B.results[i] = out; xl.fence(); return false; }
    WwCtx<XL, SMALL> c;
    c.P = &P; c.sh = &sh; c.xl = &xl; c.lane = lane;
    uint32_t own_points;                                                      // this launch's own limit (a caller's budget may lie below it)
    if constexpr (SMALL) {
        c.slot = nullptr; c.log = nullptr; c.path = nullptr; c.runs = nullptr; c.masks = nullptr; c.mask_width = 0;
        c.mask = WW_SMALL_SLOTS - 1; c.path_cap = WW_SMALL_PATH;
        own_points = P.small_points && P.small_points < (uint32_t)WW_SMALL_POINTS ? P.small_points : (uint32_t)WW_SMALL_POINTS;
    } else {
        c.slot = P.slots + (size_t)slab * P.n_slots; c.mask = P.n_slots - 1; c.log = P.logs + (size_t)slab * P.max_points;
        c.path = P.paths + (size_t)slab * P.path_cap; c.path_cap = P.path_cap; c.runs = P.edit_runs + (size_t)slab * W_EDITS;
        own_points = P.max_points;
        // the item filter needs every source wavefront of a step inside the ring of penalties it keeps
        const bool ring_holds = B.mismatch < WW_MASK_ROWS && B.gap_open + B.gap_extend < WW_MASK_ROWS && B.gap_extend > 0;
        c.masks = P.node_masks && ring_holds ? P.node_masks + (size_t)slab * WW_MASK_ROWS * 3u * P.mask_width : nullptr; c.mask_width = P.mask_width;
        if (lane == 0) { sh.filter_off = 0u; sh.masks_from = WW_NO_MASKS; }
    }
    c.seq = B.seqs + pb.seq_off; c.L = pb.seq_len;
    c.no_to = pb.to_node == VGK_WFA_NO_NODE; c.to_node = (int32_t)pb.to_node; c.to_off = pb.to_off;
    // the two positions in the index this kernel walks: with merged runs, the run each lies in and where in it the original node starts
    uint32_t from_walked = pb.from_node, from_inner = 0;
    c.to_inner = 0; c.to_orig_len = 0;
    if (P.merge.on) { const uint64_t t = P.merge.seed_map[pb.from_node]; from_walked = (uint32_t)t; from_inner = (uint32_t)(t >> 32); }
    if (!c.no_to) {
        if (pb.to_node < B.index.n_oriented) {
            c.to_orig_len = g_len(B.index, (int32_t)pb.to_node);
            if (P.merge.on) { const uint64_t t = P.merge.seed_map[pb.to_node]; c.to_node = (int32_t)(uint32_t)t; c.to_inner = (uint32_t)(t >> 32); }
        } else c.to_node = -2;                                                 // (no such node: nothing the walk meets is the target)
    }
    { const uint32_t budget = c.no_to ? B.max_points_tail : B.max_points; c.max_points = budget < own_points ? budget : own_points; }
    // no position gets further into a trie node than the sequence plus the deletions the score cap pays for (+ where the root starts)
    c.grow_cap = c.L + (uint32_t)(pb.score_bound / B.gap_extend) + 2u;
    c.cand_score = 0x7fffffff; c.cand_diag = 0; c.cand_seq = 0; c.cand_off = 0; c.cand_node = 0; c.cand_leaf = 0;
#if VGK_WW_FINE_STATS
    c.us_fine[0] = c.us_fine[1] = c.us_fine[2] = c.us_fine[3] = 0; c.t_fine = 0;
#endif
    c.max_distance = 0; c.min_distance = 0; c.overflow = false; c.why = 0; c.n_chunks = 0; c.n_steps = 0; c.n_items = 0; c.us_extend = c.us_next = c.us_score = 0;
    const int32_t top_score = pb.score_bound + B.gap_open + B.gap_extend + B.mismatch;
    for (int32_t s = (int32_t)lane; s <= top_score && s < W_SCORES; s += 64) sh.ps_flags[s] = 0;
    if (lane < (uint32_t)W_NODES) sh.expanded_at[lane] = 0;
    if (lane == 0) {
        sh.n_nodes = 0; sh.n_path = 0; sh.n_points = 0; sh.leaves = 0;
        const uint32_t root_rec = P.index.rec_off[from_walked];
        const uint32_t* rec = P.index.rec + root_rec;
        // (the root's first piece starts with the original `from` node: the trie's offsets count from there, as the node-by-node walk's do)
        WwStep root; root.state.node = (int32_t)from_walked; root.state.lo = 0; root.state.hi = (int32_t)rec[0] - 1; root.len = rec[2] - from_inner; root.seq_off = rec[3] + from_inner; root.rec = root_rec; root.inner = from_inner;
        ww_node_create(c, 0, root, 0, c.grow_cap + pb.from_off + 1); sh.n_nodes = 1;
        if (!c.overflow) ww_store(c, 0, WK_MATCH, 0, 0, 0, pb.from_off + 1);
        ww_mark(c, 0, false);
    }
    xl.fence();
    int32_t best_score = 0x7fffffff, best_diag = 0; uint32_t best_seq = 0, best_off = 0, best_node = 0;
    int32_t score = 0;
    bool failed = ww_any_overflow(c);
    const bool timed = P.stats != nullptr;
    uint32_t t_mark = timed ? xl.clock_us() : 0u;
    if (timed) c.us_score += t_mark - t_begin;                                 // (the problem's start — its record, the root's walk, the first point — counts with the bookkeeping)
    const uint32_t t_first = t_mark - t_begin;
    uint32_t t_after[3] = {0, 0, 0};
    auto lap = [&](uint32_t& into) { if (timed) { const uint32_t t = xl.clock_us(); into += t - t_mark; t_mark = t; } };
    while (!failed) {
        ww_extend(c, score, best_score, best_diag, best_seq, best_off, best_node);
        lap(c.us_extend);
        if ((failed = ww_any_overflow(c))) break;
        if (pb.distance_band < c.max_distance) c.min_distance = c.max_distance - pb.distance_band;
        if (best_score <= score) break;
        int32_t next = 0;
        if (lane == 0) next = ww_next_score(c, score);
        xl.fence();
        score = (int32_t)xl.bcast((uint32_t)next, 0);
        lap(c.us_score);
        if (score > pb.score_bound) break;
        ww_next(c, score, best_score, best_diag, best_seq, best_off, best_node);
        lap(c.us_next);
        ++c.n_steps;
        if ((failed = ww_any_overflow(c))) break;
    }
    const uint32_t t_loop_end = timed ? xl.clock_us() : 0u;
    // the rest is one chain of dependent lookups: lane 0
    const uint32_t n_points = sh.n_points < c.max_points ? sh.n_points : c.max_points;
    uint32_t* runs = c.run_buf();
    uint32_t n_edits = 0, n_chain = 0, used = 0; bool ok = false, drop_first = false;      // lane 0's until they are handed round
    if (lane == 0 && !failed) {
        c.cand_score = best_score; c.cand_diag = best_diag; c.cand_seq = best_seq; c.cand_off = best_off; c.cand_node = best_node;
        ok = true;
        uint32_t unaligned_tail = c.L - c.cand_seq;
        if (c.cand_score > pb.score_bound) {
            unaligned_tail = 0;
            if (c.no_to) {                                                     // WFATree::trim (:1849-1868), ties as in wfa_device.hpp
                c.cand_score = 0; c.cand_diag = 0; c.cand_seq = 0; c.cand_off = 0; c.cand_node = 0;
                int32_t best = 0; uint32_t best_order = 0xffffffffu;
                for (uint32_t k = 0; k < n_points; ++k) {
                    const unsigned long long s = xl.load64(c.tbl(c.log_at(k)));
                    const uint32_t key = (uint32_t)(s >> 32) - 1;
                    if (((key >> 5) & 3u) != (uint32_t)WK_MATCH) continue;
                    const uint32_t node = key & 31u; const int32_t sc = (int32_t)((key >> 7) & 1023u), dg = (int32_t)((key >> 17) & 1023u) - 512;
                    const uint32_t seq = (uint32_t)(s >> 16) & 0xffffu, off = (uint32_t)s & 0xffffu;
                    const int32_t as = ww_alignment_score(c, sc, dg, seq, 0);
                    const uint32_t order = (node << 20) | ((uint32_t)sc << 10) | (uint32_t)(dg + 512);
                    if (as > best || (as == best && best_order != 0xffffffffu && order < best_order)) {
                        best = as; best_order = order;
                        c.cand_score = sc; c.cand_diag = dg; c.cand_seq = seq; c.cand_off = off; c.cand_node = node;
                    }
                }
            } else ok = false;
        }
        bool lost = false;
        if (ok) {
            out.ok = 1; out.node_offset = pb.from_off + 1;
            out.length = c.cand_seq + unaligned_tail;
            out.score = ww_alignment_score(c, c.cand_score, c.cand_diag, c.cand_seq, unaligned_tail);
            int32_t p_score = c.cand_score, p_diag = c.cand_diag; uint32_t p_seq = c.cand_seq, p_off = c.cand_off, node = c.cand_node;
            if (unaligned_tail > 0) { ww_append_edit(c, runs, n_edits, VGK_WFA_INSERTION, c.L - c.cand_seq); p_score -= ww_gap_penalty(c, unaligned_tail); }
            int edit = VGK_WFA_MATCH;
            uint32_t steps = 0;
            while ((p_seq > 0 || p_diag != 0) && !c.overflow && !lost) {
                if (steps++ > 4u * (c.L + 2048u)) { c.overflow = true; c.why = 10; break; }
                int pe; WPos pred;
                switch (edit) {
                case VGK_WFA_MATCH:
                    pred = ww_match_predecessor(c, node, p_score, p_diag, pe);
                    if (pred.empty && (p_score != 0 || p_diag != 0)) { lost = true; break; }
                    ww_append_edit(c, runs, n_edits, VGK_WFA_MATCH, p_seq - pred.seq);
                    p_seq = pred.seq; p_off = pred.off;
                    if (!pred.empty) node = pred.cur;
                    edit = pe; break;
                case VGK_WFA_MISMATCH:
                    ww_append_edit(c, runs, n_edits, VGK_WFA_MISMATCH, 1);
                    p_seq--; ww_predecessor_offset(c, node, p_off);
                    p_score -= B.mismatch; edit = VGK_WFA_MATCH; break;
                case VGK_WFA_INSERTION:
                    pred = ww_ins_predecessor(c, node, p_score, p_diag, pe);
                    if (pred.empty) { lost = true; break; }
                    ww_append_edit(c, runs, n_edits, VGK_WFA_INSERTION, 1);
                    p_seq--;
                    p_score -= pe == VGK_WFA_INSERTION ? B.gap_extend : B.gap_open + B.gap_extend;
                    p_diag--; edit = pe; break;
                default:
                    pred = ww_del_predecessor(c, node, p_score, p_diag, pe);
                    if (pred.empty) { lost = true; break; }
                    ww_append_edit(c, runs, n_edits, VGK_WFA_DELETION, 1);
                    ww_predecessor_offset(c, node, p_off);
                    p_score -= pe == VGK_WFA_DELETION ? B.gap_extend : B.gap_open + B.gap_extend;
                    p_diag++; edit = pe; break;
                }
            }
            ok = !c.overflow && !lost;
        }
        if (timed) t_after[0] = xl.clock_us() - t_loop_end;
        if (lost) { out.status = VGK_ENOBAND; out.ok = 0; out.score = 0; out.node_offset = 0; out.length = 0; }
        if (ok) {
            // the chain of trie nodes, leaf first, where the lanes can read it (expanded_at is next()'s: done with)
            for (uint32_t x = c.cand_node;; x = sh.nodes[x].parent) { sh.expanded_at[n_chain++] = (int32_t)x; if (x == 0) break; }
            uint32_t ref_len = 0;
            for (uint32_t e = 0; e < n_edits; ++e) if ((runs[e] & 3u) != (uint32_t)VGK_WFA_INSERTION) ref_len += runs[e] >> 2;
            const uint32_t first_len = g_len(B.index, (int32_t)pb.from_node);
            drop_first = out.node_offset >= first_len;
            if (drop_first) out.node_offset = 0;
            used = out.node_offset + ref_len;
        } else if (c.overflow) failed = true;                                  // (the backtrace's edit runs: reported like any table that ran out)
    }
    // The path's ORIGINAL graph nodes and the output: the whole wavefront.  On lane 0 alone this was one chain of dependent loads through the merged-run
    // tables — two per original — and 12 % of the kernel's wavefront time on the chr22-scale graph (profiles/r06/NOTES.md §4); now a path entry (a piece of a
    // merged run) is two trips: its run's bounds, then every lane its own original's stretch of `ocol`.
    if (xl.bcast(lane == 0 && ok ? 1u : 0u, 0) != 0u) {
        n_edits = xl.bcast(n_edits, 0); n_chain = xl.bcast(n_chain, 0); used = xl.bcast(used, 0); drop_first = xl.bcast(drop_first ? 1u : 0u, 0) != 0u;
        const uint32_t node_offset = xl.bcast(out.node_offset, 0);
        xl.fence();                                                            // (the chain, the edit runs: lane 0's, read by everyone)
        // the ORIGINAL graph nodes along the chain, root first (a path entry is one of them, or a piece of a merged run: its originals in a row), one after
        // the other on the calling lane; visit(oriented node, length) -> false: enough
        auto originals = [&](auto&& visit) {
            bool first = true;
            for (uint32_t k = n_chain; k-- > 0;) {
                const WwNode& n = sh.nodes[(uint32_t)sh.expanded_at[k]];
                for (uint32_t j = n.path_head; j != W_NIL; j = c.pth(j).next) {
                    const WwPath e = c.pth(j);
                    if (P.merge.on) {
                        WwOriginals it = ww_originals(P.merge, (uint32_t)e.node, (uint32_t)e.inner);
                        for (uint32_t got = 0; got < e.len && !it.done(); it.step()) {
                            const uint32_t gl = it.length(); got += gl;
                            const bool skip = first && drop_first; first = false;
                            if (!skip && !visit(it.oriented(), gl)) return;
                        }
                    } else {
                        const bool skip = first && drop_first; first = false;
                        if (!skip && !visit((uint32_t)e.node, (uint32_t)e.len)) return;
                    }
                }
            }
        };
        // The originals it passes are kept where the penalties' table was (done with after the loop; W_SCORES words of this wavefront's LDS) and copied out
        // once their place in `paths` is known.  A path of more nodes than the table holds is walked a second time, by lane 0.
        uint32_t kept = 0, at = 0, last_start = 0, last_len = 0;              // (the same on every lane)
        uint32_t* const walked = sh.ps_range;
        if (P.merge.on) {
            bool skip_pending = drop_first, stop = false;
            for (uint32_t k = n_chain; k-- > 0 && !stop;) {
                const WwNode& n = sh.nodes[(uint32_t)sh.expanded_at[k]];
                for (uint32_t j = n.path_head; j != W_NIL && !stop; j = c.pth(j).next) {
                    const WwPath e = c.pth(j);
                    const uint32_t m = (uint32_t)e.node >> 1; const bool rev = ((uint32_t)e.node & 1u) != 0;
                    const uint32_t v0 = P.merge.run_first[m], v1 = P.merge.run_first[m + 1], nq = v1 - v0;
                    const uint32_t edge = P.merge.ocol[rev ? v1 : v0];         // the run's first base on this strand's side
                    bool have_s0 = false, entry_done = false; uint32_t s0 = 0;
                    for (uint32_t base = 0; base < nq && !entry_done && !stop; base += 64u) {
                        const uint32_t q = base + lane; const bool valid = q < nq;
                        const uint32_t v = valid ? (rev ? v1 - 1u - q : v0 + q) : v0;
                        const uint32_t a = P.merge.ocol[v], b = P.merge.ocol[v + 1];
                        const uint32_t start = rev ? edge - b : a - edge, gl = b - a;      // bases of the run before this original, on this strand
                        const bool from = valid && start >= (uint32_t)e.inner;
                        if (!have_s0) {
                            const unsigned long long fm = xl.ballot(from);
                            if (fm) { have_s0 = true; s0 = xl.bcast(start, (uint32_t)__builtin_ctzll(fm)); }
                        }
                        const bool in = from && have_s0 && start - s0 < (uint32_t)e.len;
                        if (xl.ballot(from && !in) != 0ull) entry_done = true;   // (an original that starts behind the piece: the piece is through)
                        unsigned long long in_mask = xl.ballot(in);
                        if (skip_pending && in_mask) { in_mask &= in_mask - 1ull; skip_pending = false; }
                        if (!in_mask) continue;
                        const bool mine = ((in_mask >> lane) & 1ull) != 0ull;
                        const uint32_t rank = (uint32_t)__builtin_popcountll(in_mask & ((1ull << lane) - 1ull));
                        const uint32_t first_start = xl.bcast(start, (uint32_t)__builtin_ctzll(in_mask));
                        const uint32_t pos = at + (start - first_start);
                        const bool keep = mine && (kept + rank == 0u || pos < used);
                        const unsigned long long keep_mask = xl.ballot(keep);
                        if (keep && kept + rank < (uint32_t)W_SCORES) walked[kept + rank] = 2u * v + (rev ? 1u : 0u);
                        if (keep_mask) {
                            const uint32_t last = 63u - (uint32_t)__builtin_clzll(keep_mask);
                            last_start = xl.bcast(pos, last); last_len = xl.bcast(gl, last);
                            at = last_start + last_len; kept += (uint32_t)__builtin_popcountll(keep_mask);
                        }
                        if (keep_mask != in_mask) stop = true;                 // (everything from here on starts behind the alignment's last base)
                    }
                }
            }
        } else {
            if (lane == 0) originals([&](uint32_t o, uint32_t gl) {
                if (kept != 0 && at >= used) return false;
                if (kept < (uint32_t)W_SCORES) walked[kept] = o;
                ++kept; last_start = at; last_len = gl; at += gl;
                return true;
            });
            kept = xl.bcast(kept, 0); last_start = xl.bcast(last_start, 0); last_len = xl.bcast(last_len, 0);
        }
        xl.fence_lds();                                                        // (walked[]: written by the lanes, copied out by others)
        const bool walked_all = kept <= (uint32_t)W_SCORES;
        if (timed && lane == 0) t_after[1] = xl.clock_us() - t_loop_end - t_after[0];
        if (kept == 1 && used == node_offset) kept = 0;
        // room in the dense outputs: lane 0 asks for the path's, lane 1 for the edits' (one trip for both)
        unsigned long long got = 0;
        if (lane < 2u) got = g_bump(B.counters + lane, lane == 0 ? kept : n_edits);
        const unsigned long long p0 = (unsigned long long)xl.bcast((uint32_t)got, 0) | ((unsigned long long)xl.bcast((uint32_t)(got >> 32), 0) << 32),
                                 e0 = (unsigned long long)xl.bcast((uint32_t)got, 1) | ((unsigned long long)xl.bcast((uint32_t)(got >> 32), 1) << 32);
        if (p0 + kept > B.caps[0] || e0 + n_edits > B.caps[1]) { if (lane == 0) { out.status = VGK_EOPS; out.ok = 0; } }
        else {
            const bool flip = pb.mode == VGK_WFA_PREFIX;
            if (walked_all) { for (uint32_t w = lane; w < kept; w += 64u) { const uint32_t o = walked[w]; B.paths[p0 + (flip ? kept - 1 - w : w)] = flip ? (o ^ 1u) : o; } }
            else if (lane == 0) {
                uint32_t w = 0;
                originals([&](uint32_t o, uint32_t) {
                    if (w >= kept) return false;
                    B.paths[p0 + (flip ? kept - 1 - w : w)] = flip ? (o ^ 1u) : o;
                    ++w;
                    return true;
                });
            }
            for (uint32_t e = lane; e < n_edits; e += 64u) B.edits[e0 + e] = runs[flip ? e : n_edits - 1 - e];
            if (lane == 0) {
                out.path_begin = (uint32_t)p0; out.path_len = kept; out.edit_begin = (uint32_t)e0; out.n_edits = n_edits;
                if (pb.mode != VGK_WFA_CONNECT && n_edits && out.length == c.L) {
                    const uint32_t last = runs[0] & 3u;
                    if (last == (uint32_t)VGK_WFA_MATCH || last == (uint32_t)VGK_WFA_MISMATCH) out.score += B.bonus;
                }
                if (flip) {
                    out.seq_offset = c.L - out.seq_offset - out.length;
                    if (kept) out.node_offset = last_len - (used - last_start);
                }
            }
        }
        if (timed && lane == 0) t_after[2] = xl.clock_us() - t_loop_end - t_after[0] - t_after[1];
    }
    failed = xl.ballot(failed) != 0ull;
    // tables of the small size were not enough: the large size takes the problem over — unless it was the caller's own point budget that
    // ran out —; what outgrows the large size is declined
    bool retry = false;
    if (lane == 0) {
        if (failed) {
            retry = SMALL && ((c.why == 1 && c.max_points == own_points) || c.why == 3 || c.why == 7);
            if (retry && P.n_declined) g_bump(P.n_declined, 1);
            out.status = VGK_ETOOBIG; out.ok = 0; out.score = c.why; out.node_offset = 0; out.length = 0;
        }
        if (!retry) B.results[i] = out;
        if (P.stats && !retry) {
            uint32_t* st = P.stats + WW_STAT_WORDS * (size_t)i;
            st[0] = sh.n_points; st[1] = c.n_steps; st[2] = c.n_chunks; st[3] = sh.n_nodes | (c.n_items << 8);      // (items: those the filter let through; 0 when every item is looked at)
            st[4] = c.us_extend; st[5] = c.us_next; st[6] = c.us_score; st[7] = xl.clock_us() - t_loop_end;
#if VGK_WW_FINE_STATS
            st[8] = c.us_fine[0]; st[9] = c.us_fine[1]; st[10] = c.us_fine[2]; st[11] = c.us_fine[3];
#else
            st[8] = t_after[0]; st[9] = t_after[1]; st[10] = t_after[2]; st[11] = t_first;
#endif                           // after the loop: backtrace, the path's originals, room + output; before the loop
        }
    }
    retry = xl.bcast(retry ? 1u : 0u, 0) != 0u;
    // leave the table all-zero: the touched slots from the log, or everything when the log ran over
    xl.fence();
    if (sh.n_points > c.max_points) { for (uint32_t k = lane; k <= c.mask; k += 64) *c.tbl(k) = 0; }
    else for (uint32_t k = lane; k < n_points; k += 64) *c.tbl(c.log_at(k)) = 0;
    xl.fence();
    return retry;
}

// One resident wavefront: problems are handed out one at a time; each starts in the small size (everything in LDS) and, if it outgrows
// that, is run again at once by the same wavefront in the large size (its slab in HBM).  The two sizes share the wavefront's LDS.
union WwSharedBoth { WwShared<true> small; WwShared<false> large; };
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ uint32_t ww_peek32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ unsigned long long ww_peek64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void ww_nap() { __builtin_amdgcn_s_sleep(32); }
static __device__ __forceinline__ void ww_acquire() { __threadfence(); }
#else
static inline uint32_t ww_peek32(const uint32_t* p) { return *p; }
static inline unsigned long long ww_peek64(const unsigned long long* p) { return *p; }
static inline void ww_nap() {}
static inline void ww_acquire() {}
#endif
template <class XL> VGK_HD void wfa_wave(const WwParams& P, uint32_t slab, uint32_t lane, WwSharedBoth& sh, XL& xl) {
    bool table_clean = false;
    const uint32_t n_todo = P.n_todo_dev ? (uint32_t)*P.n_todo_dev : P.n_todo;
    // (a list that is complete when the launch begins: the NEXT hand-out is asked for when a problem starts — the counter's trip runs beside the problem's
    //  first loads — and looked at when it ends; a list that is still being written is asked for entry by entry, as before)
    const bool ahead = !P.producers_done;
    uint32_t k_ahead = 0;
    if (ahead) { if (lane == 0) k_ahead = (uint32_t)g_bump(P.hand_out ? P.hand_out : P.base.counters + 2, 1); k_ahead = xl.bcast(k_ahead, 0); }
    for (;;) {
        uint32_t k = 0;
        if (ahead) k = k_ahead;
        else { if (lane == 0) k = (uint32_t)g_bump(P.hand_out ? P.hand_out : P.base.counters + 2, 1); k = xl.bcast(k, 0); }
        uint32_t i = 0;
        if (P.producers_done) {
            // the k-th entry of a list that is being written: there already, or still to come, or never (every producer has finished and
            // the list ends before it).  Lane 0 waits, napping between looks; the producers never wait for this kernel.
            uint32_t verdict = 0;                                            // 1 = take entry i, 2 = the list has ended
            if (lane == 0) {
                for (;;) {
                    if ((unsigned long long)k < ww_peek64(P.n_todo_dev)) { verdict = 1; break; }
                    if (ww_peek32(P.producers_done) >= P.n_producers) { ww_acquire(); verdict = (unsigned long long)k < ww_peek64(P.n_todo_dev) ? 1u : 2u; break; }
                    ww_nap();
                }
                if (verdict == 1) while ((i = ww_peek32(P.todo + k)) == 0xffffffffu) ww_nap();
            }
            verdict = xl.bcast(verdict, 0); i = xl.bcast(i, 0);
            if (verdict == 2) break;
        } else {
            if (k >= n_todo) break;
            i = P.todo[k];
            if (lane == 0) k_ahead = (uint32_t)g_bump(P.hand_out ? P.hand_out : P.base.counters + 2, 1);      // (read after the problem, below)
        }
        if (!table_clean) {                                                  // (LDS comes up with whatever was there; the large size used it for its own lists)
            for (uint32_t j = lane; j < (uint32_t)WW_SMALL_SLOTS; j += 64) sh.small.slot[j] = 0;
            xl.fence(); table_clean = true;
        }
        if (wfa_wave_problem<XL, true>(P, i, slab, lane, sh.small, xl)) { wfa_wave_problem<XL, false>(P, i, slab, lane, sh.large, xl); table_clean = false; }
        if (ahead) k_ahead = xl.bcast(k_ahead, 0);
    }
}

}  // namespace vgk
