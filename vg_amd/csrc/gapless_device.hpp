// gapless_device.hpp — haplotype-consistent gapless seed extension, one thread per read
// (replaces GaplessExtender::extend of the reference's src/gbwt_extender.cpp:533-737 with its helpers
// match_initial / match_forward / match_backward :213-296, set_score :201-209, handle_full_length :301-329,
// remove_duplicates :332-365, find_mismatches :368-387, trim_mismatches :421-529).
//
// The haplotype index is resident in HBM in uncompressed form (per oriented node: visit count, outgoing edges in
// node order with the rank offset into the successor's record, and per visit the edge it leaves through), which
// turns GBWT's LF-mapping into a short scan over at most #haplotypes entries — a read-only, cache-friendly
// gather instead of the CPU's compressed-record decode.  Everything a read needs while it is being extended
// (the best-first queue, the tree of partial extensions, the per-seed winners) lives in a fixed scratch slab per
// thread; finished extension sets are packed behind each other with atomic bumps.
//
// The same code runs on the CPU under tests/emu (test infrastructure only).
#pragma once
#include <stdint.h>
#include "pk16.hpp"
#include "../../include/vgk.h"

namespace vgk {

// One record per oriented node, padded to a multiple of 16 words so that a hop touches one or two cache lines:
//   word 0 visit count | 1 outgoing edges | 2 node length | 3 offset of the node's bases in `seq`
//   then per edge (ascending successor), four words: successor (or -1: a thread ends) | where this node's visits start in the
//     successor's record (low 16 bits) and the successor's length (high 16) | offset of the successor's bases in `seq` | offset of
//     the successor's record — everything a hop needs to start comparing bases and to fetch the next record without first
//     reading the successor's own record through rec_off (two dependent loads saved per hop)
//   then per visit: the number of the edge it leaves through, one byte each (a record of 16 visits and 2 edges is one 64-byte line)
//   — or, when word 1 has bit 31 set, the same sequence RUN-LENGTH encoded like a GBWT record body: one word per run, edge number in
//   the low byte, run length above it.  Thousands of haplotypes that share a path leave a node through the same edge in long runs
//   (the visits are sorted by where they came from), so a record of 5000 visits is typically a handful of words instead of 5 KB, and
//   extending a search state is a walk over those runs.  The builder picks whichever form is smaller per record.
struct GIndex {                       // device pointers
    uint32_t n_oriented;
    const uint32_t* rec_off;          // per oriented node, in words
    const uint32_t* rec;
    uint32_t        strand_shift;     // offset of a node's reverse-complement bases = offset of its forward bases + strand_shift
    const char*     seq;              // forward strands, then reverse complements (8 bytes of padding at either end)
    uint32_t        max_node_len, max_visits;      // over the whole index: what the compact in-LDS entries of the fast kernel can hold
    // per oriented node: offset of its bases in `seq` (low word) | its length (high word).  What a walk along a finished path needs of a
    // node — the set rules' mismatch positions, overlaps and trims; the tail forests; WFA — in ONE load from a table whose neighbours are
    // the path's next nodes, instead of rec_off[node] and then the record's header (two dependent lines per node: 4.8 of the rules kernel's
    // 7.5 kB fetched per read went there)
    const uint64_t* node_tab;
};
VGK_HD uint32_t g_seq_off(const GIndex& h, uint32_t o) { return (uint32_t)h.node_tab[o]; }
VGK_HD uint32_t g_seq_of(const GIndex& h, uint32_t o, uint32_t& len) { const uint64_t t = h.node_tab[o]; len = (uint32_t)(t >> 32); return (uint32_t)t; }
VGK_HD const uint32_t* g_rec(const GIndex& h, uint32_t o) { return h.rec + h.rec_off[o]; }
VGK_HD uint32_t g_len(const GIndex& h, int32_t o) { return (uint32_t)(h.node_tab[(uint32_t)o] >> 32); }
VGK_HD int32_t  ge_to(const uint32_t* rec, uint32_t e) { return (int32_t)rec[4 + 4 * e]; }
VGK_HD uint32_t ge_base(const uint32_t* rec, uint32_t e) { return rec[5 + 4 * e] & 0xffffu; }
VGK_HD uint32_t ge_len(const uint32_t* rec, uint32_t e) { return rec[5 + 4 * e] >> 16; }
VGK_HD uint32_t ge_seq(const uint32_t* rec, uint32_t e) { return rec[6 + 4 * e]; }
VGK_HD uint32_t ge_rec(const uint32_t* rec, uint32_t e) { return rec[7 + 4 * e]; }
VGK_HD uint32_t g_ne(const uint32_t* rec) { return rec[1] & 0xffffu; }          // outgoing edges
VGK_HD bool g_rle(const uint32_t* rec) { return (rec[1] >> 31) != 0; }         // the visit body is run-length encoded
VGK_HD const uint32_t* g_visits(const uint32_t* rec) { return rec + 4 + 4 * g_ne(rec); }
struct GProb { uint32_t read_off, read_len, seed_off, n_seeds, max_mm, flags; double overlap; };

// Unary runs merged at index build (gapless_api.cpp merge_unary_runs): the search and the set rules run on an index whose nodes are maximal
// runs of CONSECUTIVE nodes v, v + 1, ... in which every visit of a node leaves through the one edge to the next and every visit of the next
// arrives that way — so a search state's ranges map through unchanged — of at most 255 bases together (the fast kernel's 8-bit offsets).  A
// read of 150 bases over nodes of at most 32 then crosses one or two records per direction instead of five to ten dependent hops; what the
// reference does per problem for WFA (WFANode: unary paths up to 1 024 bp, src/gbwt_extender.cpp:1431-1487) is done once per index.  Seeds
// come in and extension sets go out in ORIGINAL nodes: seed_map translates a seed on its way in, the emit stage expands a merged path into the
// original nodes its aligned interval touches.  on == 0: the index is the original one, nothing is translated.
struct GMerge {
    uint32_t on, n_orig_oriented;
    const uint64_t* seed_map;        // per ORIGINAL oriented node: merged oriented node (low word) | offset of its first base in that node (high word)
    const uint32_t* run_first;       // per merged node m: first original node of its run (run_first[m + 1] - 1 = the last)
    const uint32_t* ocol;            // per original node v: bases of the nodes before it (ocol[n_nodes] = all)
};

struct GState { int32_t fn, flo, fhi, bn, blo, bhi; };      // forward / backward strand: node, visit range [lo, hi]
VGK_HD bool gs_empty(const GState& s) { return s.flo > s.fhi; }
VGK_HD uint32_t gs_size(const GState& s) { return s.flo > s.fhi ? 0u : (uint32_t)(s.fhi - s.flo + 1); }
VGK_HD GState gs_flip(const GState& s) { GState r = { s.bn, s.blo, s.bhi, s.fn, s.flo, s.fhi }; return r; }
VGK_HD GState gs_find(const GIndex& h, int32_t node) {
    GState s = { node, 0, (int32_t)g_rec(h, (uint32_t)node)[0] - 1, node ^ 1, 0, (int32_t)g_rec(h, (uint32_t)node ^ 1u)[0] - 1 };
    return s;
}
VGK_HD int32_t g_rkey(int32_t x) { return x < 0 ? -1 : (x ^ 1); }
VGK_HD uint32_t g_body(const uint32_t* body, uint32_t i) { return (body[i >> 2] >> (8 * (i & 3u))) & 0xffu; }
// One pass over the visits [0, hi] of a record with at most four edges: how many leave through each edge before the range
// [lo, hi] and inside it, 16 bits per edge (a node has at most 65 535 visits).  Every extension of the state follows from these.
struct GCounts { uint64_t before, inside; };
VGK_HD GCounts g_counts(const uint32_t* rec, int32_t lo, int32_t hi) {
    const uint32_t* body = g_visits(rec);
    GCounts c = { 0, 0 };
    if (g_rle(rec)) {
        int32_t pos = 0;
        for (uint32_t k = 0; pos <= hi; ++k) {
            const uint32_t run = body[k]; const int32_t len = (int32_t)(run >> 8), end = pos + len;      // visits [pos, end)
            const int32_t b = (end < lo ? end : lo) - pos, last = end - 1 < hi ? end - 1 : hi, first = pos > lo ? pos : lo;
            if (b > 0) c.before += (uint64_t)b << (16 * (run & 3u));
            if (last >= first) c.inside += (uint64_t)(last - first + 1) << (16 * (run & 3u));
            pos = end;
        }
        return c;
    }
    for (int32_t i = 0; i <= hi; i += 4) {
        uint32_t w = body[i >> 2];
        for (int32_t k = 0; k < 4 && i + k <= hi; ++k, w >>= 8) {
            const uint64_t one = 1ull << (16 * (w & 3u));
            if (i + k < lo) c.before += one; else c.inside += one;
        }
    }
    return c;
}
VGK_HD uint32_t g_count_of(uint64_t packed, uint32_t e) { return (uint32_t)(packed >> (16 * e)) & 0xffffu; }
// bdExtendForward through edge number e of the forward node's record, from the counts
VGK_HD GState gs_extend_counted(const uint32_t* rec, const GState& s, uint32_t e, const GCounts& cn) {
    const int32_t to = ge_to(rec, e);
    GState r = s; r.fn = to;
    const int32_t inside = (int32_t)g_count_of(cn.inside, e);
    if (!inside) { r.flo = 0; r.fhi = -1; r.bhi = r.blo - 1; return r; }
    int32_t rev_off = 0;
    for (uint32_t x = 0; x < g_ne(rec); ++x) if (x != e && g_rkey(ge_to(rec, x)) < g_rkey(to)) rev_off += (int32_t)g_count_of(cn.inside, x);
    r.flo = (int32_t)ge_base(rec, e) + (int32_t)g_count_of(cn.before, e); r.fhi = r.flo + inside - 1;
    r.blo = s.blo + rev_off; r.bhi = r.blo + inside - 1;
    return r;
}
// A record's words are read where they are needed; an edge is one 16-byte load.  (Fetching the whole first 64-byte line of the record up
// front, four 16-byte loads issued together, was measured: 7.8 ms against 7.0 ms for the search kernel — DESIGN.md §11.)
struct alignas(16) GQuad { uint32_t x, y, z, w; };
VGK_HD GQuad g_quad(const uint32_t* p) { GQuad q; __builtin_memcpy(&q, __builtin_assume_aligned(p, 16), 16); return q; }
struct GRecMem {
    const uint32_t* rec;
    VGK_HD uint32_t ne() const { return g_ne(rec); }
    VGK_HD bool rle() const { return g_rle(rec); }
    VGK_HD GQuad edge(uint32_t e) const { return g_quad(rec + 4 + 4 * e); }                           // one 16-byte load
    VGK_HD int32_t to(uint32_t e) const { return ge_to(rec, e); }
};
VGK_HD GCounts g_counts(const GRecMem& v, int32_t lo, int32_t hi) { return g_counts(v.rec, lo, hi); }
template <class V> VGK_HD GState gs_extend_counted(const V& v, const GState& s, uint32_t e, const GQuad& ed, const GCounts& cn) {
    const int32_t to = (int32_t)ed.x;
    GState r = s; r.fn = to;
    const int32_t inside = (int32_t)g_count_of(cn.inside, e);
    if (!inside) { r.flo = 0; r.fhi = -1; r.bhi = r.blo - 1; return r; }
    int32_t rev_off = 0;
    const uint32_t ne = v.ne();
    for (uint32_t x = 0; x < ne; ++x) if (x != e && g_rkey(v.to(x)) < g_rkey(to)) rev_off += (int32_t)g_count_of(cn.inside, x);
    r.flo = (int32_t)(ed.y & 0xffffu) + (int32_t)g_count_of(cn.before, e); r.fhi = r.flo + inside - 1;
    r.blo = s.blo + rev_off; r.bhi = r.blo + inside - 1;
    return r;
}
// bdExtendForward: follow the visits of the forward range that leave through `to`
VGK_HD GState gs_extend(const GIndex& h, const GState& s, int32_t to) {
    const uint32_t o = (uint32_t)s.fn;
    const uint32_t* rec = g_rec(h, o);
    const uint32_t ne = g_ne(rec);
    const uint32_t* body = g_visits(rec);
    uint32_t e = 0; while (e < ne && ge_to(rec, e) != to) ++e;
    GState r = s; r.fn = to;
    if (e == ne || gs_empty(s)) { r.flo = 0; r.fhi = -1; r.bhi = r.blo - 1; return r; }
    if (ne <= 4) return gs_extend_counted(rec, s, e, g_counts(rec, s.flo, s.fhi));
    int32_t before = 0, inside = 0, rev_off = 0;
    if (g_rle(rec)) {
        int32_t pos = 0;
        for (uint32_t k = 0; pos <= s.fhi; ++k) {
            const uint32_t run = body[k], b = run & 0xffu; const int32_t end = pos + (int32_t)(run >> 8);
            const int32_t nb = (end < s.flo ? end : s.flo) - pos, last = end - 1 < s.fhi ? end - 1 : s.fhi, first = pos > s.flo ? pos : s.flo;
            const int32_t in = last >= first ? last - first + 1 : 0;
            if (b == e) { if (nb > 0) before += nb; inside += in; }
            else if (in && g_rkey(ge_to(rec, b)) < g_rkey(to)) rev_off += in;
            pos = end;
        }
    } else
    for (int32_t i = 0; i <= s.fhi; ++i) {
        const uint32_t b = g_body(body, (uint32_t)i);
        if (b == e) { if (i < s.flo) ++before; else ++inside; }
        else if (i >= s.flo && g_rkey(ge_to(rec, b)) < g_rkey(to)) ++rev_off;
    }
    r.flo = (int32_t)ge_base(rec, e) + before; r.fhi = r.flo + inside - 1;
    r.blo = s.blo + rev_off; r.bhi = r.blo + inside - 1;
    return r;
}

// ---- per-thread scratch ----
constexpr int G_POOL = 160;          // partial extensions alive per seed (tree nodes + queue)
constexpr int G_SEEDS = 64;          // seeds per cluster the engine takes
constexpr int G_PATH = 48;           // nodes per extension
constexpr int G_MISM = 48;           // mismatches per extension

struct GEntry {                      // a partial extension; its path is the chain of `parent`s
    int32_t  parent, node;           // node added by this step (or -1: same path as the parent)
    uint32_t offset, r0, r1;
    int32_t  score;
    uint32_t internal, old, number;
    GState   state;
    uint8_t  front;                  // the node was added in front of the path
    uint8_t  left_full, right_full, left_max, right_max, pad[3];
    uint32_t frec, brec;             // offsets of the records of state.fn / state.bn when known (they come with the edge that was followed;
};                                   // G_NO_REC after a reload from the pool, which does not keep them)
constexpr uint32_t G_NO_REC = 0xffffffffu;
struct GExt {                        // a finished per-seed winner
    uint32_t offset, r0, r1, internal;
    int32_t  score;
    GState   state;
    uint8_t  left_full, right_full, pad[2];
    uint32_t path_len, n_mism;
    int32_t  path[G_PATH];
};                                    // mismatch positions are recomputed where they are needed: they would double the slab
// What a search keeps in registers between steps — the held-back candidate and the best finished entry so far — is kept in packed
// words: as plain GEntry copies (26 registers each, the byte flags one register apiece) they pushed the kernel past its 128
// registers and into scratch memory.  Same limits as the pool's packed entries: 16-bit read and node offsets and visit ranks.
struct GLean { uint32_t rr, oi, of, pn, fr, br, frec, brec; int32_t score, node, fn, bn; };      // 12 words
struct GBest { uint32_t rr, oi, fr, br, full; int32_t score, fn, bn; };                             // 8 words: what the winner keeps
VGK_HD uint32_t g_range(int32_t lo, int32_t hi) { return lo > hi ? 1u : ((uint32_t)lo | ((uint32_t)hi << 16)); }      // an empty range is stored as [1, 0]
VGK_HD void g_unrange(uint32_t w, int32_t& lo, int32_t& hi) { const uint32_t a = w & 0xffffu, b = w >> 16; const bool e = a > b; lo = e ? 0 : (int32_t)a; hi = e ? -1 : (int32_t)b; }
VGK_HD GLean g_lean(const GEntry& e) {
    GLean l;
    l.rr = e.r0 | (e.r1 << 16); l.oi = e.offset | (e.internal << 16);
    l.of = e.old | ((uint32_t)(e.front | (e.left_full << 1) | (e.right_full << 2) | (e.left_max << 3) | (e.right_max << 4)) << 16);
    l.pn = ((uint32_t)e.parent & 0xffffu) | (e.number << 16);
    l.fr = g_range(e.state.flo, e.state.fhi); l.br = g_range(e.state.blo, e.state.bhi); l.frec = e.frec; l.brec = e.brec;
    l.score = e.score; l.node = e.node; l.fn = e.state.fn; l.bn = e.state.bn;
    return l;
}
VGK_HD GEntry g_fat(const GLean& l) {
    GEntry e;
    e.r0 = l.rr & 0xffffu; e.r1 = l.rr >> 16; e.offset = l.oi & 0xffffu; e.internal = l.oi >> 16; e.old = l.of & 0xffffu;
    const uint32_t fl = l.of >> 16;
    e.front = fl & 1; e.left_full = (fl >> 1) & 1; e.right_full = (fl >> 2) & 1; e.left_max = (fl >> 3) & 1; e.right_max = (fl >> 4) & 1; e.pad[0] = e.pad[1] = e.pad[2] = 0;
    e.parent = (int32_t)(int16_t)(uint16_t)(l.pn & 0xffffu); e.number = l.pn >> 16;
    g_unrange(l.fr, e.state.flo, e.state.fhi); g_unrange(l.br, e.state.blo, e.state.bhi); e.frec = l.frec; e.brec = l.brec;
    e.score = l.score; e.node = l.node; e.state.fn = l.fn; e.state.bn = l.bn;
    return e;
}
VGK_HD GBest g_best(const GEntry& e) {
    GBest b;
    b.rr = e.r0 | (e.r1 << 16); b.oi = e.offset | (e.internal << 16); b.fr = g_range(e.state.flo, e.state.fhi); b.br = g_range(e.state.blo, e.state.bhi);
    b.full = (uint32_t)e.left_full | ((uint32_t)e.right_full << 1); b.score = e.score; b.fn = e.state.fn; b.bn = e.state.bn;
    return b;
}
// what the pool stores of an entry: 40 bytes (reads and nodes up to 65 535 bases, up to 65 535 visits per node)
struct GPacked {
    int16_t  parent; uint16_t number;
    int32_t  node;
    uint16_t offset, r0, r1, internal, old;
    uint8_t  flags, pad;                  // front | left_full << 1 | right_full << 2 | left_max << 3 | right_max << 4
    int32_t  score;
    int32_t  fn, bn;
    uint16_t flo, fhi, blo, bhi;          // an empty range is stored as [1, 0]
};
VGK_HD GPacked g_pack(const GEntry& e) {
    GPacked p;
    p.parent = (int16_t)e.parent; p.number = (uint16_t)e.number; p.node = e.node;
    p.offset = (uint16_t)e.offset; p.r0 = (uint16_t)e.r0; p.r1 = (uint16_t)e.r1; p.internal = (uint16_t)e.internal; p.old = (uint16_t)e.old;
    p.flags = (uint8_t)(e.front | (e.left_full << 1) | (e.right_full << 2) | (e.left_max << 3) | (e.right_max << 4)); p.pad = 0;
    p.score = e.score; p.fn = e.state.fn; p.bn = e.state.bn;
    const bool fe = e.state.flo > e.state.fhi, be = e.state.blo > e.state.bhi;
    p.flo = fe ? 1 : (uint16_t)e.state.flo; p.fhi = fe ? 0 : (uint16_t)e.state.fhi;
    p.blo = be ? 1 : (uint16_t)e.state.blo; p.bhi = be ? 0 : (uint16_t)e.state.bhi;
    return p;
}
VGK_HD GEntry g_unpack(const GPacked& p) {
    GEntry e;
    e.parent = p.parent; e.number = p.number; e.node = p.node;
    e.offset = p.offset; e.r0 = p.r0; e.r1 = p.r1; e.internal = p.internal; e.old = p.old;
    e.front = p.flags & 1; e.left_full = (p.flags >> 1) & 1; e.right_full = (p.flags >> 2) & 1; e.left_max = (p.flags >> 3) & 1; e.right_max = (p.flags >> 4) & 1;
    e.pad[0] = e.pad[1] = e.pad[2] = 0; e.frec = e.brec = G_NO_REC;
    e.score = p.score;
    e.state.fn = p.fn; e.state.bn = p.bn;
    const bool fe = p.flo > p.fhi, be = p.blo > p.bhi;           // only a node no haplotype visits has empty ranges, and those are [0, -1]
    e.state.flo = fe ? 0 : p.flo; e.state.fhi = fe ? -1 : (int32_t)p.fhi; e.state.blo = be ? 0 : p.blo; e.state.bhi = be ? -1 : (int32_t)p.bhi;
    return e;
}
// Per-thread scratch in two slabs: the hot one (queue keys, the pool of the seed being extended, the first few per-seed winners) stays
// small, because the kernel's speed follows the address spread of what the resident threads touch (measured: 47 M reads/s with a
// 15.9 KB slab, 38 M with 23.8 KB); winners beyond G_HOT — rare: most clusters resolve into one or two extensions — go to a cold slab.
constexpr int G_HOT = 8;
// What the path of an entry needs, for every entry: 8 bytes.  The 40-byte packed entry is written only when an entry actually
// waits in the queue; on a non-branching stretch every new entry is the held-back candidate and is popped from registers.
struct GLink { int32_t node; int16_t parent; uint8_t front, pad; };
struct GScratch { uint64_t heap[G_POOL]; GPacked pool[G_POOL]; GLink link[G_POOL]; GExt res[G_HOT]; uint8_t order[G_SEEDS]; int32_t diag[G_PATH]; };      // diag[]: flat form, see gx_diagonals
//      // order[]: the permutation the set rules sort
struct GCold { GExt res[G_SEEDS - G_HOT]; };
struct GRes {                          // the G_SEEDS winners of a read as one array
    GExt* hot; GExt* cold;
    VGK_HD GExt& operator[](uint32_t i) const { return i < (uint32_t)G_HOT ? hot[i] : cold[i - G_HOT]; }
};

struct GaplessParams {
    GIndex index;                     // what the search walks first and the set rules see: the merged index when merge.on
    GIndex orig;                      // merge.on: the original index — a search whose best score two finished extensions share is run again on it
    GMerge merge;                     //   (which of them finishes first depends on the steps' granularity; everything else about a search does not)
    const GProb* probs; uint32_t n;
    const uint32_t* order;            // the order the threads take the problems in: sorted by the node of the first seed, so that the reads of a
                                      // wavefront (and of the wavefronts around it) walk the same few records and bases of the index
    const char* reads;                // masked: ACGT or X
    const vgk_seed* seeds;
    int32_t match, mismatch, bonus;
    uint32_t  flat_min_idle;          // flat form: lanes without a search that make a wavefront run the branch that begins one
    GExt*     winners;                // flat form: room for one per seed of the batch (a read's winners start at its seed_off)
    GScratch* scratch;                // one per resident thread
    GCold*    cold;                   // likewise
    vgk_gapless_result* results;      // per problem (ext_begin indexes `ext`)
    vgk_extension* ext; uint32_t* nodes; uint32_t* mism;
    uint8_t* retry;                   // [n]: 1 = the flat search handed the read to the slab kernel (kept apart from results[].status, which that kernel
                                      // rewrites while the rules kernel runs beside it)
    unsigned long long* counters;     // [0] extensions, [1] nodes, [2] mismatches handed out, [3] reads the fast kernel passed on to the slab kernel, [4] reads handed to lanes (flat form), [5] searches run again on the original index (merged runs)
    unsigned long long caps[3];
};

VGK_HD unsigned long long g_bump(unsigned long long* counter, unsigned long long n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(counter, n);
#else
    const unsigned long long at = *counter; *counter += n; return at;
#endif
}

struct GCtx { const GaplessParams* P; const char* seq; uint32_t L; };

// seed `idx` of the batch as the search sees it.  status: VGK_OK, G_BADNODE (node outside the index) — with merged runs the node and the
// diagonal are those on the merged node, `seed_end` the offset in that node where the ORIGINAL seed node ends (the initial match with any
// number of mismatches covers the seed node only, :213-237) and `orig_len` that node's length (the offset check is against it)
struct GSeedIn { int32_t node; int64_t diff; uint32_t seed_begin, seed_end, orig_len; };
// (MG: the kernel is built with the merged-run code at all — a build without it keeps the registers of the plain search)
template <bool MG> VGK_HD bool g_seed_in(const GaplessParams& P, uint32_t idx, GSeedIn& out, bool merged = true) {
    const vgk_seed sd = P.seeds[idx];
    if (!(MG && P.merge.on) || !merged) {
        const GIndex& h = (MG && P.merge.on) ? P.orig : P.index;
        if (sd.node >= h.n_oriented) return false;
        out.node = (int32_t)sd.node; out.diff = sd.diff; out.seed_begin = 0; out.orig_len = g_len(h, (int32_t)sd.node); out.seed_end = out.orig_len;
        return true;
    }
    if (sd.node >= P.merge.n_orig_oriented) return false;
    const uint64_t mp = P.merge.seed_map[sd.node];
    const uint32_t v = sd.node >> 1, olen = P.merge.ocol[v + 1] - P.merge.ocol[v], off = (uint32_t)(mp >> 32);
    out.node = (int32_t)(uint32_t)mp; out.diff = (int64_t)sd.diff - (int64_t)off; out.seed_begin = off; out.seed_end = off + olen; out.orig_len = olen;
    return true;
}

// Eight bases per compare, like the reference's memcpy'd uint64 words (:219-224).  Both buffers carry 8 bytes of padding at
// either end, so a word that straddles the end of the data stays inside the allocation.
VGK_HD uint64_t g_load8(const char* p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
// A word that differs is not walked byte by byte: the bytes that differ are marked with one bit each, counted, and taken in one
// step when the mismatch budget covers them — in a wavefront some lane has a mismatch in nearly every word, and a byte loop that one
// lane needs is paid by all 64.  Only the word in which the budget runs out is searched for the stopping position.
VGK_HD uint64_t g_nzbytes(uint64_t x) { const uint64_t L = 0x7f7f7f7f7f7f7f7full; return (((x & L) + L) | x) & ~L; }      // bit 7 of every nonzero byte
#if defined(__HIP_DEVICE_COMPILE__)
VGK_HD uint32_t g_pop64(uint64_t x) { return (uint32_t)__popcll(x); }
VGK_HD uint32_t g_ctz64(uint64_t x) { return (uint32_t)__ffsll((unsigned long long)x) - 1u; }
VGK_HD uint32_t g_clz64(uint64_t x) { return (uint32_t)__clzll((long long)x); }
#else
VGK_HD uint32_t g_pop64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }
VGK_HD uint32_t g_ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }
VGK_HD uint32_t g_clz64(uint64_t x) { return (uint32_t)__builtin_clzll(x); }
#endif
// forward: compare a[0..left) with b[0..left); stops BEFORE the mismatch that would reach `limit`; returns the bases consumed
VGK_HD uint32_t g_match_fwd(const char* a, const char* b, uint32_t left, uint32_t& internal, uint32_t limit) {
    uint32_t n = 0;
    while (n < left) {
        const uint64_t wa = g_load8(a + n), wb = g_load8(b + n);
        const uint32_t len = left - n < 8 ? left - n : 8;
        uint64_t m = g_nzbytes(wa ^ wb);
        if (len < 8) m &= (1ull << (8 * len)) - 1ull;                       // the first `len` bytes are the low ones
        if (m) {
            const uint32_t cnt = g_pop64(m), allowed = limit > internal + 1 ? limit - 1 - internal : 0u;
            if (cnt > allowed) {                                            // the budget runs out inside this word: at mismatch number allowed + 1
                for (uint32_t k = 0; k < allowed; ++k) m &= m - 1;
                internal += allowed;
                return n + (g_ctz64(m) >> 3);
            }
            internal += cnt;
        }
        n += len;
    }
    return n;
}
// backward: compare a[-1], a[-2], ... with b[-1], b[-2], ... for up to `left` bases
VGK_HD uint32_t g_match_bwd(const char* a, const char* b, uint32_t left, uint32_t& internal, uint32_t limit) {
    uint32_t n = 0;
    while (n < left) {
        const uint64_t wa = g_load8(a - n - 8), wb = g_load8(b - n - 8);
        const uint32_t len = left - n < 8 ? left - n : 8;
        uint64_t m = g_nzbytes(wa ^ wb);
        if (len < 8) m &= ~0ull << (8 * (8 - len));                         // the first `len` bytes are the high ones
        if (m) {
            const uint32_t cnt = g_pop64(m), allowed = limit > internal + 1 ? limit - 1 - internal : 0u;
            if (cnt > allowed) {
                for (uint32_t k = 0; k < allowed; ++k) m &= ~(1ull << (63u - g_clz64(m)));
                internal += allowed;
                return n + (g_clz64(m) >> 3);
            }
            internal += cnt;
        }
        n += len;
    }
    return n;
}

// either way through one body: backward, the words are fetched from below the pointers and byte-swapped, so that the first base
// compared is the low byte as it is forward
VGK_HD uint64_t g_bswap64(uint64_t x) { return __builtin_bswap64(x); }
VGK_HD uint32_t g_match_dir(const char* a, const char* b, uint32_t left, uint32_t& internal, uint32_t limit, bool back) {
    uint32_t n = 0;
    while (n < left) {
        const int64_t at = back ? -(int64_t)n - 8 : (int64_t)n;
        uint64_t x = g_load8(a + at) ^ g_load8(b + at);
        if (back) x = g_bswap64(x);
        const uint32_t len = left - n < 8 ? left - n : 8;
        uint64_t m = g_nzbytes(x);
        if (len < 8) m &= (1ull << (8 * len)) - 1ull;
        if (m) {
            const uint32_t cnt = g_pop64(m), allowed = limit > internal + 1 ? limit - 1 - internal : 0u;
            if (cnt > allowed) {
                for (uint32_t k = 0; k < allowed; ++k) m &= m - 1;
                internal += allowed;
                return n + (g_ctz64(m) >> 3);
            }
            internal += cnt;
        }
        n += len;
    }
    return n;
}

VGK_HD void g_set_score(const GCtx& c, GEntry& e) {                                   // (:201-209)
    e.score = (int32_t)((e.r1 - e.r0) * (uint32_t)c.P->match) - (int32_t)(e.internal * (uint32_t)(c.P->match + c.P->mismatch))
            + e.left_full * c.P->bonus + e.right_full * c.P->bonus;
}
// The queue orders (score, insertion number) — highest score first, the later insertion among equals (:567-571).  Its keys
// carry both and an index, so sifting touches only the small key array, never the entries themselves.
VGK_HD uint64_t g_key(const GEntry& e, uint32_t idx) { return ((uint64_t)((uint32_t)e.score ^ 0x80000000u) << 32) | ((uint64_t)e.number << 16) | idx; }
VGK_HD uint64_t g_key(const GLean& e, uint32_t idx) { return ((uint64_t)((uint32_t)e.score ^ 0x80000000u) << 32) | ((uint64_t)(e.pn >> 16) << 16) | idx; }
VGK_HD GLink g_link(const GEntry& e) { GLink l; l.node = e.node; l.parent = (int16_t)e.parent; l.front = e.front; l.pad = 0; return l; }

// ---- where a seed's search keeps its state ------------------------------------------------------------------------------
// Two stores behind one interface.  GStoreSlab: everything in the per-thread slab in HBM (round 1's layout; an entry that waits in
// the queue sits at its own index: 40 bytes x 160).  GStoreLds, the fast kernel's: the QUEUE — the partial extensions that wait while
// a better one is being extended, rarely more than three — in LDS, lane-interleaved dwords, an entry packed into 20 bytes (which
// needs reads and nodes of at most 255 bases and at most 254 visits per node); the path links stay in the slab, where they are
// 8 bytes written once, behind each other, and read once for the winner.  What made the slab kernel move 42 x its algorithmic
// bytes was the queue: 40-byte entries and 8-byte keys scattered over an 11 KB slab per thread, 2.9 GB over the resident threads.
// A search whose queue outgrows the LDS slots makes the read G_RETRY: the slab kernel runs it again.
// Index field of a queue key: entry number in the low byte, slot in the byte above.
constexpr int32_t  G_RETRY = 1;                  // vgk_gapless_result::status of a read the fast kernel hands to the slab kernel
#ifndef VGK_GAPLESS_FAST_QUEUE
#define VGK_GAPLESS_FAST_QUEUE 5
#endif
constexpr uint32_t G_FAST_QUEUE = VGK_GAPLESS_FAST_QUEUE;       // at most 32
constexpr uint32_t G_FAST_DW = 2 * G_FAST_QUEUE + 5 * G_FAST_QUEUE;      // dwords of LDS per thread

struct GStoreSlab {
    GScratch& s;

    static constexpr uint32_t ENTRIES = (uint32_t)G_POOL;
    static constexpr int32_t  FULL = VGK_ETOOBIG;
    VGK_HD void begin_seed() {}
    VGK_HD uint64_t heap_get(uint32_t i) const { return s.heap[i]; }
    VGK_HD void heap_set(uint32_t i, uint64_t k) { s.heap[i] = k; }
    VGK_HD bool slot_take(uint32_t entry, uint32_t& idx) { idx = entry; return true; }
    VGK_HD void slot_free(uint32_t) {}
    VGK_HD uint32_t entry_of(uint32_t idx) const { return idx; }
    VGK_HD void pool_set(uint32_t idx, const GEntry& e) { s.pool[idx] = g_pack(e); }
    VGK_HD GEntry pool_get(const GCtx&, uint32_t idx, uint64_t) const { return g_unpack(s.pool[idx]); }
    VGK_HD void link_set(uint32_t i, const GEntry& e) { s.link[i] = g_link(e); }
    VGK_HD GLink link_get(uint32_t i) const { return s.link[i]; }
};

struct GStoreLds {
    uint32_t* base; uint32_t stride;             // dword k of this thread = base[k * stride]
    GScratch& s;                                 // the thread's slab: path links only
    uint32_t free_slots;
    static constexpr uint32_t ENTRIES = (uint32_t)G_POOL;
    static constexpr int32_t  FULL = G_RETRY;
    static constexpr uint32_t HEAP0 = 0, POOL0 = 2 * G_FAST_QUEUE;
    VGK_HD uint32_t& dw(uint32_t k) const { return base[k * stride]; }
    VGK_HD void begin_seed() { free_slots = (1u << G_FAST_QUEUE) - 1u; }
    VGK_HD uint64_t heap_get(uint32_t i) const { return (uint64_t)dw(HEAP0 + 2 * i) | ((uint64_t)dw(HEAP0 + 2 * i + 1) << 32); }
    VGK_HD void heap_set(uint32_t i, uint64_t k) { dw(HEAP0 + 2 * i) = (uint32_t)k; dw(HEAP0 + 2 * i + 1) = (uint32_t)(k >> 32); }
    VGK_HD bool slot_take(uint32_t entry, uint32_t& idx) {
        if (!free_slots) return false;
        uint32_t slot = 0; while (!((free_slots >> slot) & 1u)) ++slot;
        free_slots &= ~(1u << slot); idx = (slot << 8) | entry; return true;
    }
    VGK_HD void slot_free(uint32_t idx) { free_slots |= 1u << (idx >> 8); }
    VGK_HD uint32_t entry_of(uint32_t idx) const { return idx & 0xffu; }
    // 20 bytes: offset r0 r1 internal | old flags score16 | fn | bn | flo fhi blo bhi (an empty range is [1, 0])
    VGK_HD void pool_set(uint32_t idx, const GEntry& e) {
        const uint32_t k = POOL0 + 5 * (idx >> 8);
        const uint32_t flags = (uint32_t)(e.front | (e.left_full << 1) | (e.right_full << 2) | (e.left_max << 3) | (e.right_max << 4));
        const bool fe = e.state.flo > e.state.fhi, be = e.state.blo > e.state.bhi;
        dw(k) = e.offset | (e.r0 << 8) | (e.r1 << 16) | (e.internal << 24);
        dw(k + 1) = e.old | (flags << 8) | ((uint32_t)(uint16_t)(int16_t)e.score << 16);
        dw(k + 2) = (uint32_t)e.state.fn; dw(k + 3) = (uint32_t)e.state.bn;
        dw(k + 4) = (fe ? 1u : (uint32_t)e.state.flo) | ((fe ? 0u : (uint32_t)e.state.fhi) << 8) | ((be ? 1u : (uint32_t)e.state.blo) << 16) | ((be ? 0u : (uint32_t)e.state.bhi) << 24);
    }
    VGK_HD GEntry pool_get(const GCtx&, uint32_t idx, uint64_t key) const {
        const uint32_t k = POOL0 + 5 * (idx >> 8);
        const uint32_t a = dw(k), b = dw(k + 1), r = dw(k + 4);
        GEntry e;
        e.parent = -1; e.node = -1; e.number = (uint32_t)(key >> 16) & 0xffffu;
        e.offset = a & 0xffu; e.r0 = (a >> 8) & 0xffu; e.r1 = (a >> 16) & 0xffu; e.internal = a >> 24;
        e.old = b & 0xffu; const uint32_t fl = (b >> 8) & 0xffu; e.score = (int32_t)(int16_t)(uint16_t)(b >> 16);
        e.front = fl & 1; e.left_full = (fl >> 1) & 1; e.right_full = (fl >> 2) & 1; e.left_max = (fl >> 3) & 1; e.right_max = (fl >> 4) & 1;
        e.pad[0] = e.pad[1] = e.pad[2] = 0; e.frec = e.brec = G_NO_REC;
        e.state.fn = (int32_t)dw(k + 2); e.state.bn = (int32_t)dw(k + 3);
        const uint32_t flo = r & 0xffu, fhi = (r >> 8) & 0xffu, blo = (r >> 16) & 0xffu, bhi = r >> 24;
        const bool fe = flo > fhi, be = blo > bhi;
        e.state.flo = fe ? 0 : (int32_t)flo; e.state.fhi = fe ? -1 : (int32_t)fhi; e.state.blo = be ? 0 : (int32_t)blo; e.state.bhi = be ? -1 : (int32_t)bhi;
        return e;
    }
    VGK_HD void link_set(uint32_t i, const GEntry& e) { s.link[i] = g_link(e); }      // (streaming stores/loads for the links, to spare the caches: measured, no fewer bytes fetched and 6 % slower)
    VGK_HD GLink link_get(uint32_t i) const { return s.link[i]; }
};

template <class ST> VGK_HD bool g_heap_push(ST& S, uint32_t& hn, const GEntry& e, uint32_t entry) {
    uint32_t idx;
    if (!S.slot_take(entry, idx)) return false;
    const uint64_t key = g_key(e, idx);
    S.pool_set(idx, e);                                            // it will be popped from the store
    uint32_t i = hn++;
    while (i) { const uint32_t p = (i - 1) / 2; const uint64_t pk = S.heap_get(p); if (pk >= key) break; S.heap_set(i, pk); i = p; }
    S.heap_set(i, key);
    return true;
}
// a new entry: the better of it and the held-back candidate stays in registers, the other one goes to the queue
template <class ST> VGK_HD bool g_offer(ST& S, uint32_t& hn, bool& have_cand, GLean& cand, uint32_t& cand_idx, const GEntry& e, uint32_t idx) {
    if (!have_cand) { cand = g_lean(e); cand_idx = idx; have_cand = true; return true; }
    if (g_key(e, idx) > g_key(cand, cand_idx)) { const bool ok = g_heap_push(S, hn, g_fat(cand), cand_idx); cand = g_lean(e); cand_idx = idx; return ok; }
    return g_heap_push(S, hn, e, idx);
}
template <class ST> VGK_HD uint64_t g_heap_pop(ST& S, uint32_t& hn) {
    const uint64_t top = S.heap_get(0);
    const uint64_t last = S.heap_get(--hn);
    uint32_t i = 0;
    for (;;) {
        const uint32_t l = 2 * i + 1, r = l + 1;
        if (l >= hn) break;
        const uint64_t kl = S.heap_get(l), kr = r < hn ? S.heap_get(r) : 0;
        const uint32_t m = (r < hn && kr > kl) ? r : l; const uint64_t km = m == r ? kr : kl;
        if (km <= last) break;
        S.heap_set(i, km); i = m;
    }
    if (hn) S.heap_set(i, last);
    return top;
}
// path of an entry, front to back; returns its length or -1 when it does not fit
template <class ST> VGK_HD int g_path(const ST& S, int32_t idx, int32_t* out) {
    int32_t fwd[G_PATH]; int nf = 0, nb = 0;
    for (int32_t i = idx; i >= 0;) {
        const GLink e = S.link_get((uint32_t)i);
        i = e.parent;
        if (e.node < 0) continue;
        if (e.front) { if (nb >= G_PATH) return -1; out[nb++] = e.node; }      // the latest front node is the first of the path
        else { if (nf >= G_PATH) return -1; fwd[nf++] = e.node; }              // collected back to front
    }
    if (nb + nf > G_PATH) return -1;
    for (int k = 0; k < nf; ++k) out[nb + k] = fwd[nf - 1 - k];
    return nb + nf;
}
VGK_HD bool gx_full(const GExt& e) { return e.left_full && e.right_full; }
VGK_HD bool gx_contains(const GIndex& h, const GExt& e, int32_t node, int64_t diff) {             // (:17-41)
    uint32_t read_offset = e.r0, node_offset = e.offset;
    for (uint32_t i = 0; i < e.path_len; ++i) {
        const uint32_t a = g_len(h, e.path[i]) - node_offset, b = e.r1 - read_offset; const uint32_t len = a < b ? a : b;
        if (e.path[i] == node && (int64_t)read_offset - (int64_t)node_offset == diff) return true;
        read_offset += len; node_offset = 0;
    }
    return false;
}
// The same test against the diagonals (read offset - node offset) of the path's nodes, computed once when an extension becomes the
// best: gx_contains reads two dependent words per path node to learn its length, for every seed it is asked about.
VGK_HD void gx_diagonals(const GIndex& h, const GExt& e, int32_t* diag) {
    uint32_t read_offset = e.r0, node_offset = e.offset;
    for (uint32_t i = 0; i < e.path_len; ++i) {
        const uint32_t a = g_len(h, e.path[i]) - node_offset, b = e.r1 - read_offset; const uint32_t len = a < b ? a : b;
        diag[i] = (int32_t)read_offset - (int32_t)node_offset;
        read_offset += len; node_offset = 0;
    }
}
VGK_HD bool gx_contains_diag(const GExt& e, const int32_t* diag, int32_t node, int64_t diff) {
    bool hit = false;
    for (uint32_t i = 0; i < e.path_len; ++i) hit |= e.path[i] == node && (int64_t)diag[i] == diff;
    return hit;
}
VGK_HD uint32_t gx_overlap(const GIndex& h, const GExt& x, const GExt& y) {                        // (:69-103)
    uint32_t result = 0, xp = x.r0, yp = y.r0, xi = 0, yi = 0, xo = x.offset, yo = y.offset;
    while (xp < x.r1 && yp < y.r1) {
        if (xp == yp && x.path[xi] == y.path[yi] && xo == yo) {
            uint32_t len = g_len(h, x.path[xi]) - xo;
            if (x.r1 - xp < len) len = x.r1 - xp;
            if (y.r1 - yp < len) len = y.r1 - yp;
            result += len; xp += len; yp += len; ++xi; ++yi; xo = yo = 0;
        } else if (xp <= yp) { xp += g_len(h, x.path[xi]) - xo; ++xi; xo = 0; }
        else { yp += g_len(h, y.path[yi]) - yo; ++yi; yo = 0; }
    }
    return result;
}
VGK_HD bool gs_eq(const GState& a, const GState& b) { return a.fn == b.fn && a.flo == b.flo && a.fhi == b.fhi && a.bn == b.bn && a.blo == b.blo && a.bhi == b.bhi; }
VGK_HD bool gx_eq(const GExt& a, const GExt& b) { return a.r0 == b.r0 && a.r1 == b.r1 && gs_eq(a.state, b.state) && a.offset == b.offset; }
VGK_HD bool gx_dup_less(const GExt& a, const GExt& b) {                                            // (:333-349)
    if (a.r0 != b.r0) return a.r0 < b.r0;
    if (a.r1 != b.r1) return a.r1 < b.r1;
    if (a.state.bn != b.state.bn) return a.state.bn < b.state.bn;
    if (a.state.fn != b.state.fn) return a.state.fn < b.state.fn;
    if (a.state.blo != b.state.blo) return a.state.blo < b.state.blo;
    if (a.state.bhi != b.state.bhi) return a.state.bhi < b.state.bhi;
    if (a.state.flo != b.state.flo) return a.state.flo < b.state.flo;
    if (a.state.fhi != b.state.fhi) return a.state.fhi < b.state.fhi;
    return a.offset < b.offset;
}
VGK_HD bool gx_full_less(const GExt& a, const GExt& b) {                                           // (:302-307)
    if (gx_full(a) && gx_full(b)) return a.internal < b.internal;
    return gx_full(a) && !gx_full(b);
}
// stable insertion sort of an index permutation (the sets are small; moving 500-byte records would not pay)
template <class RESV, class Less> VGK_HD void gx_sort(const RESV& v, uint8_t* order, uint32_t n, Less less) {
    for (uint32_t i = 1; i < n; ++i) { const uint8_t x = order[i]; uint32_t j = i; while (j && less(v[x], v[order[j - 1]])) { order[j] = order[j - 1]; --j; } order[j] = x; }
}
template <class RESV> VGK_HD uint32_t gx_remove_duplicates(const RESV& v, uint8_t* order, uint32_t n) {                  // (:332-365)
    gx_sort(v, order, n, [](const GExt& a, const GExt& b) { return gx_dup_less(a, b); });
    uint32_t tail = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const GExt& e = v[order[i]];
        if (e.r1 == e.r0) continue;
        if (tail == 0 || !gx_eq(e, v[order[tail - 1]])) order[tail++] = order[i];
    }
    return tail;
}
// the mismatch positions of an extension into mm[] (at most G_MISM); sets e.n_mism
VGK_HD void gx_find_mismatches(const GCtx& c, GExt& e, uint32_t* mm, bool& overflow) {             // (:368-387)
    e.n_mism = 0;
    if (!e.internal) return;
    const GIndex& h = c.P->index;
    uint32_t node_offset = e.offset, read_offset = e.r0;
    for (uint32_t i = 0; i < e.path_len && read_offset < e.r1; ++i) {
        uint32_t tl; const char* t = h.seq + g_seq_of(h, (uint32_t)(e.path[i]), tl);
        uint32_t left = tl - node_offset < e.r1 - read_offset ? tl - node_offset : e.r1 - read_offset;
        while (left) {                                                       // eight bases per compare; the positions come out of the mask of differing bytes
            const uint32_t len = left < 8 ? left : 8;
            uint64_t m = g_nzbytes(g_load8(t + node_offset) ^ g_load8(c.seq + read_offset));
            if (len < 8) m &= (1ull << (8 * len)) - 1ull;
            while (m) {
                if (e.n_mism >= G_MISM) { overflow = true; return; }
                mm[e.n_mism++] = read_offset + (g_ctz64(m) >> 3);
                m &= m - 1;
            }
            node_offset += len; read_offset += len; left -= len;
        }
        node_offset = 0;
    }
}
VGK_HD bool gx_trim(const GCtx& c, GExt& e, uint32_t* mm) {                                         // (:421-529)
    if (!e.n_mism) return false;
    const GIndex& h = c.P->index; const int32_t match = c.P->match, mismatch = c.P->mismatch, bonus = c.P->bonus;
    uint32_t mi = 0, c0 = e.r0, c1 = mm[0];
    int32_t cur = (int32_t)(c1 - c0) * match + (e.left_full ? bonus : 0);
    uint32_t b0 = c0, b1 = c1; int32_t best = cur;
    while (mi < e.n_mism) {
        if (cur >= mismatch) { ++c1; cur -= mismatch; }
        else { c0 = c1 = mm[mi] + 1; cur = 0; }
        ++mi;
        if (mi == e.n_mism) { cur += (int32_t)(e.r1 - c1) * match; c1 = e.r1; if (e.right_full) cur += bonus; }
        else { cur += (int32_t)(mm[mi] - c1) * match; c1 = mm[mi]; }
        if (cur > best || (cur > 0 && cur == best && c1 - c0 > b1 - b0)) { b0 = c0; b1 = c1; best = cur; }
    }
    if (b0 == e.r0 && b1 == e.r1) return false;
    if (b1 == b0) { e.path_len = 0; e.r0 = b0; e.r1 = b1; e.n_mism = 0; e.score = 0; e.left_full = e.right_full = 0; return true; }
    if (b0 > e.r0) e.left_full = 0;
    if (b1 < e.r1) e.right_full = 0;
    uint32_t node_offset = e.offset, read_offset = e.r0;
    e.r0 = b0; e.r1 = b1; e.score = best;
    uint32_t head = 0;
    while (head < e.path_len) {
        const uint32_t nl = g_len(h, e.path[head]);
        read_offset += nl - node_offset; node_offset = 0;
        if (read_offset > e.r0) { e.offset = nl - (read_offset - e.r0); break; }
        ++head;
    }
    uint32_t tail = head + 1;
    while (read_offset < e.r1) { read_offset += g_len(h, e.path[tail]); ++tail; }
    if (head > 0 || tail < e.path_len) {
        for (uint32_t k = 0; k < tail - head; ++k) e.path[k] = e.path[head + k];
        e.path_len = tail - head;
        GState s = gs_find(h, e.path[0]);                                                          // bd_find
        for (uint32_t k = 1; k < e.path_len; ++k) s = gs_extend(h, s, e.path[k]);
        e.state = s;
    }
    uint32_t mh = 0; while (mh < e.n_mism && mm[mh] < e.r0) ++mh;
    uint32_t mt = mh; while (mt < e.n_mism && mm[mt] < e.r1) ++mt;
    for (uint32_t k = 0; k < mt - mh; ++k) mm[k] = mm[mh + k];
    e.n_mism = mt - mh;
    return true;
}

// ---- one seed's search, cut into begin / step / end so that a kernel can run it as ONE flat loop -----------------------------------
// GaplessExtender::extend is "for every seed: a best-first search; then rules over the set of winners".  A thread that runs the
// nested loops waits, at the end of every search, for the slowest lane of its wavefront; a thread that runs `begin` when it has no
// search, else one `step`, in a single loop, never waits: lanes of a wavefront work on different seeds (of different reads) but on
// the same instruction — the loop body is the expansion of one partial extension.  A seed's search does not depend on the other
// seeds of its cluster; the one rule that couples them ("skip a seed that the best exact full-length extension so far already
// contains", :545-549) only decides whether a winner is USED, and is applied afterwards, in seed order, by gapless_finish_read.
// Section timing for kernel work (-DVGAMD_GAPLESS_PROF): wave cycles per section of the flat loop, summed into counters[8 + section]
#if defined(VGAMD_GAPLESS_PROF) && defined(__HIPCC__)
struct GProf { unsigned long long t0, acc[12]; __device__ void start() { t0 = __builtin_readcyclecounter(); for (int i = 0; i < 12; ++i) acc[i] = 0; } __device__ void tick(int s) { const unsigned long long t = __builtin_readcyclecounter(); acc[s] += t - t0; t0 = t; } };
#define G_TICK(prof, s) do { if (prof) (prof)->tick(s); } while (0)
#else
struct GProf {};
#define G_TICK(prof, s) do { } while (0)
#endif
struct GSearch {
    GProf* prof;
    uint32_t np, hn, number; bool have_cand; GLean cand; uint32_t cand_idx; int32_t best; GBest best_e;
    uint32_t L, max_mm;
    bool merged, tie;                // the search runs on the merged index | the best score so far was reached by two finished extensions
};
constexpr int32_t G_REDO = 4;        // g_search_end: a merged search whose best score is a tie — the caller begins the same seed again with merged = false
constexpr int32_t G_BADNODE = 2, G_BADOFF = 3;          // winner statuses (beside VGK_OK, VGK_ETOOBIG, G_RETRY): a seed node out of range (checked
                                                        // before the skip rule), a seed offset out of range (checked after it)
// winner record of a seed = GExt with pad[0] = 1 when there is an extension, pad[1] = status
template <bool MG, class ST>
VGK_HD int g_search_begin(const GaplessParams& P, const GCtx& c, const GProb& pb, uint32_t si, ST& Q, GSearch& s, bool merged = true) {
    s.merged = MG && merged && P.merge.on; s.tie = false;
    const GIndex& h = (MG && P.merge.on && !s.merged) ? P.orig : P.index;
    GSeedIn sin;
    if (!g_seed_in<MG>(P, pb.seed_off + si, sin, s.merged)) return G_BADNODE;
    const uint32_t L = pb.read_len;
    // (the seed's own offsets are those on the ORIGINAL node: read_offset - node_offset = the caller's diff; the merged node adds seed_begin)
    const int64_t diff0 = sin.diff + (int64_t)sin.seed_begin;
    const uint32_t read_offset = diff0 < 0 ? 0u : (uint32_t)diff0, node_offset0 = diff0 < 0 ? (uint32_t)(-diff0) : 0u;
    if (read_offset > L || node_offset0 > sin.orig_len) return G_BADOFF;
    const int32_t snode = sin.node;
    const uint32_t node_offset = node_offset0 + sin.seed_begin;
    const uint32_t ro_f = h.rec_off[(uint32_t)snode], ro_b = h.rec_off[(uint32_t)snode ^ 1u];
    const GQuad hf = g_quad(h.rec + ro_f), hb = g_quad(h.rec + ro_b);           // {visits, edges, length, bases} of the seed node on either strand
    const uint32_t slen = hf.z;
    s.np = 0; s.hn = 0; s.number = 0; s.have_cand = false; s.cand_idx = 0; s.best = -1; s.L = L; s.max_mm = pb.max_mm;
    s.best_e.score = 0; s.best_e.rr = 0;
    Q.begin_seed();
    // the seed node itself: any number of mismatches (:213-237)
    GEntry m;
    m.parent = -1; m.node = snode; m.front = 0; m.offset = node_offset; m.r0 = m.r1 = read_offset; m.internal = 0;
    m.left_full = m.right_full = m.left_max = m.right_max = 0; m.pad[0] = m.pad[1] = m.pad[2] = 0; m.frec = ro_f; m.brec = ro_b;
    m.state.fn = snode; m.state.flo = 0; m.state.fhi = (int32_t)hf.x - 1; m.state.bn = snode ^ 1; m.state.blo = 0; m.state.bhi = (int32_t)hb.x - 1;      // gs_find
    const char* t = h.seq + hf.w;
    const uint32_t left = L - m.r1 < sin.seed_end - node_offset ? L - m.r1 : sin.seed_end - node_offset;
    m.r1 += g_match_fwd(c.seq + m.r1, t + node_offset, left, m.internal, 0xffffffffu);
    m.old = m.internal;
    if (m.r0 == 0) m.left_full = m.left_max = 1;
    if (m.r1 >= L) m.right_full = m.right_max = 1;
    else if (MG && sin.seed_end < slen) {
        // merged runs: the seed node ends inside the merged node — what follows in it are the run's next nodes, which match_forward (:239-266) would
        // take one by one under the mismatch limit: here in one piece, and where it stops the entry is right-maximal exactly as the node-by-node
        // form leaves it (a next node that matches nothing is dropped and the state kept as right-maximal, :617-619, :633-637)
        const uint32_t lim_a = pb.max_mm + 1, lim_b = pb.max_mm / 2 + m.old + 1, limit = lim_a > lim_b ? lim_a : lim_b;
        const uint32_t rest = slen - sin.seed_end, room = L - m.r1 < rest ? L - m.r1 : rest;
        const uint32_t no = g_match_fwd(c.seq + m.r1, t + sin.seed_end, room, m.internal, limit);
        m.r1 += no;
        if (m.r1 >= L) { m.right_full = m.right_max = 1; m.old = m.internal; }
        else if (no < rest) { m.right_max = 1; m.old = m.internal; }
    }
    g_set_score(c, m); m.number = s.number++;
    Q.link_set(s.np, m);
    s.cand = g_lean(m); s.cand_idx = s.np; s.have_cand = true; ++s.np;
    return VGK_OK;
}
// one pop + expansion; VGK_OK = go on (or the queue ran dry: check g_search_live), else the status that ends the search
// The queue pops (score, insertion number) maxima.  The best entry created by an expansion is held back in registers (`cand`): when
// it beats the queue's top — always, on a non-branching stretch — it is the next one popped and the round trip through the store
// is skipped; otherwise it joins the queue first.
VGK_HD bool g_search_live(const GSearch& s) { return s.hn || s.have_cand; }
template <bool MG, class ST>
VGK_HD int g_search_step(const GaplessParams& P, const GCtx& c, ST& Q, GSearch& s) {
    const GIndex& h = (MG && P.merge.on && !s.merged) ? P.orig : P.index;
    const uint32_t L = s.L, max_mm = s.max_mm;
    uint32_t ci; GEntry cur;
    if (s.have_cand && (s.hn == 0 || g_key(s.cand, s.cand_idx) > Q.heap_get(0))) { ci = s.cand_idx; cur = g_fat(s.cand); s.have_cand = false; }
    else {
        if (s.have_cand) { if (!g_heap_push(Q, s.hn, g_fat(s.cand), s.cand_idx)) return ST::FULL; s.have_cand = false; }
        const uint64_t top = g_heap_pop(Q, s.hn);
        const uint32_t idx = (uint32_t)(top & 0xffffu);
        cur = Q.pool_get(c, idx, top); Q.slot_free(idx);
        ci = Q.entry_of(idx);
    }
    G_TICK(s.prof, 2);
    // One expansion, to the right while the entry can grow there, else to the left (:590-700).  The two directions run through the
    // same instructions — which record, which state, which way the bases are compared are data — because in a wavefront there are
    // always lanes going either way, and two code paths would each be paid by all of them.
    const bool right = !cur.right_max;
    if (MG && s.merged && !right && !cur.left_max && cur.offset > 0) {
        // merged runs: the path's first node has bases before the alignment (the seed node lies inside a run): the run's earlier nodes, which
        // match_backward (:268-296) would take one by one — in one piece, the same limit, the same flags where it stops
        const uint32_t lim_a = max_mm + 1, lim_b = max_mm / 2 + cur.old + 1, limit = lim_a > lim_b ? lim_a : lim_b;
        const uint32_t first = (uint32_t)cur.state.bn ^ 1u;
        const char* t = h.seq + g_seq_off(h, first) + cur.offset;                        // one past the last base before the alignment
        const uint32_t room = cur.r0 < cur.offset ? cur.r0 : cur.offset;
        GEntry nx = cur; nx.parent = (int32_t)ci; nx.node = -1;
        const uint32_t no = g_match_dir(c.seq + nx.r0, t, room, nx.internal, limit, true);
        if (no == 0) cur.left_max = 1;                                                   // (a node that matches nothing is dropped: the entry is left-maximal as it stands)
        else {
            if (s.np >= ST::ENTRIES) return ST::FULL;
            nx.r0 -= no; nx.offset -= no;
            if (nx.r0 == 0) nx.left_full = nx.left_max = 1;
            else if (nx.offset > 0) nx.left_max = 1;
            g_set_score(c, nx); nx.number = s.number++;
            Q.link_set(s.np, nx);
            if (!g_offer(Q, s.hn, s.have_cand, s.cand, s.cand_idx, nx, s.np)) return ST::FULL;
            ++s.np;
            return VGK_OK;
        }
    }
    if (right || !cur.left_max) {
        uint32_t num_ext = 0; bool found = false;
        const uint32_t lim_a = max_mm + 1, lim_b = max_mm / 2 + cur.old + 1, limit = lim_a > lim_b ? lim_a : lim_b;
        uint32_t ri = right ? cur.frec : cur.brec;
        if (ri == G_NO_REC) { ri = h.rec_off[(uint32_t)(right ? cur.state.fn : cur.state.bn)]; if (right) cur.frec = ri; else cur.brec = ri; }
        const GRecMem orec{h.rec + ri};
        const uint32_t ne = orec.ne();
        const GState from = right ? cur.state : gs_flip(cur.state);
        const bool few = ne <= 4 && !gs_empty(from);
        const GCounts cn = few ? g_counts(orec, from.flo, from.fhi) : GCounts{0, 0};
        for (uint32_t e = 0; e < ne; ++e) {
            const GQuad ed = orec.edge(e);
            const int32_t x = (int32_t)ed.x; if (x < 0) continue;
            GState ns = few ? gs_extend_counted(orec, from, e, ed, cn) : gs_extend(h, from, x);
            if (!right) ns = gs_flip(ns);                                                        // bdExtendBackward
            G_TICK(s.prof, 7);
            if (gs_empty(ns)) continue;
            if (s.np >= ST::ENTRIES) return ST::FULL;
            const uint32_t wl = ed.y >> 16;
            // match_forward (:239-266) over the successor's bases from their start / match_backward (:268-296) over the bases of the other
            // strand of x (same length, one strand_shift away) from their end
            const char* t = right ? h.seq + ed.z : h.seq + ((x & 1) ? ed.z - h.strand_shift : ed.z + h.strand_shift) + wl;
            GEntry nx = cur; nx.parent = (int32_t)ci; nx.node = right ? x : (ns.bn ^ 1); nx.front = right ? 0 : 1; nx.state = ns;
            if (right) nx.frec = ed.w; else { nx.brec = ed.w; nx.offset = wl; }
            const uint32_t room = right ? (L - nx.r1 < wl ? L - nx.r1 : wl) : (nx.r0 < wl ? nx.r0 : wl);
            const uint32_t no = g_match_dir(c.seq + (right ? nx.r1 : nx.r0), t, room, nx.internal, limit, !right);
            G_TICK(s.prof, 8);
            if (right) {
                nx.r1 += no;
                if (no == 0) continue;
                if (nx.r1 >= L) { nx.right_full = nx.right_max = 1; nx.old = nx.internal; }
                else if (no < wl) { nx.right_max = 1; nx.old = nx.internal; }
            } else {
                nx.r0 -= no; nx.offset -= no;
                if (nx.offset >= wl) continue;
                if (nx.r0 == 0) nx.left_full = nx.left_max = 1;
                else if (nx.offset > 0) nx.left_max = 1;
            }
            g_set_score(c, nx); nx.number = s.number++;
            num_ext += gs_size(ns);
            Q.link_set(s.np, nx);
            if (!g_offer(Q, s.hn, s.have_cand, s.cand, s.cand_idx, nx, s.np)) return ST::FULL;
            ++s.np;
            found = true;
            G_TICK(s.prof, 9);
        }
        if (right) {
            if (num_ext < gs_size(cur.state)) {                                           // some haplotype ends here: keep it (:633-637)
                if (s.np >= ST::ENTRIES) return ST::FULL;
                GEntry nx = cur; nx.parent = (int32_t)ci; nx.node = -1; nx.right_max = 1; nx.old = nx.internal; nx.number = s.number++;
                Q.link_set(s.np, nx);
                    if (!g_offer(Q, s.hn, s.have_cand, s.cand, s.cand_idx, nx, s.np)) return ST::FULL;
                ++s.np;
            }
            G_TICK(s.prof, 3);
            return VGK_OK;
        }
        G_TICK(s.prof, 4);
        if (found) return VGK_OK;
        cur.left_max = 1;
    }
    // (every partial extension is extended until nothing is left: the winner is the best finished one, the first among equals — the one thing
    // that depends on the ORDER of the steps, and so on their granularity: a tie at the top sends a merged search back to the original index)
    if (s.best < 0 || s.best_e.score < cur.score) { s.best = (int32_t)ci; s.best_e = g_best(cur); s.tie = false; }
    else if (s.best_e.score == cur.score) s.tie = true;
    G_TICK(s.prof, 5);
    return VGK_OK;
}
// the winner of a finished search into `r` (pad[0] = 1 when there is one); VGK_ETOOBIG when its path does not fit
template <bool MG, class ST>
VGK_HD int g_search_end(const GaplessParams& P, const ST& Q, const GSearch& s, GExt& r) {
    r.pad[0] = 0; r.pad[1] = 0; r.path_len = 0; r.n_mism = 0;
    if (MG && s.merged && s.tie) return G_REDO;
    const GBest& b = s.best_e;
    if (!(s.best >= 0 && (b.rr >> 16) > (b.rr & 0xffffu))) return VGK_OK;
    const int plen = g_path(Q, s.best, r.path);
    if (plen < 0) return VGK_ETOOBIG;
    r.path_len = (uint32_t)plen; r.offset = b.oi & 0xffffu; r.r0 = b.rr & 0xffffu; r.r1 = b.rr >> 16; r.internal = b.oi >> 16; r.score = b.score;
    r.state.fn = b.fn; r.state.bn = b.bn; g_unrange(b.fr, r.state.flo, r.state.fhi); g_unrange(b.br, r.state.blo, r.state.bhi);
    r.left_full = (uint8_t)(b.full & 1u); r.right_full = (uint8_t)((b.full >> 1) & 1u); r.n_mism = 0; r.pad[0] = 1;
    if (MG && P.merge.on && !s.merged) {
        // a winner found on the original index, into the merged index's terms (the rules see one kind of path): consecutive nodes of a run
        // collapse into their merged node, the offset counts from that node's start, the states' nodes are the merged ones (the ranges are the same)
        const uint64_t first = P.merge.seed_map[(uint32_t)r.path[0]];
        r.offset += (uint32_t)(first >> 32);
        uint32_t n = 0;
        for (uint32_t k = 0; k < r.path_len; ++k) {
            const int32_t mo = (int32_t)(uint32_t)P.merge.seed_map[(uint32_t)r.path[k]];
            if (n == 0 || r.path[n - 1] != mo || (uint32_t)(P.merge.seed_map[(uint32_t)r.path[k]] >> 32) == 0u) r.path[n++] = mo;      // (a run entered again from its start — a cycle — is a node of its own)
        }
        r.path_len = n;
        r.state.fn = (int32_t)(uint32_t)P.merge.seed_map[(uint32_t)r.state.fn]; r.state.bn = (int32_t)(uint32_t)P.merge.seed_map[(uint32_t)r.state.bn];
    }
    return VGK_OK;
}

// merged runs: the ORIGINAL oriented nodes an extension's aligned interval touches, in path order -> their number; out (nullable): written there;
// first_offset (nullable): in: the offset in the first merged node, out: the offset in the first original node
VGK_HD uint32_t gx_expand(const GaplessParams& P, const GExt& e, uint32_t* out, uint32_t* first_offset) {
    const GMerge& M = P.merge;
    uint32_t n = 0, left = e.r1 - e.r0, a = e.offset;
    for (uint32_t i = 0; i < e.path_len && left; ++i) {
        const uint32_t mo = (uint32_t)e.path[i], m = mo >> 1, rev = mo & 1u;
        const uint32_t v0 = M.run_first[m], v1 = M.run_first[m + 1];                        // original nodes [v0, v1)
        const uint32_t mlen = M.ocol[v1] - M.ocol[v0];
        const uint32_t b = a + left < mlen ? a + left : mlen;                                // the interval [a, b) of the merged node
        // original node k of the run in path direction starts at s_k: forward v0 + k at ocol[v0 + k] - ocol[v0]; reverse v1 - 1 - k at ocol[v1] - ocol[v1 - k]
        for (uint32_t k = 0; k < v1 - v0; ++k) {
            const uint32_t v = rev ? v1 - 1u - k : v0 + k;
            const uint32_t s0 = rev ? M.ocol[v1] - M.ocol[v + 1] : M.ocol[v] - M.ocol[v0], s1 = s0 + (M.ocol[v + 1] - M.ocol[v]);
            if (s1 <= a) continue;
            if (s0 >= b) break;
            if (n == 0 && first_offset) *first_offset = a - s0;
            if (out) out[n] = 2u * v + rev;
            ++n;
        }
        left -= b - a; a = 0;
    }
    return n;
}

// The rules over a read's winners (the second half of GaplessExtender::extend): RES(i) = the i-th USED winner, n_res of them, in seed
// order; best_alignment as the seed loop left it.  `order` = n_res bytes of scratch for the permutation the rules sort.
template <bool MG, class RESV>
VGK_HD void gapless_set_rules(const GaplessParams& P, uint32_t pi, const GProb& pb, const GCtx& c, const RESV& RES, uint32_t n_res, uint32_t best_alignment, uint8_t* order) {
    const GIndex& h = P.index;
    vgk_gapless_result& out = P.results[pi];
    const uint32_t max_mm = pb.max_mm;
    for (uint32_t i = 0; i < n_res; ++i) order[i] = (uint8_t)i;
    bool overflow = false;
    uint32_t n_out = n_res;
    // the mismatch positions of the extension looked at last stay in mm[]: for the usual set of one extension the output below copies
    // them instead of walking the path against the read a second time (that walk was 40 % of what the rules kernel fetched)
    uint32_t mm[G_MISM]; uint32_t mm_owner = 0xffffffffu;
    if (best_alignment < n_res && RES[best_alignment].internal <= max_mm) {
        // the non-overlapping full-length extensions, fewest mismatches first (:301-329)
        gx_sort(RES, order, n_res, [](const GExt& a, const GExt& b) { return gx_full_less(a, b); });
        uint32_t tail = 0;
        for (uint32_t i = 0; i < n_res; ++i) {
            const GExt& e = RES[order[i]];
            if (!gx_full(e)) break;
            bool ov = false;
            for (uint32_t prev = 0; prev < tail && !ov; ++prev) {
                const GExt& q = RES[order[prev]];
                ov = (double)gx_overlap(h, e, q) > pb.overlap * (double)(q.r1 - q.r0);
            }
            if (!ov) order[tail++] = order[i];
        }
        n_out = tail;
        for (uint32_t i = 0; i < n_out; ++i) { gx_find_mismatches(c, RES[order[i]], mm, overflow); mm_owner = order[i]; }      // counts, for the output sizes
        out.full_length = 1;
    } else {
        n_out = gx_remove_duplicates(RES, order, n_res);
        bool trimmed = false;
        for (uint32_t i = 0; i < n_out && !overflow; ++i) {
            gx_find_mismatches(c, RES[order[i]], mm, overflow); mm_owner = order[i];
            if (!overflow && (pb.flags & VGK_GAPLESS_TRIM)) trimmed |= gx_trim(c, RES[order[i]], mm);
        }
        if (trimmed) n_out = gx_remove_duplicates(RES, order, n_out);
    }
    if (overflow) { out.status = VGK_ETOOBIG; return; }
    // hand the set out (merged runs: in the ORIGINAL nodes the aligned interval touches — gx_expand)
    uint32_t nn = 0, nm = 0;
    for (uint32_t i = 0; i < n_out; ++i) { nn += (MG && P.merge.on) ? gx_expand(P, RES[order[i]], nullptr, nullptr) : RES[order[i]].path_len; nm += RES[order[i]].n_mism; }
    const unsigned long long e0 = g_bump(P.counters + 0, n_out), n0 = g_bump(P.counters + 1, nn), m0 = g_bump(P.counters + 2, nm);
    if (e0 + n_out > P.caps[0] || n0 + nn > P.caps[1] || m0 + nm > P.caps[2]) { out.status = VGK_EOPS; return; }
    out.ext_begin = (uint32_t)e0; out.n_ext = n_out;
    uint32_t na = 0, ma = 0;
    for (uint32_t i = 0; i < n_out; ++i) {
        const GExt& e = RES[order[i]];
        vgk_extension x;
        x.path_begin = (uint32_t)(n0 + na); x.path_len = e.path_len; x.offset = e.offset; x.read_begin = e.r0; x.read_end = e.r1;
        x.mism_begin = (uint32_t)(m0 + ma); x.n_mismatches = e.n_mism; x.score = e.score; x.left_full = e.left_full; x.right_full = e.right_full;
        x.pad[0] = x.pad[1] = 0;
        x.state[0] = (uint32_t)e.state.fn; x.state[1] = (uint32_t)e.state.flo; x.state[2] = (uint32_t)e.state.fhi;
        x.state[3] = (uint32_t)e.state.bn; x.state[4] = (uint32_t)e.state.blo; x.state[5] = (uint32_t)e.state.bhi;
        uint32_t plen = e.path_len;
        if (MG && P.merge.on) {
            uint32_t first_off = e.offset;
            plen = gx_expand(P, e, P.nodes + n0 + na, &first_off);
            x.path_len = plen; x.offset = first_off;
            if (plen) { x.state[0] = P.nodes[n0 + na + plen - 1]; x.state[3] = P.nodes[n0 + na] ^ 1u; }      // the states' nodes: the last / the first original node (their ranges run through a run unchanged)
        } else
        for (uint32_t k = 0; k < e.path_len; ++k) P.nodes[n0 + na + k] = (uint32_t)e.path[k];
        P.ext[e0 + i] = x;
        if (e.n_mism) {
            if (order[i] == mm_owner) for (uint32_t k = 0; k < e.n_mism; ++k) P.mism[m0 + ma + k] = mm[k];
            else { GExt& me = RES[order[i]]; bool ov = false; gx_find_mismatches(c, me, P.mism + m0 + ma, ov); }      // written straight to the output
        }
        na += plen; ma += e.n_mism;
    }
}

// one read, the nested form: every seed's search in turn (skipping seeds the best exact full-length extension so far contains), then
// the rules.  ST = where a seed's search lives (GStoreSlab / GStoreLds); the winners and the permutation sit in the thread's slab.
template <bool MG, class ST>
VGK_HD void gapless_extend_one(const GaplessParams& P, uint32_t pi, ST& Q, GScratch& S, GCold& C) {
    const GRes RES{S.res, C.res};
    const GProb pb = P.probs[pi];
    const GIndex& h = P.index;
    vgk_gapless_result& out = P.results[pi];
    out.status = VGK_OK; out.ext_begin = 0; out.n_ext = 0; out.full_length = 0;
    if (!pb.read_len || !pb.n_seeds) return;
    if (pb.n_seeds > (uint32_t)G_SEEDS) { out.status = VGK_ETOOBIG; return; }
    if (ST::FULL == G_RETRY && (pb.read_len > 255u || P.index.max_node_len > 255u || P.index.max_visits > 254u)) { out.status = G_RETRY; g_bump(P.counters + 3, 1); return; }   // beyond the compact entries
    GCtx c; c.P = &P; c.seq = P.reads + pb.read_off; c.L = pb.read_len;
    uint32_t n_res = 0, best_alignment = 0xffffffffu;
    int status = VGK_OK;
    for (uint32_t si = 0; si < pb.n_seeds && status == VGK_OK; ++si) {
        GSeedIn sd;
        if (!g_seed_in<MG>(P, pb.seed_off + si, sd)) { status = VGK_EINVAL; break; }
        if (best_alignment < n_res && RES[best_alignment].internal == 0 && gx_contains(h, RES[best_alignment], sd.node, sd.diff)) continue;
        GSearch s; s.prof = nullptr;
        GExt& r = RES[n_res];
        for (bool merged = true;; merged = false) {
            const int b = g_search_begin<MG>(P, c, pb, si, Q, s, merged);
            if (b != VGK_OK) { status = VGK_EINVAL; break; }
            while (status == VGK_OK && g_search_live(s)) status = g_search_step<MG>(P, c, Q, s);
            if (status != VGK_OK) break;
            status = g_search_end<MG>(P, Q, s, r);
            if (status != G_REDO) break;
            status = VGK_OK; g_bump(P.counters + 5, 1);                      // the search branched on the merged index: once more on the original one
        }
        if (status != VGK_OK) break;
        if (r.pad[0]) {
            if (gx_full(r) && (best_alignment >= n_res || r.internal < RES[best_alignment].internal)) best_alignment = n_res;
            ++n_res;
        }
    }
    if (status != VGK_OK) { out.status = status; if (status == G_RETRY) g_bump(P.counters + 3, 1); return; }
    gapless_set_rules<MG>(P, pi, pb, c, RES, n_res, best_alignment, S.order);
}

// ---- the flat form ----------------------------------------------------------------------------------------------------------------------
// One loop per lane: "no search in hand -> take the next seed of my read (or the next read of the batch) and begin; else one step".
// Measured on the bench's reads, the nested form keeps a quarter of a wavefront's lanes busy: searches take 5..60 steps and reads
// have 1..6 of them, and every lane waits for the slowest of its 64 at the end of every search and of every read.  Here a lane
// never waits for another's search.  The branch that begins a search is run for the whole wavefront only when `G_FLAT_MIN_IDLE`
// lanes want it (or nobody is stepping), so its cost is shared.  The winners go to HBM (P.winners, the read's seed_off + k for its
// k-th winner) and the rules over them run in a second kernel (gapless_rules_one), one read per lane: they are short and alike.
// Between the two, vgk_gapless_result carries the read's state: status, n_ext = winners, ext_begin = best_alignment.
constexpr uint32_t G_FLAT_MIN_IDLE = 12;
struct GWinArr { GExt* base; VGK_HD GExt& operator[](uint32_t i) const { return base[i]; } };
// W: next_read(P) -> position in P.order or 0xffffffff; vote(idle, searching) -> 0 = every lane is done, 1 = idle lanes act now, 2 = only step
template <bool MG, class ST, class W>
VGK_HD void gapless_search_lane(const GaplessParams& P, ST& Q, GScratch& S, W& wave) {
    const GIndex& h = P.index;
    constexpr uint32_t NONE = 0xffffffffu;
    uint32_t pi = NONE, si = 0, n_res = 0, best_alignment = NONE; int status = VGK_OK;
    GProb pb; pb.n_seeds = 0; pb.read_len = 0; pb.seed_off = 0; pb.read_off = 0; pb.max_mm = 0; pb.flags = 0; pb.overlap = 0;
    GCtx c; c.P = &P; c.seq = P.reads; c.L = 0;
    GSearch s; s.hn = 0; s.have_cand = false; s.prof = wave.prof();
    bool searching = false, done = false, redo = false;
    for (;;) {
        const bool idle = !searching && !done;
        const int v = wave.vote(idle, searching);
        if (v == 0) break;
        G_TICK(s.prof, 0);
        if (v == 1 && idle) {
            for (;;) {
                if (pi == NONE || status != VGK_OK || si >= pb.n_seeds) {
                    if (pi != NONE) {                                                        // this read's searches are over
                        vgk_gapless_result& out = P.results[pi];
                        out.status = status; out.n_ext = n_res; out.ext_begin = best_alignment; out.full_length = 0;
                        if (status == G_RETRY) { g_bump(P.counters + 3, 1); if (P.retry) P.retry[pi] = 1; }
                    }
                    const uint32_t k = wave.next_read(P);
                    if (k == NONE) { done = true; pi = NONE; break; }
                    pi = P.order[k]; pb = P.probs[pi]; si = 0; n_res = 0; best_alignment = NONE; status = VGK_OK;
                    c.seq = P.reads + pb.read_off; c.L = pb.read_len;
                    if (!pb.read_len || !pb.n_seeds) si = pb.n_seeds;
                    else if (pb.n_seeds > (uint32_t)G_SEEDS) status = VGK_ETOOBIG;
                    else if (ST::FULL == G_RETRY && (pb.read_len > 255u || h.max_node_len > 255u || h.max_visits > 254u)) status = G_RETRY;
                    continue;
                }
                GSeedIn sd;
                if (!g_seed_in<MG>(P, pb.seed_off + si, sd)) { status = VGK_EINVAL; continue; }
                if (best_alignment != NONE) {
                    const GExt& ba = P.winners[pb.seed_off + best_alignment];
                    if (ba.internal == 0 && gx_contains_diag(ba, S.diag, sd.node, sd.diff)) { ++si; continue; }
                }
                if (g_search_begin<MG>(P, c, pb, si, Q, s, !redo) != VGK_OK) { status = VGK_EINVAL; continue; }
                redo = false;
                ++si; searching = true;
                break;
            }
        }
        G_TICK(s.prof, 1);
        if (searching) {
            status = g_search_step<MG>(P, c, Q, s);
            if (status != VGK_OK) searching = false;
            else if (!g_search_live(s)) {
                searching = false;
                GExt& r = P.winners[pb.seed_off + n_res];
                status = g_search_end<MG>(P, Q, s, r);
                if (status == G_REDO) { status = VGK_OK; --si; redo = true; g_bump(P.counters + 5, 1); }          // the same seed again, on the original index
                else if (status == VGK_OK && r.pad[0]) {
                    if (gx_full(r) && (best_alignment == NONE || r.internal < P.winners[pb.seed_off + best_alignment].internal)) {
                        best_alignment = n_res;
                        if (r.internal == 0) gx_diagonals(h, r, S.diag);
                    }
                    ++n_res;
                }
                G_TICK(s.prof, 6);
            }
        }
    }
}
// the rules of one read over the winners its searches left
template <bool MG> VGK_HD void gapless_rules_one(const GaplessParams& P, uint32_t pi, uint8_t* order) {
    if (P.retry && P.retry[pi]) return;                                        // the slab kernel's read (it may be at work on it right now): hands off
    const GProb pb = P.probs[pi];
    vgk_gapless_result& out = P.results[pi];
    const uint32_t n_res = out.n_ext, best_alignment = out.ext_begin;
    out.ext_begin = 0; out.n_ext = 0; out.full_length = 0;
    if (out.status != VGK_OK || !pb.read_len || !pb.n_seeds) return;           // an error, or nothing to do
    GCtx c; c.P = &P; c.seq = P.reads + pb.read_off; c.L = pb.read_len;
    const GWinArr RES{P.winners + pb.seed_off};
    gapless_set_rules<MG>(P, pi, pb, c, RES, n_res, best_alignment, order);
}

// ---- the sets in problem order ------------------------------------------------------------------------------------------------------
// The rules kernel packs finished sets behind each other in completion order.  The caller gets them in problem order; that
// re-ordering was a pass over a million reads on the host with scattered copies into fresh pages (30 of the 55 ms of a call).  Here:
// per read the sizes of its set, three prefix sums, one gather — the host then copies three contiguous arrays.
struct GOrderParams {
    uint32_t n; const vgk_gapless_result* res; const vgk_extension* ext; const uint32_t* nodes; const uint32_t* mism;
    uint32_t* size_e; uint32_t* size_n; uint32_t* size_m;                  // [n + 1] each, the last entry 0: the scans' inputs
    const uint32_t* off_e; const uint32_t* off_n; const uint32_t* off_m;    // their exclusive prefix sums
    vgk_gapless_result* res_out; vgk_extension* ext_out; uint32_t* nodes_out; uint32_t* mism_out;
    uint32_t* read_of;                                                      // [extensions]: the read an extension belongs to (for vgk_tail_stage; nullable)
};
VGK_HD void g_order_sizes_one(const GOrderParams& P, uint32_t i) {
    const vgk_gapless_result r = P.res[i];
    uint32_t nn = 0, nm = 0;
    if (r.status == VGK_OK) for (uint32_t k = 0; k < r.n_ext; ++k) { nn += P.ext[r.ext_begin + k].path_len; nm += P.ext[r.ext_begin + k].n_mismatches; }
    P.size_e[i] = r.status == VGK_OK ? r.n_ext : 0u; P.size_n[i] = nn; P.size_m[i] = nm;
}
VGK_HD void g_order_gather_one(const GOrderParams& P, uint32_t i) {
    vgk_gapless_result r = P.res[i];
    const uint32_t src = r.ext_begin;
    r.ext_begin = P.off_e[i];
    if (r.status == VGK_OK) {
        uint32_t wn = P.off_n[i], wm = P.off_m[i];
        for (uint32_t k = 0; k < r.n_ext; ++k) {
            vgk_extension x = P.ext[src + k];
            for (uint32_t j = 0; j < x.path_len; ++j) P.nodes_out[wn + j] = P.nodes[x.path_begin + j];
            for (uint32_t j = 0; j < x.n_mismatches; ++j) P.mism_out[wm + j] = P.mism[x.mism_begin + j];
            x.path_begin = wn; x.mism_begin = wm;
            P.ext_out[r.ext_begin + k] = x;
            if (P.read_of) P.read_of[r.ext_begin + k] = i;
            wn += x.path_len; wm += x.n_mismatches;
        }
    } else r.n_ext = 0;
    P.res_out[i] = r;
}
// ReadMasker (src/gbwt_extender.cpp:160-176) on the device: anything but ACGT never matches
VGK_HD char g_mask_base(char c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'X'; }

// ---- clusters that never left the device (vgk_gapless_extend_seeded): the problem descriptors and the hand-out keys from what
//      vgk_minimizer_seeds left in HBM
struct GSeededParams {
    uint32_t n; const uint64_t* read_off;       // reads relative to the first; the read buffer carries 8 bytes of padding in front
    const uint32_t* seed_off; const vgk_seed* seeds;
    uint32_t max_mm, flags; double overlap; uint32_t buckets;
    GProb* probs; uint32_t* key; uint32_t* idx;
};
VGK_HD void g_seeded_one(const GSeededParams& P, uint32_t i) {
    GProb pb;
    if (P.read_off) {
        pb.read_off = (uint32_t)P.read_off[i] + 8u; pb.read_len = (uint32_t)(P.read_off[i + 1] - P.read_off[i]);
        pb.seed_off = P.seed_off[i]; pb.n_seeds = P.seed_off[i + 1] - P.seed_off[i];
        pb.max_mm = P.max_mm; pb.flags = P.flags; pb.overlap = P.overlap;
        P.probs[i] = pb;
    } else pb = P.probs[i];                            // the descriptors came from the host (vgk_gapless_extend): only the hand-out keys are made here
    const uint32_t v = pb.n_seeds ? P.seeds[pb.seed_off].node / 2u : P.buckets - 1u;       // the hand-out order: by the node of the first seed, reads without seeds last
    P.key[i] = v < P.buckets - 1u ? v : P.buckets - 1u; P.idx[i] = i;
}

}  // namespace vgk
