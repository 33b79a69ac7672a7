// gssw_device.hpp — the graph Smith-Waterman engine (gssw semantics) for gfx950.
//
// Replaces gssw_graph_fill_pinned + gssw_graph_trace_back* as called from
// Aligner::align_internal (reference: src/aligner.cpp:396-402, 423-435, 537-557).
//
// MI355X mapping (see DESIGN.md §3 for the full rationale):
//   * two reads share every 32-bit register (lo/hi unsigned 16-bit halves), so
//     each packed VALU instruction (v_pk_*_u16) advances two DP cells;
//   * a read pair is spread over G = ceil(L/K) adjacent lanes, lane g owning
//     K consecutive read rows (K = 16, 20 or 24, chosen per batch so that the
//     wavefront's 64 lanes and the padded rows are used best; 150 bp reads run
//     with K = 20: 8 lanes per pair, 8 pairs = 16 reads per wavefront);
//     floor(64/G) pairs share one wavefront;
//   * lanes run a skewed wavefront: at step t lane g computes graph column
//     t-g, taking the vertical-gap value F and the diagonal H of the row above
//     from lane g-1 with one DPP wave_shr:1 each — no LDS, no barriers;
//   * the unsigned, bias-shifted arithmetic and saturating subtracts reproduce
//     gssw's (SSW's) zero-floored E/F/H exactly;
//   * per cell a 4-bit traceback code is produced; K rows x 2 reads = K bytes per
//     lane per step, stored step-major so every wave store is one contiguous
//     burst of 64*K bytes;
//   * node boundaries that are not simple chain links go through a small
//     HBM scratch of per-node last columns (H and next-column E), which the
//     traceback also uses to choose predecessors.
//
// Everything in this header is plain C++ over pk16.hpp so that the identical
// lane code can be single-stepped by the CPU emulator in tests/emu.
#pragma once
#include <stdint.h>
#include "../../include/vgk.h"
#include "pk16.hpp"

namespace vgk {

constexpr int KMAX = 24;   // most read rows per lane any instantiation uses

// X-drop (dozeu) mode has no zero floor.  Its scores are carried with a constant offset so that the
// same unsigned saturating arithmetic applies: a value that saturates at 0 stands for "< -XOFF",
// which can never rejoin an optimal path as long as L*max_score + bonus < XOFF (checked at pack time).
constexpr uint32_t XOFF = 1023;

// per-column info byte
enum : uint32_t {
    CI_BASE_MASK  = 7,     // 0..3 = ACGT, 4 = N
    CI_NODE_START = 8,     // first column of a node
    CI_STORE_END  = 16,    // last column of a node whose last column must be saved to scratch
    CI_SEED_SLOW  = 32,    // first column of a node whose predecessors are not exactly {previous node}
    CI_INVALID    = 128    // no column (before the start / past the end of this read's graph)
};
constexpr uint32_t CI_INVALID2 = CI_INVALID | (CI_INVALID << 16);

struct ProbDesc {          // one per read
    uint32_t col_off;      // byte offset of this read's column-info stream (4-aligned)
    uint32_t R;            // graph columns
    uint32_t L;            // read length
    uint32_t read_off;     // byte offset of the read's base codes
    uint32_t node_off;     // first NodeRec
    uint32_t n_nodes;
    uint32_t scratch_off;  // u32 index of this read's boundary scratch (16-aligned)
    uint32_t n_slots;
    uint32_t flags;        // VGK_GSSW_*
    uint32_t ops_off;      // first vgk_op of this read's output window
    uint32_t ops_cap;
    uint32_t max_gap;      // XDROP: leading-insertion cells of the root column (dz_align_init), multiple of 8
    // where the fill put this read (reads are bucketed by length; each bucket has its own K x G geometry)
    uint32_t wave;         // wavefront (index into waves[])
    uint32_t lane0;        // first lane of the read pair's lane group inside that wavefront
    uint32_t geom;         // K | G << 8 | half << 16   (half: 0 = low 16-bit halves, 1 = high)
    uint32_t Lpad;         // G*K rows per scratch slot
    // scoring of this read: full-length bonuses actually in force (0 where the mode grants none; quality-adjusted
    // contexts take them from the end bases' qualities), and — quality-adjusted only — its per-row profile words
    uint32_t bonus_start, bonus_end;   // pre-multiplied by the score scale
    uint32_t prof_off;                 // first per-row profile dword in `prof`, or 0xffffffff (profile = prof4[base])
    uint32_t pad;
};

struct NodeRec {
    uint32_t col_start;    // first column
    uint32_t col_end;      // one past the last column
    uint32_t pred_begin;   // index into preds[]
    uint32_t n_pred;
    int32_t  slot;         // scratch slot or -1
    uint32_t pinning;      // 1 = pinning node (PINNED mode)
};

struct WaveDesc {
    uint64_t tb_off;       // first traceback dword of this wave (records of ceil(K/4) dwords, step-major)
    uint32_t n_steps;
    uint32_t first_pair;
    uint32_t G;            // lanes per read pair in this wavefront (length buckets differ)
    uint32_t pair_end;     // pairs of this wavefront's bucket end here
};

struct GsswParams {
    const ProbDesc* probs;
    const uint8_t*  colinfo;
    const uint8_t*  reads;
    const uint32_t* prof;       // quality-adjusted contexts: per read row 4 biased score bytes (ref A,C,G,T), bonus included
    const NodeRec*  nodes;
    const uint32_t* preds;
    uint32_t*       scratch;    // per (slot,row): lo16 = H of the node's last column, hi16 = E for the column after it
    uint32_t*       tb;         // ceil(K/4) dwords per (step, lane)
    const WaveDesc* waves;
    const uint32_t* order;      // read pairs: pair p = reads order[2p] (low halves) and order[2p+1] (high halves), 0xffffffff = none
    unsigned long long* best;   // LOCAL mode: per read, max over cells of key64(score, col, row)
    vgk_result*     results;
    vgk_op*         ops;
    uint32_t n_problems, n_pairs, n_waves;
    // per fill launch (one launch per rows-per-lane instantiation K; lanes-per-pair G varies per wavefront):
    uint32_t wave_begin, wave_count;   // waves [wave_begin, wave_begin + wave_count)
    uint32_t K;                 // read rows per lane (16, 20 or 24) of this fill launch
    uint32_t prof4[6];          // per read base q: byte r = matrix[5r+q] + bias, r = 0..3; [5] = 0 (X-drop row 0: consumes nothing)
    uint32_t bias;
    uint32_t go, ge;
    int32_t  bonus;             // full-length bonus (plain contexts; per-read values live in ProbDesc)
    int32_t  want_tb;           // any problem wants traceback -> store codes
    // The speculative fill (spec_fill != 0; needs walk_passes == 2, one fill launch, one lanes-per-pair geometry): the fill of ALL reads builds
    // no traceback codes (two thirds of the time); walk_diag_one settles the alignments that are one diagonal run from the end cells alone;
    // the reads it leaves — the miss list — are laid out as wavefronts of their own behind the batch's (refill_layout_one), filled again
    // WITH codes, and walked by them.  refill_wave0 / refill_pair0: where those wavefronts and their pairs start in waves[] / order[];
    // refill_slot: traceback dwords per such wavefront (sized for the batch's widest window); refill_count[0]: how many there are (device).
    int32_t  spec_fill;
    uint32_t key3;              // the speculative first fill may take the column key maximum with v_pk_maximum3_f16 (gssw_key3_ok: every key of the batch below 0x7c00)
    int32_t  restore_probs;     // this run does not speculate, an earlier run of the same batch did: the displaced reads' descriptors go back first (refill_restore_one)
    uint32_t refill_wave0, refill_pair0, refill_G, refill_K;
    unsigned long long refill_slot;
    uint32_t* refill_count;
    const uint32_t* wave_limit;  // a fill launch over wavefronts whose number only the device knows: wave_begin + *wave_limit is the end
    int32_t  walk_passes;       // 2: the tracebacks run as two kernels — walk_diag_one for every read (whole alignments that are one diagonal run, from the read's and the columns' bytes alone), then walk_one for the reads it left on the miss list; 1: walk_one for all
    int32_t  tb_mode;           // TB_CODES: the fill stores a 4-bit code per cell; TB_REWALK: it stores what the traceback needs to compute them again
                                // where the path runs (see "the traceback that does not tax the fill" below)
    int32_t  dbg;               // timing experiments (VGAMD_TB_DBG; results are wrong when set): 1 = the band kernel stops after its first column, 2 = it stores nothing
    int32_t  fused;             // 1 = each wavefront traces its own reads back at the end of the fill kernel
    uint32_t scale;             // 1 or 8: every DP quantity above (prof4, bias, go, ge, bonus, xoff, scratch) is pre-multiplied.
                                // With 8, non-zero score differences are >= 8, so min(diff, 1|2|4|8) yields the four traceback
                                // bits already weighted and they merge with full-rate ORs instead of v_pk_mad.
    uint32_t xoff;              // XOFF * scale
    int8_t   matrix[25];
};

VGK_HD uint32_t rep2(uint32_t x) { return (x & 0xffffu) * 0x00010001u; }

// The profile word of a read base — P.prof4[q], q per lane — chosen among five values in registers.  Written as a plain chain of selects over
// P.prof4[...] it compiles into a choice between the words' ADDRESSES and a vector load per use (from the parameter block, or from a stack copy):
// in the tracebacks' loops a dependent load per cell.  The empty asm makes the five values results of an instruction, which cannot be folded
// back into a load.
struct ProfWords {               // read base A, C, G, T, N: each the four biased scores against the graph's bases
    uint32_t w0, w1, w2, w3, w4;
    VGK_HD uint32_t of(uint32_t q) const {
        uint32_t a = w0, b = w1, c = w2, e = w3, f = w4;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f));
#endif
        uint32_t w = a; w = q == 1 ? b : w; w = q == 2 ? c : w; w = q == 3 ? e : w; w = q == 4 ? f : w;
        return w;
    }
    VGK_HD uint32_t of_row(uint32_t q) const { return q == 5 ? 0u : of(q); }      // (the fill's rows: code 5 = row 0 of an X-drop problem, which consumes nothing)
};
VGK_HD ProfWords prof_words(const GsswParams& P) { return ProfWords{P.prof4[0], P.prof4[1], P.prof4[2], P.prof4[3], P.prof4[4]}; }

// gssw: start bonus on read base 0; end bonus on base L-1 unless pinned (src/aligner.cpp:401-402).
// dozeu: one bonus, on consuming the last packed query base (row L-1 of the L = len+1 rows).
// The host resolves which apply (and their quality-adjusted values) into ProbDesc::bonus_start / bonus_end.
VGK_HD uint32_t row_bonus(uint32_t bonus_start, uint32_t bonus_end, uint32_t row, uint32_t L) {
    return (row == 0 ? bonus_start : 0u) + (row + 1 == L ? bonus_end : 0u);
}

VGK_HD unsigned long long key64(uint32_t score, uint32_t col, uint32_t row) {
    return ((unsigned long long)score << 40) | ((unsigned long long)(0xFFFFFu - col) << 20) | (unsigned long long)(0xFFFFFu - row);
}

// ---------------------------------------------------------------------------
// per-lane state of the fill
// ---------------------------------------------------------------------------
// VGK_PB_LDS (device builds of the fill kernels; tools/build_variant.sh): read B's profile words live in LDS, [row][lane] of the wavefront's own
// 64 K dwords (conflict-free: a row is one 256-byte line), instead of K registers per lane — the rows' only loop-invariant registers the
// allocator may give up without a spill.  Without it the K = 19 / 20 kernels keep 15 profile words in scratch and load them back every step.
#ifndef VGK_PB_LDS
#define VGK_PB_LDS 0
#endif
template <int K>
struct Lane {
    uint32_t H[K];      // H of the last processed column          {B:A}
    uint32_t E[K];      // E for the next column                   {B:A}
    uint32_t PA[K];     // query profile of read A: 4 biased bytes (ref A,C,G,T) per row
#if VGK_PB_LDS && defined(__HIPCC__)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* PBL;       // this lane's column of the wavefront's LDS block: row m at PBL[64 m] (a 32-bit LDS address: ds_read_b32 with the row as the instruction's offset)
    VGK_HD uint32_t pb(int m) const { return *(const volatile lds_u32*)(PBL + 64 * m); }      // (volatile: read where it is used — hoisted out of the step loop the K words would be registers again, and spilled)
    VGK_HD void set_pb(int m, uint32_t v) { PBL[64 * m] = v; }
#else
    uint32_t PB[K];
    VGK_HD uint32_t pb(int m) const { return PB[m]; }
    VGK_HD void set_pb(int m, uint32_t v) { PB[m] = v; }
#endif
    uint32_t out_h, out_f, info;    // handed to lane+1 at the next step
    uint32_t prev_rh;               // H of the row above this lane's block, previous column
    uint32_t best_lo, best_hi, step_lo, step_hi;
    uint32_t nodeA, nodeB;
    uint32_t g;                     // lane index inside the group
    uint32_t Lpad;                  // rows per scratch slot of this wavefront's bucket (G*K)
    uint32_t probA, probB;          // read indices or 0xffffffff
    uint32_t LA, LB, flagsA, flagsB;
    uint32_t bsA, beA, bsB, beB;    // full-length bonuses in force for read A / B
    uint32_t RA, RB, colA, colB;    // graph columns and column-info stream offsets
    uint32_t ciA, ciB, ciA_n, ciB_n;  // leader only: info bytes of columns [4k,4k+4) and the prefetched next word
    uint32_t one;                   // 0x00010001 kept opaque to the optimiser (see lane_rows)
};

// The word is made good for its four steps when it is loaded — columns from R on read CI_INVALID (the stream goes on with another read's bytes) —, so that
// a step takes its byte of both reads with ONE v_perm_b32 instead of two shifts, two compares against R and two selects (7 of ~270 VALU instructions per step).
VGK_HD uint32_t ci_word(uint32_t w, uint32_t t, uint32_t R) {
    const uint32_t none = CI_INVALID * 0x01010101u;
    if (t + 4u <= R) return w;
    if (t >= R) return none;
    const uint32_t keep = (1u << (8u * (R - t))) - 1u;                          // (R - t = 1 .. 3 columns left)
    return (w & keep) | (none & ~keep);
}
template <int K>
VGK_HD void lane_init(Lane<K>& s, const GsswParams& P, const WaveDesc& wd, uint32_t lane_id) {
    const uint32_t q = lane_id / wd.G;
    s.g = lane_id - q * wd.G;
    s.Lpad = wd.G * K;
    const uint32_t pair = wd.first_pair + q;
    const bool live = (q < 64u / wd.G) && (pair < wd.pair_end);
    s.probA = live ? P.order[2 * pair] : 0xffffffffu;
    s.probB = live ? P.order[2 * pair + 1] : 0xffffffffu;
    s.LA = s.LB = 0; s.flagsA = s.flagsB = 0; s.RA = s.RB = 0; s.colA = s.colB = 0;
    s.ciA = s.ciB = s.ciA_n = s.ciB_n = CI_INVALID * 0x01010101u;
    uint32_t roA = 0, roB = 0, poA = 0xffffffffu, poB = 0xffffffffu;
    s.bsA = s.beA = s.bsB = s.beB = 0;
    if (s.probA != 0xffffffffu) { const ProbDesc& d = P.probs[s.probA]; s.LA = d.L; s.flagsA = d.flags; roA = d.read_off; s.RA = d.R; s.colA = d.col_off;
                                   s.bsA = d.bonus_start; s.beA = d.bonus_end; poA = d.prof_off; }
    if (s.probB != 0xffffffffu) { const ProbDesc& d = P.probs[s.probB]; s.LB = d.L; s.flagsB = d.flags; roB = d.read_off; s.RB = d.R; s.colB = d.col_off;
                                   s.bsB = d.bonus_start; s.beB = d.bonus_end; poB = d.prof_off; }
    if (s.g == 0) {   // streams are padded with 8 readable bytes, so these loads never run off the arena
        if (s.probA != 0xffffffffu) s.ciA_n = ci_word(*(const uint32_t*)(P.colinfo + s.colA), 0, s.RA);
        if (s.probB != 0xffffffffu) s.ciB_n = ci_word(*(const uint32_t*)(P.colinfo + s.colB), 0, s.RB);
    }
    const ProfWords pw = prof_words(P);
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const uint32_t row = s.g * K + m;
        uint32_t pa = 0, pb = 0;
        if (row < s.LA) pa = poA != 0xffffffffu ? P.prof[poA + row] : pw.of_row(P.reads[roA + row]) + 0x01010101u * row_bonus(s.bsA, s.beA, row, s.LA);
        if (row < s.LB) pb = poB != 0xffffffffu ? P.prof[poB + row] : pw.of_row(P.reads[roB + row]) + 0x01010101u * row_bonus(s.bsB, s.beB, row, s.LB);
        s.PA[m] = pa; s.set_pb(m, pb); s.H[m] = 0; s.E[m] = 0;
    }
    s.out_h = 0; s.out_f = 0; s.info = CI_INVALID2; s.prev_rh = 0;
    s.best_lo = s.best_hi = 0; s.step_lo = s.step_hi = 0;
    s.nodeA = s.nodeB = 0xffffffffu;
    s.one = 0x00010001u;
}

// Group leaders (g == 0) pull the column-info stream 4 columns at a time, one
// word ahead, so the HBM/L2 latency of the load hides behind four steps of DP.
// Call at every step with (t & 3) == 0, before lane_step.
template <int K>
VGK_HD void lane_prefetch(Lane<K>& s, const GsswParams& P, uint32_t t) {
    if (s.g != 0) return;
    s.ciA = s.ciA_n; s.ciB = s.ciB_n;
    s.ciA_n = t + 4 < s.RA ? ci_word(*(const uint32_t*)(P.colinfo + s.colA + t + 4), t + 4, s.RA) : CI_INVALID * 0x01010101u;
    s.ciB_n = t + 4 < s.RB ? ci_word(*(const uint32_t*)(P.colinfo + s.colB + t + 4), t + 4, s.RB) : CI_INVALID * 0x01010101u;
}

// fresh column info for the group leader at step t: {infoB<<16 | infoA} — byte t & 3 of either word
template <int K>
VGK_HD uint32_t fetch_info(const Lane<K>& s, const GsswParams& P, uint32_t t) {
    (void)P;
    return byte_perm(s.ciB, s.ciA, 0x0c040c00u + (t & 3u) * 0x00010001u);
}

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ uint32_t scratch_load(const uint32_t* p) {
    // written earlier in this kernel, possibly by the neighbouring lane: bypass the CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
static inline uint32_t scratch_load(const uint32_t* p) { return *p; }
#endif

// SEED_SLOW: seed column = element-wise max over the predecessors' saved last
// columns (gssw_create_seed_*); HALF = 0 for read A (low halves), 1 for read B.
template <int HALF, int K>
VGK_HD void seed_from_scratch(Lane<K>& s, const GsswParams& P, uint32_t prob, uint32_t node, uint32_t& diag0) {
    const ProbDesc& d = P.probs[prob];
    const NodeRec& nr = P.nodes[d.node_off + node];
    const uint32_t keep = HALF == 0 ? 0xffff0000u : 0x0000ffffu;     // the other read's half stays
#pragma unroll
    for (int m = 0; m < K; ++m) { s.H[m] &= keep; s.E[m] &= keep; }
    diag0 &= keep;
    if (nr.n_pred == 0 && (d.flags & 15u) == VGK_XDROP_PINNED) {
        // dozeu root column (dz_align_init): nothing consumed = 0, i leading inserted bases = -(go + (i-1) ge)
        // for i <= max_gap cells, unreachable beyond; E of the next column opens a deletion from it.
#pragma unroll
        for (int m = 0; m <= K; ++m) {
            const uint32_t row = s.g * K + m - 1;               // m = 0 is the row above this lane's block
            uint32_t h = 0;
            if (m > 0 || s.g > 0) {
                if (row == 0) h = P.xoff;
                else if (row <= d.max_gap) { const uint32_t pen = P.go + (row - 1) * P.ge; h = pen < P.xoff ? P.xoff - pen : 0; }
            }
            if (m == 0) { diag0 |= HALF == 0 ? h : (h << 16); }
            else {
                const uint32_t e = h > P.go ? h - P.go : 0;
                s.H[m - 1] |= HALF == 0 ? h : (h << 16);
                s.E[m - 1] |= HALF == 0 ? e : (e << 16);
            }
        }
        return;
    }
    for (uint32_t k = 0; k < nr.n_pred; ++k) {
        const NodeRec& pr = P.nodes[d.node_off + P.preds[nr.pred_begin + k]];
        const uint32_t* base = P.scratch + d.scratch_off + (uint32_t)pr.slot * s.Lpad + s.g * K;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const uint32_t v = scratch_load(base + m);     // lo16 = H, hi16 = E-next of the predecessor's last column
            // a packed max against a value whose other half is 0 leaves the other read untouched
            s.H[m] = pk_max(s.H[m], HALF == 0 ? (v & 0xffffu) : (v << 16));
            s.E[m] = pk_max(s.E[m], HALF == 0 ? (v >> 16) : (v & 0xffff0000u));
        }
        if (s.g > 0) {
            const uint32_t v = scratch_load(base - 1);
            diag0 = pk_max(diag0, HALF == 0 ? (v & 0xffffu) : (v << 16));
        }
    }
}

template <int HALF, int K>
VGK_HD void store_to_scratch(const Lane<K>& s, const GsswParams& P, uint32_t prob, uint32_t node) {
    const ProbDesc& d = P.probs[prob];
    const NodeRec& nr = P.nodes[d.node_off + node];
    uint32_t* base = P.scratch + d.scratch_off + (uint32_t)nr.slot * s.Lpad + s.g * K;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        uint32_t h = HALF == 0 ? (s.H[m] & 0xffffu) : (s.H[m] >> 16);
        uint32_t e = HALF == 0 ? (s.E[m] & 0xffffu) : (s.E[m] >> 16);
        base[m] = h | (e << 16);
    }
}

// Where (step t, lane)'s traceback record lies.  Shipped: STEP-MAJOR (VGK_TB_TILED=0: every wave store one contiguous burst of 64*K
// bytes; a walker's consecutive cells lie 64*K bytes apart).  Round 3 built and measured the TILED form (VGK_TB_TILED=1, tests and the
// emulator run with either): walk 4.54 -> 3.75 ms per million reads, but the fill 20.09 -> 21.94 ms although its tiles go through LDS and
// leave as whole bursts — a net loss of 1.05 ms per step, so it stays a build option.  Tiled: the records of TB_TILE = 8
// consecutive steps of a lane lie next to each other, split in two parts: part A = the record's first four dwords (rows 0-15 of the lane; 8 steps x 16 B = one 128-byte line per
// lane), part B = the rest (K = 19 / 20: one dword, K = 24: two).  A tile is [64 lanes x 8 steps x 16 B][64 lanes x 8 steps x part B].
// The walk moves one step back per cell and stays in its lane for K rows: the eight cells it fetches together lie in one or two lines
// instead of eight.  The fill writes a tile through LDS (backend_hip.hip) so that its global stores stay whole bursts.
#ifndef VGK_TB_TILED
#define VGK_TB_TILED 0
#endif
struct alignas(16) VgkU4 { uint32_t v[4]; };
constexpr uint32_t TB_TILE = VGK_TB_TILED ? 8 : 1;      // consecutive steps of a lane that lie next to each other
VGK_HD uint64_t tb_tile_base(uint64_t tb_off, uint32_t t, uint32_t rec_dwords) { return tb_off + (uint64_t)(t / TB_TILE) * (64u * TB_TILE * rec_dwords); }
VGK_HD uint64_t tb_dword(uint64_t tb_off, uint32_t t, uint32_t lane, uint32_t rec_dwords, uint32_t j) {
    if (TB_TILE == 1) return tb_off + ((uint64_t)t * 64u + lane) * rec_dwords + j;
    const uint64_t base = tb_tile_base(tb_off, t, rec_dwords);
    const uint32_t at = lane * TB_TILE + t % TB_TILE;
    return j < 4u ? base + at * 4u + j : base + 64u * TB_TILE * 4u + at * (rec_dwords - 4u) + (j - 4u);
}

// Build switches of the fill's hot row (kernel experiments: tools/build_variant.sh NAME -DVGK_...=1):
//   VGK_ACC_SHLOR  the 4-bit codes of four rows are merged by v_lshl_or_b32 instead of v_pk_mad_u16 (exact: a half holds at most 16 bits of
//                  codes, nothing crosses into the other read's half): off — measured no gain (20.00 vs 20.01 ms; with a constant shift
//                  v_lshl_or_b32 issues at the packed ops' rate, tools/pkmax3_check.hip)
//   VGK_H_MAX3     H = max(diagonal, E, F) as one v_pk_maximum3_f16 on the bit patterns (pk16.hpp) instead of two v_pk_max_u16: ON — exact on
//                  the device for every triple of halves below 0x7c00 (tools/pkmax3_check.hip: 0 of 1.2e10, denormals included), fill
//                  20.0 -> 19.15 ms per million reads (profiles/r04)
#ifndef VGK_ACC_SHLOR
#define VGK_ACC_SHLOR 0
#endif
#ifndef VGK_H_MAX3
#define VGK_H_MAX3 1
#endif
//   K3 (a template flag, not a build switch; GsswParams::key3): the speculative first fill's column key maximum over row PAIRS — one v_pk_maximum3_f16 per two
//                  rows instead of two v_pk_max_u16 — when every key of the batch stays below 0x7c00 (scores up to 990, no X-drop offset; the packers decide):
//                  hot block 243 -> 228 instructions, first fill 12.80 -> 12.39 ms per million reads (profiles/r06/ab_key3)

// Whether every end-cell key of a batch stays below 0x7c00, where v_pk_maximum3_f16 on the bit patterns is the unsigned maximum (pk16.hpp): the x8 scale, plain
// scoring, no problem in the X-drop mode (its scores carry an offset of 1023), and the longest read's best possible score at most 990 (990 * 32 + 31 < 0x7c00).
VGK_HD bool gssw_key3_ok(uint32_t scale, bool quality_adjusted, bool any_xdrop, uint32_t longest_read, uint32_t max_score, uint32_t max_bonus) {
    return scale == 8u && !quality_adjusted && !any_xdrop && (uint64_t)longest_read * max_score + 2ull * max_bonus <= 990ull;
}
// best-cell key of a row: score*32 + (31 - row_in_lane), so one packed max keeps
// the best score and, on ties, the smallest row (scores stay below 2047).
constexpr uint32_t KEY_SHIFT = 5, KEY_LOW = 31;

// One row (compile-time index M) of one lane for one column.  REFN = some half sees
// an N in the graph (rare): the profile permute cannot express score 0, patch it.
// max(0, diagonal + s) of row M, d = H of the row above in the previous column (the sum stays far below 2^16 per half).  With the tagged
// candidates of the x8 traceback build the constant absorbs the diagonal's tag.
template <int K, int M, bool REFN, bool S8, bool TB>
VGK_HD uint32_t row_diag(const Lane<K>& s, const GsswParams& P, uint32_t sel, uint32_t bias2, bool nA, bool nB, uint32_t d) {
    uint32_t sb = byte_perm(s.pb(M), s.PA[M], sel);
    if (REFN) {
        const uint32_t row = s.g * K + M;
        if (nA) sb = set_lo(sb, row < s.LA ? P.bias + row_bonus(s.bsA, s.beA, row, s.LA) : 0u);
        if (nB) sb = set_hi(sb, row < s.LB ? P.bias + row_bonus(s.bsB, s.beB, row, s.LB) : 0u);
    }
    return pk_subs(pk_add_nc(d, sb), (TB && S8) ? bias2 - 0x00040004u : bias2);
}
// `d` comes in as THIS row's diagonal candidate (row_diag) and leaves as the next row's, made from this row's old H before the new one is
// written: the old value's last use then precedes the new value's definition, H[M] is updated in its register, and the step loop carries no
// copy per row (the rotation "new H[M] lives where old H[M - 1] did" cost K - 1 v_mov_b32 per step: 18 of ~290 VALU instructions).
template <int K, int M, bool REFN, bool S8, bool TB, bool K3, bool NK>
VGK_HD void lane_row(Lane<K>& s, const GsswParams& P, uint32_t sel, uint32_t bias2, uint32_t go2, uint32_t ge2,
                     bool nA, bool nB, uint32_t& f, uint32_t& d, uint32_t* acc, uint32_t& ck) {
    const uint32_t t4 = d;
    if constexpr (M + 1 < K) d = row_diag<K, M + 1, REFN, S8, TB>(s, P, sel, bias2, nA, nB, s.H[M]);
    if constexpr (!TB) {
        // The recurrence alone (TB_REWALK fills): no tags, no codes.  Its H / E / F are the tagged build's with the three tag bits
        // stripped — the tags are below the x8 scale's resolution and every constant that absorbs one is a multiple of 8 away from the
        // clean one — so a traceback that runs the tagged code again from this fill's boundary values reproduces the codes bit for bit.
        const uint32_t e = s.E[M];
        const uint32_t h = (S8 && VGK_H_MAX3) ? pk_max3_f16(t4, e, f) : pk_max(pk_max(t4, e), f);
        const uint32_t gg = pk_subs(h, go2);
        const uint32_t en = pk_max(gg, pk_subs(e, ge2)), fn = pk_max(gg, pk_subs(f, ge2));
        const uint32_t key = pk_mad_add_imm<(int)KEY_LOW - M>(h, 0x00010001u << (S8 ? KEY_SHIFT - 3 : KEY_SHIFT));
        if constexpr (S8 && K3) {
            // (acc[] is free in this build and carries the even row's key to the odd one)
            if constexpr ((M & 1) == 0) { if constexpr (M + 1 < K) acc[0] = key; else ck = M == 0 ? key : pk_max(ck, key); }
            else ck = M == 1 ? pk_max(acc[0], key) : pk_max3_f16(ck, acc[0], key);
        } else ck = M == 0 ? key : pk_max(ck, key);
        s.H[M] = h; s.E[M] = en; f = fn;
        return;
    }
    if constexpr (S8) {
        // Scores scaled by 8 leave the three low bits of every value free, and they survive the subtraction of (scaled) constants.
        // Candidates carry a tag there, so the source of a maximum is read off the maximum instead of being recomputed from four
        // differences: H candidates 4 (diagonal) > 3 (E) > <= 1 (F) — ties go diagonal, then E, as in the untagged build —, gap
        // candidates 1 (opened from H) > 0 (extended) — ties go to the open.  The constants absorb the tags (t - (bias - 4) etc.);
        // E and F inputs are normalised by an OR before they are extended; H is stored clean (it feeds the diagonal sum, the
        // end-cell key and the next gap open).  A saturated 0 has no tag: cells worth 0 are never walked.
        const uint32_t ei = s.E[M] | 0x00030003u;
        const uint32_t h = VGK_H_MAX3 ? pk_max3_f16(t4, ei, f) : pk_max(pk_max(t4, ei), f);
        const uint32_t hc = h & 0xfff8fff8u;
        const uint32_t gg = pk_subs(hc, go2 - 0x00010001u);
        const uint32_t e2 = pk_subs(ei, ge2 + 0x00030003u), f2 = pk_subs(f | 0x00010001u, ge2 + 0x00010001u);
        const uint32_t en = pk_max(gg, e2), fn = pk_max(gg, f2);
        // traceback code: bit0 = next-column E was opened, bits 1-2 = H source (2 diagonal, 1 E, 0 F), bit3 = next-row F was opened
        uint32_t code = bit_select<0x00010001u>(en, h) & 0x00070007u;
        code = shl_or<3>(fn & 0x00010001u, code);
        acc[M >> 2] = (M & 3) == 0 ? code : (VGK_ACC_SHLOR ? shl_or<4>(acc[M >> 2], code) : pk_mul_add_imm<16>(acc[M >> 2], code));
        if constexpr (!NK) {                                                       // (NK: the second fill of a speculative batch — the end cells are the first fill's)
            const uint32_t key = pk_mad_add_imm<(int)KEY_LOW - M>(hc, 0x00010001u << (KEY_SHIFT - 3));
            ck = M == 0 ? key : pk_max(ck, key);
        }
        s.H[M] = hc; s.E[M] = en; f = fn;
        return;
    }
    const uint32_t e = s.E[M];
    const uint32_t h = pk_max(pk_max(t4, e), f);
    const uint32_t gg = pk_subs(h, go2);                   // max(0, H - go)
    const uint32_t e2 = pk_subs(e, ge2), f2 = pk_subs(f, ge2);
    const uint32_t en = pk_max(gg, e2), fn = pk_max(gg, f2);
    // traceback code: bit0 = H not from diagonal, bit1 = H not from E (then F),
    // bit2 = next-column E is an extension, bit3 = next-row F is an extension.
    // min(x, 1) with an opaque `one` keeps each flag at 2 ops; the merges are v_pk_mad_u16.
    const uint32_t one = s.one;
    // every difference below has a >= b in both halves (h, en, fn are maxima over the subtrahend), so the
    // full-rate 32-bit subtract is exact on the packed pair
    const uint32_t nd = pk_min(pk_sub_nb(h, t4), one);
    const uint32_t ne = pk_min(pk_sub_nb(h, e), one);
    const uint32_t eb = pk_min(pk_sub_nb(en, gg), one);     // next-column E != H - go  <=>  extension won strictly
    const uint32_t fb = pk_min(pk_sub_nb(fn, gg), one);
    uint32_t code = pk_mul_add_imm<2>(ne, nd); code = pk_mul_add_imm<4>(eb, code); code = pk_mul_add_imm<8>(fb, code);
    acc[M >> 2] = (M & 3) == 0 ? code : pk_mul_add_imm<16>(acc[M >> 2], code);
    const uint32_t key = pk_mad_add_imm<(int)KEY_LOW - M>(h, 0x00010001u << KEY_SHIFT);
    ck = M == 0 ? key : pk_max(ck, key);
    s.H[M] = h; s.E[M] = en; f = fn;
}

template <int K, int M, bool REFN, bool S8, bool TB, bool K3, bool NK>
VGK_HD void lane_rows_from(Lane<K>& s, const GsswParams& P, uint32_t sel, uint32_t bias2, uint32_t go2, uint32_t ge2,
                           bool nA, bool nB, uint32_t& f, uint32_t& d, uint32_t* acc, uint32_t& ck) {
    lane_row<K, M, REFN, S8, TB, K3, NK>(s, P, sel, bias2, go2, ge2, nA, nB, f, d, acc, ck);
    if constexpr (M + 1 < K) lane_rows_from<K, M + 1, REFN, S8, TB, K3, NK>(s, P, sel, bias2, go2, ge2, nA, nB, f, d, acc, ck);
}

// The K rows of one lane for one column; returns the K/4 traceback dwords and the column key maximum.
template <int K, bool REFN, bool S8, bool TB, bool K3, bool NK>
VGK_HD void lane_rows(Lane<K>& s, const GsswParams& P, uint32_t sel, uint32_t diag0, uint32_t rf,
                      bool nA, bool nB, uint32_t* acc, uint32_t& colkey) {
    uint32_t bias2 = rep2(P.bias), go2 = rep2(P.go), ge2 = rep2(P.ge);
#if defined(__HIP_DEVICE_COMPILE__)
    // The rare N variant must stay a separate branch: with opaque copies of its inputs the
    // optimiser cannot hoist "common" permutes/subtracts of all rows above the branch
    // (that hoisting cost 32 live VGPRs and spilled the hot loop).
    if (REFN) asm volatile("" : "+v"(sel), "+v"(bias2), "+v"(go2), "+v"(ge2));
#endif
    uint32_t f = rf, d = row_diag<K, 0, REFN, S8, TB>(s, P, sel, bias2, nA, nB, diag0), ck = 0;
    lane_rows_from<K, 0, REFN, S8, TB, K3, NK>(s, P, sel, bias2, go2, ge2, nA, nB, f, d, acc, ck);
    s.out_h = s.H[K - 1]; s.out_f = f;
    colkey = ck;
}

// One column of one lane: rh / rf = H and F of the row above this lane's block in this column, rinfo = the column bytes of the pair.
// tb_a / tb_b = where the ceil(K/4)-dword traceback record goes — its first four dwords and the rest (the two parts of the tiled
// layout; in the step-major form tb_b = tb_a + 4) —, or nullptr.  TB = build the codes; RE = the traceback's recomputation of a
// window (no end-cell tracking, no scratch stores: the fill has done both).
// AV: every lane of the wavefront has a column of both its reads at this step (the steady middle of a fill: steps G - 1 .. the shortest window's
// end) — nothing is tested for being there.
template <int K, bool S8, bool TB, bool RE, bool K3 = false, bool NK = false, bool AV = false>
VGK_HD void lane_column(Lane<K>& s, const GsswParams& P, uint32_t t, uint32_t rh, uint32_t rf, uint32_t rinfo, uint32_t* tb_a, uint32_t* tb_b) {
    s.info = rinfo;
    const uint32_t ia = rinfo & 0xffu, ib = (rinfo >> 16) & 0xffu;
    const bool vA = AV || !(ia & CI_INVALID), vB = AV || !(ib & CI_INVALID);
    if (AV || vA || vB) {
        uint32_t diag0 = s.prev_rh;
        if (vA && (ia & CI_NODE_START)) s.nodeA += 1;
        if (vB && (ib & CI_NODE_START)) s.nodeB += 1;
        if (rinfo & (CI_SEED_SLOW * 0x00010001u)) {                                // (one test for the pair: a column that is not there carries no flag)
            if (vA && (ia & CI_SEED_SLOW)) seed_from_scratch<0, K>(s, P, s.probA, s.nodeA, diag0);
            if (vB && (ib & CI_SEED_SLOW)) seed_from_scratch<1, K>(s, P, s.probB, s.nodeB, diag0);
        }
        // selector: byte0 <- PA[baseA], byte2 <- PB[baseB] (bytes 4..7 of the permute), bytes 1,3 <- 0
        const uint32_t sel = (rinfo & 0x00030003u) | 0x0c040c00u;
        uint32_t acc[(K + 3) / 4], colkey;
        // (an N is base code 4, the only code with bit 2 set; a column that is not there reads CI_INVALID, whose base bits are 0)
        if (rinfo & 0x00040004u) { const bool nA = (ia & CI_BASE_MASK) == 4, nB = (ib & CI_BASE_MASK) == 4; lane_rows<K, true, S8, TB, K3, NK>(s, P, sel, diag0, rf, nA, nB, acc, colkey); }
        else          lane_rows<K, false, S8, TB, K3, NK>(s, P, sel, diag0, rf, false, false, acc, colkey);
        if constexpr (TB) {
            if (tb_a) {
                if (TB_TILE > 1 && !RE) {                                          // part A is a 16-byte slot: one store
                    VgkU4 v; v.v[0] = acc[0]; v.v[1] = acc[1]; v.v[2] = acc[2]; v.v[3] = acc[3];
                    *reinterpret_cast<VgkU4*>(tb_a) = v;
#pragma unroll
                    for (int j = 4; j < (K + 3) / 4; ++j) tb_b[j - 4] = acc[j];
                } else {
#pragma unroll
                    for (int j = 0; j < (K + 3) / 4; ++j) { if (j < 4) tb_a[j] = acc[j]; else tb_b[j - 4] = acc[j]; }
                }
            }
        }
        if constexpr (!RE) {
            if constexpr (!NK) {
            // local end cell: first column with the best score, smallest row (SSW end_ref/end_read rule)
            const uint32_t klo = colkey & 0xffffu, khi = colkey >> 16;
            const bool upA = vA & ((klo >> KEY_SHIFT) > (s.best_lo >> KEY_SHIFT)), upB = vB & ((khi >> KEY_SHIFT) > (s.best_hi >> KEY_SHIFT));   // selects, no branches
            s.best_lo = upA ? klo : s.best_lo; s.step_lo = upA ? t : s.step_lo;
            s.best_hi = upB ? khi : s.best_hi; s.step_hi = upB ? t : s.step_hi;
            }
            if (rinfo & (CI_STORE_END * 0x00010001u)) {
                if (vA && (ia & CI_STORE_END)) store_to_scratch<0, K>(s, P, s.probA, s.nodeA);
                if (vB && (ib & CI_STORE_END)) store_to_scratch<1, K>(s, P, s.probB, s.nodeB);
            }
        }
    } else {
        s.out_h = 0; s.out_f = 0;
    }
    s.prev_rh = rh;
}

// The steps of a wavefront at which EVERY lane has a column of both its reads: [G - 1, the shortest window's columns) — empty when a lane has no pair
// or a pair has one read (R = 0).  `shortest` = the minimum of RA and RB over the wavefront's lanes.
VGK_HD void steady_steps(uint32_t G, uint32_t shortest, uint32_t n_steps, uint32_t& from, uint32_t& to) {
    from = G - 1u < n_steps ? G - 1u : n_steps; to = shortest < n_steps ? shortest : n_steps; if (to < from) to = from;
}
// One step of one lane of the fill.  rh/rf/rinfo are lane-1's out_h/out_f/info from the
// previous step (ignored by group leaders, which start a fresh column).
template <int K, bool S8, bool TB = true, bool K3 = false, bool NK = false, bool AV = false>
VGK_HD void lane_step(Lane<K>& s, const GsswParams& P, uint32_t t, uint32_t rh, uint32_t rf, uint32_t rinfo, uint32_t* tb_a, uint32_t* tb_b) {
    if (s.g == 0) { rh = 0; rf = 0; rinfo = fetch_info(s, P, t); }
    lane_column<K, S8, TB, false, K3, NK, AV>(s, P, t, rh, rf, rinfo, tb_a, tb_b);
}

// after the last step: publish this lane's best cell (LOCAL mode)
template <int K>
VGK_HD bool lane_best(const Lane<K>& s, int half, uint32_t& prob, unsigned long long& key) {
    const uint32_t b = half ? s.best_hi : s.best_lo, st = half ? s.step_hi : s.step_lo;
    prob = half ? s.probB : s.probA;
    if (prob == 0xffffffffu || (b >> KEY_SHIFT) == 0) return false;
    const uint32_t L = half ? s.LB : s.LA, flags = half ? s.flagsB : s.flagsA;
    if ((flags & 15u) == VGK_GSSW_PINNED) return false;
    const uint32_t row = s.g * K + (KEY_LOW - (b & KEY_LOW));
    if (row >= L) return false;   // cannot win, but never report a padding row
    key = key64(b >> KEY_SHIFT, st - s.g, row);
    return true;
}

// ---------------------------------------------------------------------------
// traceback walker: one thread per read
// ---------------------------------------------------------------------------
#ifndef VGK_WALK_SPEC
#define VGK_WALK_SPEC 8
#endif
constexpr uint32_t W_SPEC = VGK_WALK_SPEC;      // diagonal cells fetched together
struct Walker {
    static constexpr uint32_t SPEC = W_SPEC;
    const GsswParams& P; const ProbDesc& d; uint32_t half, lane0, K; uint64_t tb_off;
    // bit0 = H not from the diagonal, bit1 = H from F (else E), bit2 = next-column E is an extension, bit3 = next-row F is an extension
    VGK_HD uint32_t code(uint32_t r, uint32_t c) const {
        const uint32_t g = r / K, m = r - g * K, t = c + g, j = m >> 2, i = m & 3u;
#if defined(VGK_WALK_EXP) && VGK_WALK_EXP >= 1      // timing experiments only (results are wrong): the codes come out of a region that stays in L2 (1) / L1 (2, 3)
        const uint32_t w = P.tb[tb_off + (tb_dword(0, t, lane0 + g, (K + 3) >> 2, j) & (VGK_WALK_EXP == 1 ? 0x3ffffu : 0xfffu))];
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VGK_WALK_PLAIN_LOADS)
        // a walk uses 4 bits of a line, once: streaming loads.  The line still comes whole (FETCH_SIZE unchanged: 5.73 vs 5.68 M KiB per
        // 400 000 reads) but does not push anything else out of the caches: 2.146 -> 2.019 ms (profiles/r04/walk_nontemporal.txt)
        const uint32_t w = __builtin_nontemporal_load(&P.tb[tb_dword(tb_off, t, lane0 + g, (K + 3) >> 2, j)]);
#else
        const uint32_t w = P.tb[tb_dword(tb_off, t, lane0 + g, (K + 3) >> 2, j)];
#endif
#endif
        const uint32_t last = (4 * j + 3 < K ? 4 * j + 3 : K - 1) - 4 * j;      // a lane's last dword holds K % 4 rows when K is no multiple of 4
        const uint32_t raw = (w >> (16 * half + 4 * (last - i))) & 15u;
        if (P.scale != 8) return raw;
        // the x8 build stores the tags of the maxima (lane_row): bit0 = E opened, bits 1-2 = H source (2 diagonal, 1 E, 0 F), bit3 = F opened
        const uint32_t src = (raw >> 1) & 3u;
        return (src != 2u ? 1u : 0u) | (src == 0u ? 2u : 0u) | ((raw & 1u) ? 0u : 4u) | ((raw & 8u) ? 0u : 8u);
    }
    // aligned-dword caches of the read codes and the column-info bytes: the walk moves one
    // row / one column at a time, so each cached word serves up to four steps
    mutable uint32_t rd_word = 0, rd_base = 0xffffffffu, ci_word = 0, ci_base = 0xffffffffu;
    const ProfWords pw = prof_words(P);
    VGK_HD uint32_t read_code(uint32_t r) const {
        const uint32_t a = d.read_off + r, b = a & ~3u;
        if (b != rd_base) { rd_base = b; rd_word = *(const uint32_t*)(P.reads + b); }
        return (rd_word >> (8 * (a & 3u))) & 0xffu;
    }
    VGK_HD uint32_t col_base(uint32_t c) const {
        const uint32_t a = d.col_off + c, b = a & ~3u;
        if (b != ci_base) { ci_base = b; ci_word = *(const uint32_t*)(P.colinfo + b); }
        return (ci_word >> (8 * (a & 3u))) & CI_BASE_MASK;
    }
    VGK_HD int32_t score(uint32_t r, uint32_t c) const {
#if defined(VGK_WALK_EXP) && VGK_WALK_EXP == 3      // timing experiment: no loads of read / column bytes either
        return (int32_t)((r ^ c) & 1u) * 5 - 4;
#endif
        const uint32_t base = col_base(c);
        const int32_t bonus = (int32_t)row_bonus(d.bonus_start, d.bonus_end, r, d.L);
        if (base >= 4) return bonus;                      // N scores 0 (+ bonus)
        if (d.prof_off != 0xffffffffu)                    // quality-adjusted: per-row profile word, bonus already inside
            return (int32_t)((P.prof[d.prof_off + r] >> (8 * base)) & 0xffu) - (int32_t)P.bias;
        const uint32_t q = read_code(r);
        // profile word of read base q (wave-uniform table, per-thread select), byte = reference base
        return (int32_t)((pw.of(q) >> (8 * base)) & 0xffu) - (int32_t)P.bias + bonus;
    }
    VGK_HD uint32_t saved(const NodeRec& n, uint32_t r) const { return P.scratch[d.scratch_off + (uint32_t)n.slot * d.Lpad + r]; }
    VGK_HD int32_t saved_e(const NodeRec& n, uint32_t r) const {      // E for the next column, without the x8 build's "opened" tag
        const uint32_t v = saved(n, r) >> 16;
        return (int32_t)(P.scale == 8 ? v & ~7u : v);
    }
};

// Where a read's traceback starts: the pinned end (best pinning node's last column, row L - 1) or the best cell the fill published.
// SAVED(node, row) = H of that node's last column.  Returns false when there is nothing to walk from.
template <class SAVED>
VGK_HD bool walk_end_cell(const GsswParams& P, const ProbDesc& d, unsigned long long best_key, SAVED saved, int32_t& cur, uint32_t& c, uint32_t& node, int32_t& r) {
    const NodeRec* nodes = P.nodes + d.node_off;
    bool have = false;
    cur = 0; c = 0; node = 0; r = 0;
    if ((d.flags & 15u) == VGK_GSSW_PINNED) {
        r = (int32_t)d.L - 1;
        for (uint32_t n = 0; n < d.n_nodes; ++n) {
            if (!nodes[n].pinning) continue;
            const int32_t v = (int32_t)(saved(nodes[n], (uint32_t)r) & 0xffffu);
            if (!have || v > cur) { cur = v; node = n; have = true; }
        }
        if (have) c = nodes[node].col_end - 1;
    } else {
        const unsigned long long k = best_key;
        cur = (int32_t)(k >> 40) * (int32_t)P.scale;
        if (cur > 0) {
            have = true;
            c = 0xFFFFFu - (uint32_t)((k >> 20) & 0xFFFFFu);
            r = (int32_t)(0xFFFFFu - (uint32_t)(k & 0xFFFFFu));
            uint32_t lo = 0, hi = d.n_nodes;   // node containing column c
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nodes[mid].col_start <= c) lo = mid; else hi = mid; }
            node = lo;
        }
    }
    return have;
}

constexpr uint32_t W_MISS = 0x100u;            // a code a walker cannot serve (BandWalker: the cell lies outside the recomputed band)
constexpr int32_t  W_MISSED = 100;             // vgk_result::status of a read whose walk met one: the on-demand traceback takes it (never leaves the engine)

// The traceback proper, over whatever serves the codes: W = Walker (the codes the fill stored), BandWalker (recomputed in a band around the
// end cell's diagonal) or ReWalker (recomputed on demand, window by window).
template <class W>
VGK_HD int32_t walk_body(const GsswParams& P, uint32_t i, const ProbDesc& d, W& w, unsigned long long best_key) {
    constexpr uint32_t W_SPEC = W::SPEC;
    vgk_result res;
    res.score = 0; res.status = VGK_OK; res.end_node = -1; res.end_offset = -1; res.end_read = -1;
    res.first_offset = 0; res.n_ops = 0; res.ops_begin = d.ops_off;
    const NodeRec* nodes = P.nodes + d.node_off;
    const bool pinned = (d.flags & 15u) == VGK_GSSW_PINNED;
    const bool xdrop = (d.flags & 15u) == VGK_XDROP_PINNED;   // rows = consumed read bases 0..len, scores carry XOFF
    const int32_t go = (int32_t)P.go, ge = (int32_t)P.ge;
    const int32_t S = (int32_t)P.scale;                        // walker arithmetic runs in the kernels' scaled units
    const int32_t zero = xdrop ? (int32_t)P.xoff : 0;          // representation of score 0

    int32_t cur = 0; uint32_t c = 0, node = 0; int32_t r = 0;
    const bool have = walk_end_cell(P, d, best_key, [&](const NodeRec& n, uint32_t row) { return w.saved(n, row); }, cur, c, node, r);
    if (pinned && !have) { res.status = VGK_EINVAL; P.results[i] = res; return VGK_EINVAL; }
    if (cur >= 2047 * S) { res.status = VGK_EOVERFLOW; P.results[i] = res; return VGK_EOVERFLOW; }
    if (!have || cur <= zero) { P.results[i] = res; return VGK_OK; }  // score 0: the caller synthesises soft clips / full insertion
    res.score = (cur - zero) / S; res.end_node = (int32_t)node; res.end_offset = (int32_t)(c - nodes[node].col_start);
    res.end_read = xdrop ? r - 1 : r;
    if (!(d.flags & VGK_GSSW_TRACEBACK)) { P.results[i] = res; return VGK_OK; }

    // CIGAR elements are produced back to front into the tail of this read's window.  The run being
    // built lives in registers (rn, ro, rl) and is written once, when a different (node, op) starts.
    vgk_op* ops = P.ops + d.ops_off;
    uint32_t pos = d.ops_cap;   // first used slot
    int32_t status = VGK_OK;
    uint32_t rn = 0, ro = 0xffu, rl = 0;
#define VGK_FLUSH() do { if (rl) { if (pos == 0) status = VGK_EOPS; else { --pos; \
        ops[pos].node = rn; ops[pos].len = (uint16_t)rl; ops[pos].op = (uint8_t)ro; ops[pos].pad = 0; } } } while (0)
#define VGK_PUSH(NODE, OP, LEN) do { \
        if (rl && rn == (uint32_t)(NODE) && ro == (uint32_t)(OP)) rl += (LEN); \
        else { VGK_FLUSH(); rn = (uint32_t)(NODE); ro = (uint32_t)(OP); rl = (LEN); } } while (0)

    if (r < (int32_t)d.L - 1) VGK_PUSH(node, VGK_OP_S, d.L - 1 - (uint32_t)r);
    enum { ST_H, ST_E, ST_F } st = ST_H;
    bool at_root = false;
    uint32_t first_c = c;
    uint32_t node_start = nodes[node].col_start;
    // every iteration consumes a read base, a graph base or changes state once: bounded by 2(L+R)+4
    for (uint32_t guard = 0; guard < 2 * (d.L + d.R) + 4 && status == VGK_OK; ++guard) {
        const bool first = (c == node_start);
        (void)first;
        if (st == ST_H) {
            if (!xdrop && cur == 0) break;
            // Alignments are mostly diagonal runs: fetch the codes and scores of the next (up to) four
            // diagonal cells together so their memory latencies overlap, then consume them in order.
            const uint32_t room = c - node_start + 1, rows = (uint32_t)r + 1;
            uint32_t nspec = room < rows ? room : rows; nspec = nspec < W_SPEC ? nspec : W_SPEC;
            uint32_t fl[W_SPEC]; int32_t sc[W_SPEC];
#pragma unroll
            for (uint32_t k = 0; k < W_SPEC; ++k) {
                fl[k] = 1u; sc[k] = 0;
                if (k < nspec) { fl[k] = w.code((uint32_t)r - k, c - k); sc[k] = w.score((uint32_t)r - k, c - k); }
            }
#if defined(VGK_DEBUG_WALK) && !defined(__HIP_DEVICE_COMPILE__)
            printf("H r=%d c=%u cur=%d nspec=%u fl=%u %u sc=%d %d\n", r, c, cur, nspec, fl[0], fl[1], sc[0], sc[1]);
#endif
            if (fl[0] & W_MISS) { status = W_MISSED; break; }
            if (fl[0] & 1u) { st = (fl[0] & 2u) ? ST_F : ST_E; continue; }
            bool stop = false;
#pragma unroll
            for (uint32_t k = 0; k < W_SPEC; ++k) {
                if (stop || k >= nspec || (fl[k] & (1u | W_MISS)) || (!xdrop && cur == 0)) { stop = true; continue; }
                VGK_PUSH(node, VGK_OP_M, 1); first_c = c;
                cur -= sc[k]; r -= 1;
                if (r < 0 || (!xdrop && cur == 0)) { stop = true; continue; }
                if (c != node_start) c -= 1;
                else {
                    const NodeRec& nr = nodes[node];
                    if (xdrop && nr.n_pred == 0) {       // back at the dozeu root: r read bases are a leading insertion
                        if (r > 0) VGK_PUSH(node, VGK_OP_I, (uint32_t)r);
                        at_root = true; stop = true; continue;
                    }
                    int32_t found = -1;
                    if (nr.n_pred == 1) found = (int32_t)P.preds[nr.pred_begin];
                    else for (uint32_t kk = 0; kk < nr.n_pred; ++kk) {
                        const uint32_t p = P.preds[nr.pred_begin + kk];
                        if ((int32_t)(w.saved(nodes[p], (uint32_t)r) & 0xffffu) == cur) { found = (int32_t)p; break; } }
                    if (found < 0) { status = VGK_EINVAL; stop = true; continue; }
                    node = (uint32_t)found; c = nodes[node].col_end - 1; node_start = nodes[node].col_start;
                    stop = true;      // speculation never crosses a node boundary
                }
            }
            if (at_root || r < 0 || (!xdrop && cur == 0)) break;
        } else if (st == ST_E) {
            VGK_PUSH(node, VGK_OP_D, 1); first_c = c;
            uint32_t pnode = node, pc = c - 1;
            if (first) {
                const NodeRec& nr = nodes[node];
                if (xdrop && nr.n_pred == 0) {           // deletion opened straight from the root column
                    if (r > 0) VGK_PUSH(node, VGK_OP_I, (uint32_t)r);
                    at_root = true; break;
                }
                int32_t found = -1;
                if (nr.n_pred == 1) found = (int32_t)P.preds[nr.pred_begin];
                else for (uint32_t k = 0; k < nr.n_pred; ++k) {
                    const uint32_t p = P.preds[nr.pred_begin + k];
                    if (w.saved_e(nodes[p], (uint32_t)r) == cur) { found = (int32_t)p; break; } }
                if (found < 0) { status = VGK_EINVAL; break; }
                pnode = (uint32_t)found; pc = nodes[pnode].col_end - 1; node_start = nodes[pnode].col_start;
            }
            const uint32_t fe = w.code((uint32_t)r, pc);
            if (fe & W_MISS) { status = W_MISSED; break; }
            if (!(fe & 4u)) { st = ST_H; cur += go; } else cur += ge;
            c = pc; node = pnode;
        } else {
            VGK_PUSH(node, VGK_OP_I, 1);
            if (r == 0) { status = VGK_EINVAL; break; }
            const uint32_t ff = w.code((uint32_t)r - 1, c);
            if (ff & W_MISS) { status = W_MISSED; break; }
            if (!(ff & 8u)) { st = ST_H; cur += go; } else cur += ge;
            r -= 1;
        }
    }
    if (xdrop && status == VGK_OK && !at_root) status = VGK_EINVAL;     // a dozeu path always ends at the root
    if (!xdrop && status == VGK_OK && r >= 0) VGK_PUSH(node, VGK_OP_S, (uint32_t)r + 1);
    if (status == VGK_OK) VGK_FLUSH();
#undef VGK_PUSH
#undef VGK_FLUSH
    res.status = status;
    if (status == VGK_OK) {
        res.n_ops = d.ops_cap - pos; res.ops_begin = d.ops_off + pos;
        res.first_offset = (int32_t)(first_c - node_start);
    }
    P.results[i] = res;
    return status;
}

VGK_HD void walk_one(const GsswParams& P, uint32_t i, unsigned long long best_key) {
    const ProbDesc d = P.probs[i];      // by value: keeps the descriptor in registers across the walk's global stores
    Walker w{P, d, (d.geom >> 16) & 1u, d.lane0, d.geom & 0xffu, P.waves[d.wave].tb_off, 0u, 0xffffffffu, 0u, 0xffffffffu};
    walk_body(P, i, d, w, best_key);
}

// ---- the first of two passes over a batch of local alignments (GsswParams::walk_passes == 2; round 4) -----------------------------------
// The cell-by-cell walk is bound by the rate at which HBM answers scattered requests (§25: ~830 per loop iteration of a wavefront, ~250 per
// read); most alignments of short reads, though, are ONE diagonal run, and such a traceback need not be walked to be known.  H(r, c) >=
// H(r - 1, c - 1) + s(r, c) at every cell, and H >= 0: if subtracting the scores along the diagonal from H at the end cell arrives EXACTLY at 0
// without undercutting it on the way, every one of those inequalities is an equality, the diagonal is an optimal source at every cell of
// the run, and the traceback — which prefers the diagonal among equals (lane_row's tag order) and stops where H is 0 — is that run.  The
// scores need only the read's and the columns' bytes: the 160 of each before the end cell are fetched as ten 16-byte words apiece, all asked
// for at once, and the run is checked out of registers; nothing else is read but the descriptor, the end cell's key and two node records.
// What this pass does not settle — a gap, the window's edge,
// quality-adjusted profiles, pinned and X-drop problems, reads of more than 160 bases — it leaves alone: W_MISSED, and walk_one does the read.
constexpr uint32_t WD_STEPS = 160, WD_DWORDS = WD_STEPS / 4;
struct alignas(16) WdQuad { uint32_t x, y, z, w; };
// blk: 2 x WD_DWORDS dwords of this lane's own, dword j at blk[j * stride] (the kernel: LDS, the lanes' dwords interleaved — held in registers the
// 160 steps had to be unrolled to index them: 43 000 instructions, more than the instruction cache keeps)
// tab: the scores of the run's cells, tab[8 q + (column byte & 7)] for read base q — 0 for the column bytes that are no base (wd_table fills
// it: LDS in the kernel, so that a cell's score is one byte read instead of a choice among the profile words)
constexpr uint32_t WD_TAB = 40;
VGK_HD int16_t wd_table_entry(const GsswParams& P, uint32_t t) {
    const uint32_t q = t >> 3, b = t & 7u;
    return b < 4u ? (int16_t)((int32_t)((P.prof4[q] >> (8u * b)) & 0xffu) - (int32_t)P.bias) : (int16_t)0;
}
VGK_HD int32_t walk_diag_one(const GsswParams& P, uint32_t i, unsigned long long best_key, uint32_t* blk, uint32_t stride, const int16_t* tab) {
    const ProbDesc d = P.probs[i];
    const uint32_t mode = d.flags & 15u;
    if (mode != VGK_GSSW_LOCAL || d.prof_off != 0xffffffffu || d.L > WD_STEPS) return W_MISSED;
    const int32_t S = (int32_t)P.scale;
    const NodeRec* nodes = P.nodes + d.node_off;
    int32_t cur = 0; uint32_t c = 0, node = 0; int32_t r = 0;
    const bool have = walk_end_cell(P, d, best_key, [&](const NodeRec&, uint32_t) { return 0u; }, cur, c, node, r);
    if (have && cur >= 2047 * S) return W_MISSED;                       // (the overflow answer is walk_body's)
    vgk_result res;
    res.score = 0; res.status = VGK_OK; res.end_node = -1; res.end_offset = -1; res.end_read = -1; res.first_offset = 0; res.n_ops = 0; res.ops_begin = d.ops_off;
    if (!have || cur <= 0) { P.results[i] = res; return VGK_OK; }
    // the 160 bytes that end with the end cell's row / column (step k of the run is byte 159 - k), asked for before anything that has to be
    // waited for; a block that would start before its arena is not asked for — the first read of a batch
    const uint32_t r_at = d.read_off + (uint32_t)r, c_at = d.col_off + c;
    if (r_at + 1u < WD_STEPS || c_at + 1u < WD_STEPS) return W_MISSED;
    { const uint8_t* rsrc = P.reads + (r_at + 1u - WD_STEPS); const uint8_t* csrc = P.colinfo + (c_at + 1u - WD_STEPS);
      WdQuad rq[WD_DWORDS / 4], cq[WD_DWORDS / 4];
      for (uint32_t j = 0; j < WD_DWORDS / 4; ++j) { __builtin_memcpy(&rq[j], rsrc + 16u * j, 16); __builtin_memcpy(&cq[j], csrc + 16u * j, 16); }
      for (uint32_t j = 0; j < WD_DWORDS / 4; ++j) {
          blk[(4u * j + 0u) * stride] = rq[j].x; blk[(4u * j + 1u) * stride] = rq[j].y; blk[(4u * j + 2u) * stride] = rq[j].z; blk[(4u * j + 3u) * stride] = rq[j].w;
          blk[(WD_DWORDS + 4u * j + 0u) * stride] = cq[j].x; blk[(WD_DWORDS + 4u * j + 1u) * stride] = cq[j].y; blk[(WD_DWORDS + 4u * j + 2u) * stride] = cq[j].z; blk[(WD_DWORDS + 4u * j + 3u) * stride] = cq[j].w;
      } }
    const uint32_t end_start = nodes[node].col_start;
    res.score = cur / S; res.end_node = (int32_t)node; res.end_offset = (int32_t)(c - end_start); res.end_read = r;
    if (!(d.flags & VGK_GSSW_TRACEBACK)) { P.results[i] = res; return VGK_OK; }
    vgk_op* ops = P.ops + d.ops_off;
    uint32_t pos = d.ops_cap;
    bool room_ok = true;
    auto put = [&](uint32_t nd, uint32_t op, uint32_t len) { if (pos == 0) { room_ok = false; return; } --pos; ops[pos].node = nd; ops[pos].len = (uint16_t)len; ops[pos].op = (uint8_t)op; ops[pos].pad = 0; };
    if (r < (int32_t)d.L - 1) put(node, VGK_OP_S, d.L - 1 - (uint32_t)r);
    // the run: v = what H must be at the next cell; run_node = the node being crossed, run_from = the cells taken when it was entered.
    // A full-length bonus on the read's last base can only belong to the run's first cell (the rows go up from r), one on its first base
    // only to the cell of row 0 — step r.
    int32_t v = cur - (r + 1 == (int32_t)d.L ? (int32_t)d.bonus_end : 0);
    uint32_t run_node = node, run_from = 0, taken = 0;
    int verdict = 0;                                                  // 1: arrived at 0; -1: not settled here
    uint32_t n_it = (uint32_t)r + 1u < c + 1u ? (uint32_t)r + 1u : c + 1u;      // cells on the diagonal inside the window ...
    n_it = n_it < WD_STEPS ? n_it : WD_STEPS;                                      // ... and inside the blocks
    uint32_t rw = 0, cw = 0;
    bool fresh = true;                                                // the cached words are to be read (again)
    uint32_t col_base = c;                                            // step k's cell lies in window column col_base - k (re-based where the run jumps to another node's columns)
    for (uint32_t k = 0; k < n_it; ++k) {
        const uint32_t at = WD_STEPS - 1u - k;
        if ((at & 3u) == 3u || fresh) { rw = blk[(at >> 2) * stride]; cw = blk[(WD_DWORDS + (at >> 2)) * stride]; fresh = false; }
        const uint32_t ci = (cw >> (8u * (at & 3u))) & 0xffu, q = (rw >> (8u * (at & 3u))) & 0xffu;
        v -= (int32_t)tab[8u * q + (ci & CI_BASE_MASK)] + (k == (uint32_t)r ? (int32_t)d.bonus_start : 0);
        ++taken;
        if (v <= 0) { verdict = v == 0 ? 1 : -1; break; }
        if (ci & CI_NODE_START) {
            if (!(ci & CI_SEED_SLOW) && run_node != 0u) {              // on into the node before this one: its only predecessor, the columns go on
                put(run_node, VGK_OP_M, taken - run_from); run_node -= 1u; run_from = taken;
                continue;
            }
            // A node with other predecessors than the node before it.  The traceback takes the first of them, in their order, whose H in the
            // row above equals what the run needs (walk_body) — H of a node's last column is what the fill keeps for its successors
            // (P.scratch) — and if the run then arrives at 0 that H was the cell's (the proof above, unchanged: the chosen cell IS a
            // diagonal predecessor).  The predecessor's columns lie somewhere else in the stream: the column block is fetched again,
            // ending where the run goes on.
            const uint32_t row = (uint32_t)r - k;
            if (row == 0u || k + 1u >= WD_STEPS) { verdict = -1; break; }
            const NodeRec nr = nodes[run_node];
            int32_t found = -1;
            if (nr.n_pred == 1u) found = (int32_t)P.preds[nr.pred_begin];
            else for (uint32_t kk = 0; kk < nr.n_pred; ++kk) {
                const uint32_t p = P.preds[nr.pred_begin + kk];
                const int32_t slot = nodes[p].slot;
                if (slot >= 0 && (int32_t)(P.scratch[d.scratch_off + (uint32_t)slot * d.Lpad + (row - 1u)] & 0xffffu) == v) { found = (int32_t)p; break; }
            }
            if (found < 0) { verdict = -1; break; }
            const uint32_t pc = nodes[found].col_end - 1u, c_abs = d.col_off + pc, next_at = WD_STEPS - 2u - k;      // the next cell's column, and its byte in the block
            if (c_abs < next_at) { verdict = -1; break; }                // (a block that would start before the arena)
            put(run_node, VGK_OP_M, taken - run_from); run_node = (uint32_t)found; run_from = taken;
            const uint8_t* csrc = P.colinfo + (c_abs - next_at);
            for (uint32_t j = 0; 4u * j <= next_at; ++j) {               // dwords 0 .. next_at / 4 (the bytes behind next_at are steps already taken; never past c_abs + 3: the arena's padding)
                uint32_t w; __builtin_memcpy(&w, csrc + 4u * j, 4);
                blk[(WD_DWORDS + j) * stride] = w;
            }
            fresh = true; col_base = pc + k + 1u;
            const uint32_t more = row < pc + 1u ? row : pc + 1u;         // cells left on the diagonal: rows above, columns of the window from pc down
            n_it = k + 1u + more < WD_STEPS ? k + 1u + more : WD_STEPS;
        }
    }
    const uint32_t run_len = taken - run_from;                        // (the window's edge or the blocks' before H reached 0: walk_body's business)
    if (verdict != 1 || !room_ok) return W_MISSED;
    put(run_node, VGK_OP_M, run_len);
    const int32_t r_left = r - (int32_t)taken;                        // rows above the alignment: a soft clip
    if (r_left >= 0) put(run_node, VGK_OP_S, (uint32_t)r_left + 1u);
    if (!room_ok) return W_MISSED;
    const uint32_t first_c = col_base - (taken - 1u);
    res.n_ops = d.ops_cap - pos; res.ops_begin = d.ops_off + pos;
    res.first_offset = (int32_t)(first_c - nodes[run_node].col_start);
    P.results[i] = res;
    return VGK_OK;
}
VGK_HD uint32_t* tb_miss_count(const GsswParams& P);
VGK_HD void tb_miss_add(const GsswParams& P, uint32_t i);
VGK_HD void walk_first_one(const GsswParams& P, uint32_t i, unsigned long long best_key, uint32_t* blk, uint32_t stride, const int16_t* tab) {
    if (walk_diag_one(P, i, best_key, blk, stride, tab) == W_MISSED) tb_miss_add(P, i);
}

// ---------------------------------------------------------------------------
// The traceback that does not tax the fill (TB_REWALK).
//
// Building, merging and storing a 4-bit code per cell is 8-9 of the ~20 VALU instructions the fill spends per row pair, for codes of
// which the traceback reads one cell in four hundred (a 150-base path through 62 000 cells).  In this mode the fill runs the
// recurrence alone (lane_row<..., TB = false>) and leaves behind only what is needed to run any piece of it again:
//   * per (step, lane) the lane's outputs to the lane below — H and F of its last row — 2 dwords ("boundary rows"), and
//   * per lane and every TB_CKPT-th column its K rows of H and E after that column, 2 K dwords ("checkpoints");
// about 0.64 of the bytes of the codes.  The traceback (one lane per read) asks for codes cell by cell as before; they come out of
// a window of one lane block (K rows) x up to TB_CKPT columns that is recomputed on demand by the FILL'S OWN lane code
// (lane_column<..., TB = true, RE = true>) from the checkpoint before it and the boundary rows above it, and kept in LDS.  The codes
// are therefore the ones the code-storing fill would have written — same instructions, same inputs: H / E of a checkpoint are the
// tagged build's values without their tags, which that code strips or overwrites before use — and the path is identical.  A
// 150-base path crosses ~8 lane blocks and ~2 windows in each: ~5 000 of the 62 000 cells are computed a second time.
// ---------------------------------------------------------------------------
enum : int32_t { TB_CODES = 0, TB_REWALK = 1 };
constexpr uint32_t TB_CKPT = 16;         // columns between two checkpoints = the widest window (even: the window keeps two columns per dword)

// The band: a traceback runs close to the diagonal through its end cell, so the codes it will ask for can be computed BEFORE it starts —
// for every lane block of a read at once, which the on-demand form (a window when the walk gets there) cannot: lane block g needs the
// columns the diagonal crosses in its rows, TB_SLACK more on either side, from the checkpoint before them.  All blocks of all reads of a
// wavefront run that many columns in lock step (no lane waits for another, no input depends on a neighbour: the boundary rows are in HBM),
// both reads of a pair together as in the fill.  A path that leaves its band (a long gap, a drift of more than TB_SLACK columns) is
// noticed by the walk and handed to the on-demand form.
constexpr uint32_t TB_SLACK = 8;
VGK_HD uint32_t tb_band_cols(uint32_t K) { return K + 2u * TB_SLACK + TB_CKPT - 1u; }      // columns a lane block's band can need, counted from its checkpoint
struct TbBand { uint32_t cs, hi; bool used; };      // columns [cs, hi] of a lane block are (to be) recomputed; cs is a multiple of TB_CKPT
VGK_HD TbBand tb_band_of(uint32_t r_e, uint32_t c_e, uint32_t g, uint32_t K) {
    TbBand b; b.used = g * K <= r_e; b.cs = 0; b.hi = 0;
    if (!b.used) return b;
    // the diagonal through the end cell meets row r at column c_e - (r_e - r) (left of column 0: the path has ended or left the diagonal)
    const uint32_t r_hi = g * K + K - 1u < r_e ? g * K + K - 1u : r_e;
    const uint32_t back_lo = r_e - g * K + TB_SLACK, back_hi = r_e - r_hi;
    const uint32_t lo = c_e > back_lo ? c_e - back_lo : 0u;
    uint32_t hi = c_e > back_hi ? c_e - back_hi + TB_SLACK : TB_SLACK;
    if (hi > c_e) hi = c_e;                                             // a traceback never moves right of where it started
    b.cs = lo / TB_CKPT * TB_CKPT;
    const uint32_t last = b.cs + tb_band_cols(K) - 1u;
    b.hi = hi < last ? hi : last;
    return b;
}

constexpr uint32_t TB_BND_CHUNK = 16;    // steps of boundary rows the fill writes at a time (16 x 8 bytes = one line per lane)
VGK_HD uint32_t tb_steps_padded(uint32_t n_steps) { return (n_steps + TB_BND_CHUNK - 1u) / TB_BND_CHUNK * TB_BND_CHUNK; }
VGK_HD uint64_t tb_bnd_dwords(uint32_t n_steps) { return 128ull * tb_steps_padded(n_steps); }

// dwords of a wavefront's traceback arena: enough for either form
VGK_HD uint64_t tb_wave_dwords(uint32_t n_steps, uint32_t K) {
    const uint64_t rec = (K + 3u) >> 2;
    const uint64_t codes = (uint64_t)((n_steps + TB_TILE - 1) / TB_TILE * TB_TILE) * 64u * rec;
    const uint64_t rewalk = tb_bnd_dwords(n_steps) + (uint64_t)((n_steps + TB_CKPT - 1) / TB_CKPT) * 128u * K + 64ull * tb_band_cols(K) * rec;
    return codes > rewalk ? codes : rewalk;
}
// {out_h, out_f} of (step, lane): LANE-major — a lane's steps lie behind each other, because whoever computes a window or a band again
// reads ONE lane's run of consecutive steps (step-major, 64 x 66 scattered 8-byte reads per wavefront made the band kernel 3.5 x slower than
// its arithmetic: profiles/r04).  The fill gathers TB_BND_CHUNK steps in LDS and writes a lane's chunk as one 128-byte line.
VGK_HD uint64_t tb_bnd(uint64_t tb_off, uint32_t n_steps, uint32_t t, uint32_t lane) { return tb_off + ((uint64_t)lane * tb_steps_padded(n_steps) + t) * 2u; }
VGK_HD uint64_t tb_ckpt(uint64_t tb_off, uint32_t n_steps, uint32_t cb, uint32_t lane, uint32_t K) {                         // H[K] then E[K] after column cb * TB_CKPT + TB_CKPT - 1
    return tb_off + tb_bnd_dwords(n_steps) + ((uint64_t)cb * 64u + lane) * 2u * K;
}
VGK_HD uint64_t tb_band(uint64_t tb_off, uint32_t n_steps, uint32_t x, uint32_t lane, uint32_t K) {                          // the record (ceil(K/4) dwords, both reads of the pair) of band column x of a lane
    return tb_off + tb_bnd_dwords(n_steps) + (uint64_t)((n_steps + TB_CKPT - 1) / TB_CKPT) * 128u * K + ((uint64_t)x * 64u + lane) * ((K + 3u) >> 2);
}

// the fill's stores in this mode, after lane_step<K, S8, false> of step t
template <int K>
VGK_HD void lane_store_boundary(const Lane<K>& s, const GsswParams& P, const WaveDesc& wd, uint32_t t, uint32_t lane) {      // (the emulator: straight to its place)
    uint32_t* b = P.tb + tb_bnd(wd.tb_off, wd.n_steps, t, lane);
    b[0] = s.out_h; b[1] = s.out_f;
}
template <int K>
VGK_HD void lane_store_checkpoint(const Lane<K>& s, const GsswParams& P, const WaveDesc& wd, uint32_t t, uint32_t lane) {
    const uint32_t c = t - s.g;                                   // the column this lane has just finished
    if (t >= s.g && (c & (TB_CKPT - 1u)) == TB_CKPT - 1u && !((s.info & CI_INVALID) && (s.info & (CI_INVALID << 16)))) {
        uint32_t* k = P.tb + tb_ckpt(wd.tb_off, wd.n_steps, c / TB_CKPT, lane, (uint32_t)K);
#pragma unroll
        for (int m = 0; m < K; ++m) { k[m] = s.H[m]; k[K + m] = s.E[m]; }
    }
}

// Serves Walker::code() from a recomputed window.  `win` = this lane's slice of the wavefront's LDS buffer (stride 64 dwords):
// dword (colpair * REC + j) holds the 16 code bits of row quad j for columns 2 colpair (low half) and 2 colpair + 1 (high half).
template <int K, bool S8>
struct ReWalker {
    static constexpr uint32_t SPEC = 1;          // the codes are a few cycles away: nothing to fetch ahead
    static constexpr uint32_t REC = (K + 3) / 4;
    Walker base;
    uint32_t* win; uint32_t win_stride;
    uint32_t n_steps;
    uint32_t win_g = 0xffffffffu, win_c0 = 0, win_hi = 0;      // the window: lane block, first column (a multiple of TB_CKPT), last column computed
    Lane<K> ln;

    VGK_HD int32_t score(uint32_t r, uint32_t c) const { return base.score(r, c); }
    VGK_HD uint32_t saved(const NodeRec& n, uint32_t r) const { return base.saved(n, r); }
    VGK_HD int32_t saved_e(const NodeRec& n, uint32_t r) const { return base.saved_e(n, r); }

    VGK_HD uint32_t code(uint32_t r, uint32_t c) {
        const uint32_t g = r / K, m = r - g * K, j = m >> 2, i = m & 3u;
        if (g != win_g || c < win_c0 || c > win_hi) refill(g, c);
        const uint32_t x = c - win_c0;
        const uint32_t w = (win[((x >> 1) * REC + j) * win_stride] >> (16u * (x & 1u))) & 0xffffu;
        const uint32_t last = (4 * j + 3 < (uint32_t)K ? 4 * j + 3 : (uint32_t)K - 1) - 4 * j;
        const uint32_t raw = (w >> (4 * (last - i))) & 15u;
        if (!S8) return raw;
        const uint32_t src = (raw >> 1) & 3u;
        return (src != 2u ? 1u : 0u) | (src == 0u ? 2u : 0u) | ((raw & 1u) ? 0u : 4u) | ((raw & 8u) ? 0u : 8u);
    }

    // lane block g, columns [c / TB_CKPT * TB_CKPT, c], computed again from the checkpoint before them and the boundary rows above
    VGK_HD void refill(uint32_t g, uint32_t c) {
        const GsswParams& P = base.P; const ProbDesc& d = base.d;
        const uint32_t half = base.half, lane = base.lane0 + g;
        const uint64_t tb_off = base.tb_off;
#if VGK_PB_LDS && defined(__HIPCC__)
        ln.PBL = (typename Lane<K>::lds_u32*)(uintptr_t)0;      // (unsupported with VGK_PB_LDS: the rewalk kernels are not built for it)
#endif
        const uint32_t c0 = c / TB_CKPT * TB_CKPT;
        Lane<K>& s = ln;
        // the read in its half of the pair, the other half idle
        s.g = g; s.Lpad = d.Lpad;
        s.probA = half ? 0xffffffffu : 0u; s.probB = half ? 0u : 0xffffffffu;      // (set to the read's index below)
        s.LA = s.LB = 0; s.flagsA = s.flagsB = 0; s.RA = s.RB = 0; s.colA = s.colB = 0; s.ciA = s.ciB = s.ciA_n = s.ciB_n = 0;
        s.bsA = s.beA = s.bsB = s.beB = 0;
        if (half) { s.probB = prob_index; s.LB = d.L; s.flagsB = d.flags; s.RB = d.R; s.colB = d.col_off; s.bsB = d.bonus_start; s.beB = d.bonus_end; }
        else      { s.probA = prob_index; s.LA = d.L; s.flagsA = d.flags; s.RA = d.R; s.colA = d.col_off; s.bsA = d.bonus_start; s.beA = d.bonus_end; }
        const uint32_t keep = half ? 0xffff0000u : 0x0000ffffu;
        const uint32_t* ck = c0 ? P.tb + tb_ckpt(tb_off, n_steps, c0 / TB_CKPT - 1u, lane, (uint32_t)K) : nullptr;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const uint32_t row = g * K + m;
            uint32_t pw = 0;
            if (row < d.L) pw = d.prof_off != 0xffffffffu ? P.prof[d.prof_off + row] : P.prof4[P.reads[d.read_off + row]] + 0x01010101u * row_bonus(d.bonus_start, d.bonus_end, row, d.L);
            s.PA[m] = half ? 0u : pw; s.set_pb(m, half ? pw : 0u);
            s.H[m] = ck ? ck[m] & keep : 0u; s.E[m] = ck ? ck[K + m] & keep : 0u;
        }
        s.out_h = 0; s.out_f = 0; s.info = CI_INVALID2;
        s.best_lo = s.best_hi = 0; s.step_lo = s.step_hi = 0;
        s.one = 0x00010001u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(s.one));
#endif
        // nodes begun up to column c0 - 1; H of the row above at that column
        uint32_t node = 0xffffffffu;
        if (c0) {
            const NodeRec* nodes = P.nodes + d.node_off;
            uint32_t lo = 0, hi = d.n_nodes;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nodes[mid].col_start <= c0 - 1u) lo = mid; else hi = mid; }
            node = lo;
        }
        s.nodeA = s.nodeB = node;
        s.prev_rh = (c0 && g) ? P.tb[tb_bnd(tb_off, n_steps, c0 - 1u + g - 1u, lane - 1u)] & keep : 0u;
        uint32_t pend[REC];
        for (uint32_t col = c0; col <= c; ++col) {
            const uint32_t t = col + g;
            uint32_t rh = 0, rf = 0;
            if (g) { const uint32_t* b = P.tb + tb_bnd(tb_off, n_steps, t - 1u, lane - 1u); rh = b[0] & keep; rf = b[1] & keep; }
            const uint32_t ci = P.colinfo[d.col_off + col];
            const uint32_t rinfo = half ? ((uint32_t)CI_INVALID | (ci << 16)) : (ci | ((uint32_t)CI_INVALID << 16));
            uint32_t rec[REC];
            lane_column<K, S8, true, true>(s, P, t, rh, rf, rinfo, rec, rec + 4);
            const uint32_t x = col - c0;
#pragma unroll
            for (uint32_t j = 0; j < REC; ++j) {
                const uint32_t v = (rec[j] >> (16u * half)) & 0xffffu;
                if (x & 1u) win[((x >> 1) * REC + j) * win_stride] = pend[j] | (v << 16);
                else pend[j] = v;
            }
        }
        if (!((c - c0) & 1u)) {                                     // an odd number of columns: the last one alone in its dword
#pragma unroll
            for (uint32_t j = 0; j < REC; ++j) win[(((c - c0) >> 1) * REC + j) * win_stride] = pend[j];
        }
        win_g = g; win_c0 = c0; win_hi = c;
    }
    uint32_t prob_index = 0;
};

// ---- the band (see TB_SLACK above) ----------------------------------------------------------------------------------------------
// where a read's traceback will start, if it has one; scratch-resident H of a node's last column read like Walker::saved
VGK_HD bool tb_band_end(const GsswParams& P, const ProbDesc& d, unsigned long long best_key, uint32_t& r_e, uint32_t& c_e) {
    if (!(d.flags & VGK_GSSW_TRACEBACK)) return false;
    int32_t cur, r; uint32_t c, node;
    const bool have = walk_end_cell(P, d, best_key, [&](const NodeRec& n, uint32_t row) { return P.scratch[d.scratch_off + (uint32_t)n.slot * d.Lpad + row]; }, cur, c, node, r);
    const int32_t zero = (d.flags & 15u) == VGK_XDROP_PINNED ? (int32_t)P.xoff : 0;
    if (!have || cur <= zero || cur >= 2047 * (int32_t)P.scale || r < 0) return false;
    r_e = (uint32_t)r; c_e = c;
    return true;
}

// One lane of a fill wavefront again: lane block g of the pair's two reads, each over its own band, with the fill's code-building lane
// code.  Writes the records of band columns 0 .. tb_band_cols(K) - 1 (a read's own count may be smaller: its half is then idle).
template <int K, bool S8>
VGK_HD void band_fill_lane(const GsswParams& P, const WaveDesc& wd, uint32_t lane) {
    Lane<K> s;
#if VGK_PB_LDS && defined(__HIPCC__)
    s.PBL = (typename Lane<K>::lds_u32*)(uintptr_t)0;      // (unsupported with VGK_PB_LDS)
#endif
    lane_init(s, P, wd, lane);                   // the pair, its rows' profiles, the lane block index g — as the fill began
    const uint32_t g = s.g;
    uint32_t re[2] = {0, 0}, ce[2] = {0, 0}; TbBand bd[2]; bd[0].used = bd[1].used = false;
    const uint32_t prob[2] = {s.probA, s.probB};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (prob[h] == 0xffffffffu) continue;
        const ProbDesc& d = P.probs[prob[h]];
        if (tb_band_end(P, d, P.best[prob[h]], re[h], ce[h])) bd[h] = tb_band_of(re[h], ce[h], g, (uint32_t)K);
    }
    if (!bd[0].used && !bd[1].used) return;
    // the state before each half's first column: its checkpoint, the nodes begun so far, H of the row above
    uint32_t prev = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t keep = h ? 0xffff0000u : 0x0000ffffu;
        uint32_t node = 0xffffffffu;
        if (bd[h].used && bd[h].cs) {
            const ProbDesc& d = P.probs[prob[h]];
            const uint32_t* ck = P.tb + tb_ckpt(wd.tb_off, wd.n_steps, bd[h].cs / TB_CKPT - 1u, lane, (uint32_t)K);
#pragma unroll
            for (int m = 0; m < K; ++m) { s.H[m] |= ck[m] & keep; s.E[m] |= ck[K + m] & keep; }
            const NodeRec* nodes = P.nodes + d.node_off;
            uint32_t lo = 0, hi = d.n_nodes;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nodes[mid].col_start <= bd[h].cs - 1u) lo = mid; else hi = mid; }
            node = lo;
            if (g) prev |= P.tb[tb_bnd(wd.tb_off, wd.n_steps, bd[h].cs - 1u + g - 1u, lane - 1u)] & keep;
        }
        if (h) s.nodeB = node; else s.nodeA = node;
    }
    s.prev_rh = prev;
    // the columns: each half's column bytes and the rows above it are runs in memory (column stream; lane - 1's boundary rows), read one
    // column AHEAD of the arithmetic so that a load's latency lies under a column of DP
    uint32_t n_col[2]; const uint8_t* ci_at[2]; const uint32_t* up_at[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        n_col[h] = bd[h].used ? bd[h].hi - bd[h].cs + 1u : 0u;
        ci_at[h] = nullptr; up_at[h] = nullptr;
        if (n_col[h]) {
            ci_at[h] = P.colinfo + P.probs[prob[h]].col_off + bd[h].cs;
            if (g) up_at[h] = P.tb + tb_bnd(wd.tb_off, wd.n_steps, bd[h].cs + g - 1u, lane - 1u);
        }
    }
    uint32_t n_max = n_col[0] > n_col[1] ? n_col[0] : n_col[1];
    if ((P.dbg & 1) && n_max > 1u) n_max = 1u;
    struct In { uint32_t rinfo, rh, rf; };
    auto fetch = [&](uint32_t x) {
        In in; in.rinfo = CI_INVALID2; in.rh = 0; in.rf = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (x >= n_col[h]) continue;
            const uint32_t ci = ci_at[h][x];
            in.rinfo = h ? (in.rinfo & 0x0000ffffu) | (ci << 16) : (in.rinfo & 0xffff0000u) | ci;
            if (up_at[h]) {
                const uint32_t keep = h ? 0xffff0000u : 0x0000ffffu;
                in.rh |= up_at[h][2u * x] & keep; in.rf |= up_at[h][2u * x + 1u] & keep;
            }
        }
        return in;
    };
    uint32_t* out = P.tb + tb_band(wd.tb_off, wd.n_steps, 0, lane, (uint32_t)K);
    constexpr uint32_t OUT_STRIDE = 64u * ((K + 3) / 4);
    In next = fetch(0);
    for (uint32_t x = 0; x < n_max; ++x, out += OUT_STRIDE) {
        const In in = next;
        if (x + 1u < n_max) next = fetch(x + 1u);
        lane_column<K, S8, true, true>(s, P, x, in.rh, in.rf, in.rinfo, (P.dbg & 2) ? nullptr : out, out + 4);
    }
}

// Walker::code() out of the band records
struct BandWalker {
    static constexpr uint32_t SPEC = W_SPEC;
    Walker base;
    uint32_t n_steps, r_e, c_e;
    mutable uint32_t bg = 0xffffffffu; mutable TbBand bb{0, 0, false};      // the band of the lane block asked for last
    VGK_HD int32_t score(uint32_t r, uint32_t c) const { return base.score(r, c); }
    VGK_HD uint32_t saved(const NodeRec& n, uint32_t r) const { return base.saved(n, r); }
    VGK_HD int32_t saved_e(const NodeRec& n, uint32_t r) const { return base.saved_e(n, r); }
    VGK_HD uint32_t code(uint32_t r, uint32_t c) const {
        const uint32_t K = base.K, g = r / K, m = r - g * K, j = m >> 2, i = m & 3u;
        if (g != bg) { bg = g; bb = tb_band_of(r_e, c_e, g, K); }
        if (!bb.used || c < bb.cs || c > bb.hi) return W_MISS;
        const uint32_t w = base.P.tb[tb_band(base.tb_off, n_steps, c - bb.cs, base.lane0 + g, K) + j];
        const uint32_t last = (4 * j + 3 < K ? 4 * j + 3 : K - 1) - 4 * j;
        const uint32_t raw = (w >> (16 * base.half + 4 * (last - i))) & 15u;
        if (base.P.scale != 8) return raw;
        const uint32_t src = (raw >> 1) & 3u;
        return (src != 2u ? 1u : 0u) | (src == 0u ? 2u : 0u) | ((raw & 1u) ? 0u : 4u) | ((raw & 8u) ? 0u : 8u);
    }
};
// The reads whose walk left its band, for the on-demand form: a counter in best[n_problems] (zeroed with the keys before every run) and the
// read indices behind it (the packers allocate best[] with that room).
VGK_HD uint32_t* tb_miss_count(const GsswParams& P) { return reinterpret_cast<uint32_t*>(P.best + P.n_problems); }
VGK_HD uint32_t* tb_miss_list(const GsswParams& P) { return reinterpret_cast<uint32_t*>(P.best + P.n_problems + 1); }
VGK_HD uint64_t tb_best_entries(uint64_t n) { return n + 2u + (n + 1u) / 2u; }      // 64-bit words of best[]: the keys, the counter, the list
VGK_HD void tb_miss_add(const GsswParams& P, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = atomicAdd(tb_miss_count(P), 1u);
#else
    const uint32_t k = (*tb_miss_count(P))++;
#endif
    tb_miss_list(P)[k] = i;
}
// one wavefront of the second fill (GsswParams::spec_fill): the reads at [16 w2 .. ) of the miss list, two to a lane group as in the batch's own
// wavefronts (vgk_api.cpp / gssw_pack_device.hpp build those); the reads' descriptors learn where their codes will lie
VGK_HD void refill_layout_one(const GsswParams& P, uint32_t w2) {
    const uint32_t M = *tb_miss_count(P), G = P.refill_G, gpw = 64u / G;
    const uint32_t n_pairs2 = (M + 1u) / 2u, n_waves2 = (n_pairs2 + gpw - 1u) / gpw;
    if (w2 == 0) P.refill_count[0] = n_waves2;
    if (w2 >= n_waves2) return;
    const uint32_t* list = tb_miss_list(P);
    ProbDesc* probs = const_cast<ProbDesc*>(P.probs);                 // (the batch's own device array)
    uint32_t* order = const_cast<uint32_t*>(P.order);
    WaveDesc* waves = const_cast<WaveDesc*>(P.waves);
    WaveDesc wd; wd.first_pair = P.refill_pair0 + w2 * gpw; wd.G = G; wd.pair_end = P.refill_pair0 + n_pairs2;
    uint32_t rmax = 0;
    for (uint32_t q = 0; q < gpw; ++q) for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t k = 2u * (w2 * gpw + q) + h;
        const uint32_t i = k < M ? list[k] : 0xffffffffu;
        if (w2 * gpw + q < n_pairs2) order[2u * (size_t)P.refill_pair0 + k] = i;
        if (i == 0xffffffffu) continue;
        ProbDesc& d = probs[i];
        // a traceback never looks right of its end cell, and a LOCAL alignment's end cell is known (the first fill's best key): the second
        // fill of such a read stops behind that column.  Its own best key is then the maximum over fewer cells — the same cell, first among
        // equals as before — and atomicMax leaves best[] as it is.
        uint32_t cols = d.R;
        if ((d.flags & 15u) == (uint32_t)VGK_GSSW_LOCAL) {
            const uint32_t c_e = 0xFFFFFu - (uint32_t)((P.best[i] >> 20) & 0xFFFFFu);
            if (c_e + 1u < cols) cols = c_e + 1u;
        }
        rmax = cols > rmax ? cols : rmax;
        // (where the batch's own fill put the read: kept in the descriptor's spare word, for a later run of this resident batch WITHOUT the
        // speculation — vgk_gssw_run's feedback may decide so — which refill_restore_one hands it back to.  Only the first displacement saves.)
        if (!(d.pad & 0x80000000u)) d.pad = 0x80000000u | (d.wave & 0xffffffu) | ((d.lane0 & 63u) << 24) | (((d.geom >> 16) & 1u) << 30);
        d.wave = P.refill_wave0 + w2; d.lane0 = q * G; d.geom = P.refill_K | (G << 8) | (h << 16);
    }
    wd.n_steps = rmax ? rmax + G - 1u : 0u;
    wd.tb_off = (unsigned long long)w2 * P.refill_slot;
    waves[P.refill_wave0 + w2] = wd;
}
// a read an earlier speculative run displaced goes back to the wavefront, lane group and half the batch's own fill gives it
VGK_HD void refill_restore_one(const GsswParams& P, uint32_t i) {
    ProbDesc& d = const_cast<ProbDesc*>(P.probs)[i];
    const uint32_t s = d.pad;
    if (!(s & 0x80000000u)) return;
    d.wave = s & 0xffffffu; d.lane0 = (s >> 24) & 63u; d.geom = (d.geom & 0xffffu) | (((s >> 30) & 1u) << 16); d.pad = 0;
}
VGK_HD void bandwalk_one(const GsswParams& P, uint32_t i, unsigned long long best_key) {
    const ProbDesc d = P.probs[i];
    const WaveDesc wd = P.waves[d.wave];
    BandWalker w{Walker{P, d, (d.geom >> 16) & 1u, d.lane0, d.geom & 0xffu, wd.tb_off, 0u, 0xffffffffu, 0u, 0xffffffffu}, wd.n_steps, 0u, 0u};
    if (!tb_band_end(P, d, best_key, w.r_e, w.c_e)) { w.r_e = 0; w.c_e = 0; }      // (no traceback to do: walk_body finds out the same way and never asks for a code)
    if (walk_body(P, i, d, w, best_key) == W_MISSED) tb_miss_add(P, i);
}

template <int K, bool S8>
VGK_HD void rewalk_one(const GsswParams& P, uint32_t i, unsigned long long best_key, uint32_t* win, uint32_t win_stride) {
    const ProbDesc d = P.probs[i];
    if ((d.geom & 0xffu) != (uint32_t)K) return;        // another rows-per-lane class's launch takes it
    const WaveDesc wd = P.waves[d.wave];        // the band served this read's walk (nearly all of them)
    ReWalker<K, S8> w{Walker{P, d, (d.geom >> 16) & 1u, d.lane0, d.geom & 0xffu, wd.tb_off, 0u, 0xffffffffu, 0u, 0xffffffffu}, win, win_stride, wd.n_steps};
    w.prob_index = i;
    walk_body(P, i, d, w, best_key);
}

}  // namespace vgk
