// gssw_matrix_device.hpp — the pinned gssw fill with every cell's H / E / F kept, for the k-best tracebacks of
// vgk_gssw_align_multi (Aligner::align_pinned_multi, reference src/aligner.cpp:423-435).
//
// The packed fill kernel of gssw_device.hpp streams 4-bit traceback codes: enough for the one best walk, not for ranking
// alternates, which needs the score lost at every source not taken.  This path is mpmap's (max_alt_alns > 1) and low-volume, so
// the fill here is the plain recurrence, one thread per problem, int32 cells written to HBM column by column:
//   E[r][c] = max(0, H[r][c-1] - go, E[r][c-1] - ge)   F[r][c] = max(0, H[r-1][c] - go, F[r-1][c] - ge)
//   H[r][c] = max(H[r-1][c-1] + s(r, c), E[r][c], F[r][c]);   "c-1" at a node's first column = element-wise max over the
//   predecessors' last columns (gssw_create_seed_*); bonus at read row 0 only (pinned: the other end is the pinned one).
// Two fills produce the same matrices: gssw_matrix_one (one thread per problem, the plain loops — kept for scorings with
// gap_open < gap_extend, where the column scan below does not apply, and as the emulator's cross-check) and
// gssw_matrix_wave_lane (one WAVEFRONT per problem: lane l owns the R consecutive read rows l*R .., a column is one step of the
// whole wave, the previous column lives in registers, and the vertical gap F comes from a max-plus prefix scan over the rows —
// F[r] = max(0, max_{r' < r} (Ht[r'] + r' ge) - go - (r - 1) ge) with Ht = max(diagonal + s, E), exact for go >= ge — serial inside
// a lane and one DPP scan across lanes, as in the banded kernel).  Cells leave as coalesced runs of 64 R int32 per plane.
// The alternates are enumerated on the host over these matrices (gssw_multi_api.cpp).
#pragma once
#include <stdint.h>
#include "../../include/vgk.h"
#include "pk16.hpp"

namespace vgk {

struct MProb {
    uint32_t L, n_nodes, R;
    uint32_t read_off, graph_off, node_off;
    uint64_t mat_off;                 // in cells: H at mat_off, E at mat_off + R*L, F at mat_off + 2*R*L; cell (c, r) at c*L + r
    int32_t  start_bonus;             // gssw: bonus at read row 0; X-drop band: the bonus on consuming the last read base
    int32_t  status;                  // out: VGK_OK or VGK_EOVERFLOW
    int32_t  best, best_c, best_v, best_i;   // X-drop band: the end cell the fill found (score, column, node, row), for the walk kernel
    int32_t  gap_cells, xt;           // X-drop band only: rows of the root column that hold a leading insertion (max_gap_length rounded up to
                                      // dozeu's 8-cell vector), and the x-drop threshold (go - ge) + ge * max_gap_length
};
struct MNode { uint32_t col_start, col_end, pred_begin, n_pred; };

struct GsswMatrixParams {
    MProb* probs; uint32_t n;
    const uint8_t* reads;             // codes 0..4
    const uint8_t* quals;             // quality-adjusted contexts: raw phred per read base (same offsets as reads)
    const uint8_t* graph;             // codes 0..4
    const MNode* nodes; const uint32_t* preds;    // predecessor node indices (problem-local)
    const int8_t* mat;                // 25 scores, or 256 x 25 by base quality
    int32_t go, ge;
    int32_t* cells;
    int32_t* node_fmax;               // X-drop band only: per node (indexed like `nodes`), the best score on the way to the node's end
    unsigned long long* stats;        // X-drop band only: [0] cells inside the bands
    // X-drop band only, nullable: the wavefront that filled a problem also picks its end cell and walks its traceback (lane 0, over the
    // matrices it has just written) — results[i] as vgk_xdrop_band_align returns them, ops in a window of L + R + 3 elements from ops_off[i]
    vgk_result* xb_results; vgk_op* xb_ops; const uint64_t* xb_ops_off; const uint8_t* xb_want_tb;
    // X-drop band only: the launch order.  order[0 .. n16) = problems of at most 127 read bases (128 rows = 16 of dozeu's vectors): FOUR
    // of them share a wavefront, 16 lanes each (giraffe's tails are 1-121 bases: a whole wavefront per tail left 48-60 lanes idle);
    // order[n16 .. n16 + n64) = the longer ones, a wavefront each.  Inside a class by descending graph size, so that the four of a
    // wavefront run about equally long.  Null: every problem a wavefront of its own, in the order given.
    // (packed fill only: order[0 .. n8) = tails of at most 63 bases, EIGHT to a wavefront, 8 lanes each — then the 16-lane class, then the rest)
    const uint32_t* xb_order; uint32_t xb_n16, xb_n64, xb_n8;
    uint16_t* xb_front;               // per column (a problem's at its graph_off): first vector of the front | one past the last << 8 — only those vectors are stored
    // X-drop band only: 1 = the planes hold 16-bit cells (`cells` is then an int16_t array of the same element count).  The arithmetic is
    // the same; a stored cell is its value saturated to int16, and whatever comes back at or below XB16_DEAD is unreachable.  The caller
    // (xdrop_band_api.cpp) picks this when no reachable cell of any problem of the launch can leave (XB16_DEAD, 32767): the kernel's time
    // is half its stores (2.9 TB/s of H and E vectors), and tails' scores are a few hundred at most.
    int32_t xb_cell16;                // 0, 1, or 2 = 16-bit cells AND the packed fill (xdrop_band_pk_lane): scores + xb_sb are bytes
    int32_t xb_sb;
};

VGK_HD void gssw_matrix_one(const GsswMatrixParams& P, uint32_t i) {
    MProb& pb = P.probs[i];
    const uint32_t L = pb.L;
    const uint64_t plane = (uint64_t)pb.R * L;
    int32_t* H = P.cells + pb.mat_off; int32_t* E = H + plane; int32_t* F = E + plane;
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    int status = VGK_OK;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        const MNode nd = nodes[v];
        for (uint32_t c = nd.col_start; c < nd.col_end; ++c) {
            const bool first = c == nd.col_start;
            const uint32_t ref = gr[c];
            int32_t h_up = 0, f_up = 0;                            // H, F of the row above in this column
            int32_t d_prev = 0;                                     // H of the previous column at the row above (the diagonal)
            for (uint32_t r = 0; r < L; ++r) {
                int32_t e = 0, d_here = 0;                          // d_here: previous column's H at this row = the next row's diagonal
                if (!first) {
                    const int32_t ph = H[(uint64_t)(c - 1) * L + r], pe = E[(uint64_t)(c - 1) * L + r];
                    const int32_t a = ph - P.go, b = pe - P.ge; e = a > b ? a : b; if (e < 0) e = 0;
                    d_here = ph;
                } else {
                    for (uint32_t k = 0; k < nd.n_pred; ++k) {
                        const uint64_t pc = (uint64_t)(nodes[P.preds[nd.pred_begin + k]].col_end - 1) * L + r;
                        const int32_t ph = H[pc], pe = E[pc];
                        const int32_t a = ph - P.go, b = pe - P.ge; int32_t en = a > b ? a : b; if (en < 0) en = 0;
                        if (en > e) e = en;
                        if (ph > d_here) d_here = ph;
                    }
                }
                int32_t f = 0;
                if (r > 0) { const int32_t a = h_up - P.go, b = f_up - P.ge; f = a > b ? a : b; if (f < 0) f = 0; }
                const int32_t s = (ql ? P.mat[25 * ql[r] + 5 * ref + rd[r]] : P.mat[5 * ref + rd[r]]) + (r == 0 ? pb.start_bonus : 0);
                int32_t h = (r == 0 ? 0 : d_prev) + s;
                if (e > h) h = e;
                if (f > h) h = f;
                if (h >= 32767) status = VGK_EOVERFLOW;             // gssw's int16 limit
                const uint64_t at = (uint64_t)c * L + r;
                H[at] = h; E[at] = e; F[at] = f;
                h_up = h; f_up = f; d_prev = d_here;
            }
        }
    }
    pb.status = status;
}

// XL = the cross-lane primitives of banded_device.hpp (down, scan_excl, fence).  Rows beyond the read compute harmless values that
// are never stored (they only read rows above them).
constexpr int32_t MNEG = -(1 << 28);
struct alignas(32) MVec8 { int32_t v[8]; };       // one of dozeu's 8-cell vectors as it lies in a plane of the band matrices
// ... and as it lies there with 16-bit cells (GsswMatrixParams::xb_cell16): two rows to a word, the even row low
struct alignas(16) MVec8h { uint32_t w[4]; };
constexpr int32_t XB16_DEAD = -16384;            // a 16-bit cell at or below this: unreachable
VGK_HD uint32_t xb_pack2(int32_t lo, int32_t hi) {                 // two cells saturated to int16: one v_cvt_pk_i16_i32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi));
#else
    auto sat = [](int32_t x) { return (uint32_t)(uint16_t)(int16_t)(x < -32768 ? -32768 : x > 32767 ? 32767 : x); };
    return sat(lo) | (sat(hi) << 16);
#endif
}
VGK_HD int32_t xb_cell_of(int16_t x) { return x <= XB16_DEAD ? -(1 << 28) : (int32_t)x; }
// the planes of a problem behind one type: CT = int32_t (MVec8 per vector) or int16_t (MVec8h)
template <class CT> struct XbPlane;
template <> struct XbPlane<int32_t> {
    int32_t* p;
    VGK_HD int32_t get(uint64_t at) const { return p[at]; }
    VGK_HD int32_t raw(uint64_t at) const { return p[at]; }
    VGK_HD int32_t conv(int32_t x) const { return x; }
    VGK_HD void load8(uint64_t at, int32_t (&v)[8]) const { const MVec8 x = *reinterpret_cast<const MVec8*>(p + at); for (int k = 0; k < 8; ++k) v[k] = x.v[k]; }
    VGK_HD void store8(uint64_t at, const int32_t (&v)[8]) const { MVec8 x; for (int k = 0; k < 8; ++k) x.v[k] = v[k]; *reinterpret_cast<MVec8*>(p + at) = x; }
};
template <> struct XbPlane<int16_t> {
    int16_t* p;
    VGK_HD int32_t get(uint64_t at) const { return xb_cell_of(p[at]); }
    VGK_HD int16_t raw(uint64_t at) const { return p[at]; }
    VGK_HD int32_t conv(int16_t x) const { return xb_cell_of(x); }
    VGK_HD void load8(uint64_t at, int32_t (&v)[8]) const {
        const MVec8h x = *reinterpret_cast<const MVec8h*>(p + at);
        for (int k = 0; k < 4; ++k) { v[2 * k] = xb_cell_of((int16_t)(uint16_t)(x.w[k] & 0xffffu)); v[2 * k + 1] = xb_cell_of((int16_t)(uint16_t)(x.w[k] >> 16)); }
    }
    VGK_HD void store8(uint64_t at, const int32_t (&v)[8]) const { MVec8h x; for (int k = 0; k < 4; ++k) x.w[k] = xb_pack2(v[2 * k], v[2 * k + 1]); *reinterpret_cast<MVec8h*>(p + at) = x; }
};
// the cells as the packed fill (xb_cell16 == 2) keeps them: unsigned, biased by 32 768, anything below 16 768 unreachable — stored as they are
// (no conversion at the store: eight instructions a column); only the walk reads them through this
template <> struct XbPlane<uint16_t> {
    uint16_t* p;
    static VGK_HD int32_t of(uint16_t x) { return x < 16768u ? -(1 << 28) : (int32_t)x - 32768; }
    VGK_HD int32_t get(uint64_t at) const { return of(p[at]); }
    VGK_HD uint16_t raw(uint64_t at) const { return p[at]; }
    VGK_HD int32_t conv(uint16_t x) const { return of(x); }
};
VGK_HD void bump_stat(unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
template <int R, class XL>
VGK_HD void gssw_matrix_wave_lane(const GsswMatrixParams& P, uint32_t i, uint32_t lane, XL& xl) {
    MProb& pb_out = P.probs[i];
    const MProb pb = pb_out;                                   // (in registers: through the reference every field is loaded again after each store)
    const int32_t L = (int32_t)pb.L, go = P.go, ge = P.ge;
    const uint64_t plane = (uint64_t)pb.R * pb.L;
    int32_t* H = P.cells + pb.mat_off; int32_t* E = H + plane; int32_t* F = E + plane;
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    const int32_t r0 = (int32_t)lane * R;
    // the lane's rows: score of each row against the five reference codes (the query profile: bonus at read row 0 folded in)
    int32_t prof[R][5];
    for (int k = 0; k < R; ++k) {
        const int32_t r = r0 + k;
        for (int g = 0; g < 5; ++g)
            prof[k][g] = r < L ? (int32_t)(ql ? P.mat[25 * ql[r] + 5 * g + rd[r]] : P.mat[5 * g + rd[r]]) + (r == 0 ? pb.start_bonus : 0) : 0;
    }
    int32_t Hp[R], Ep[R];               // previous column: H, and E of that column (the next column's E derives from both)
    int32_t overflow = 0;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        const MNode nd = nodes[v];
        for (uint32_t c = nd.col_start; c < nd.col_end; ++c) {
            const bool first = c == nd.col_start;
            int32_t e[R], dg[R];                                   // this column's E, and the diagonal H (previous column, row above)
            if (!first) {
                int32_t up = xl.down(Hp[R - 1]);                   // previous column, last row of the lane above
                if (lane == 0) up = 0;
                for (int k = 0; k < R; ++k) {
                    const int32_t a = Hp[k] - go, b = Ep[k] - ge; int32_t x = a > b ? a : b; e[k] = x > 0 ? x : 0;
                    dg[k] = k ? Hp[k - 1] : up;
                }
            } else {
                // the node's first column: element-wise max over the predecessors' last columns (gssw_create_seed_*), read back from
                // the matrices (the wave wrote them; fenced at the end of every column of a node's last column below)
                for (int k = 0; k < R; ++k) { e[k] = 0; dg[k] = 0; }
                for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const uint64_t pc = (uint64_t)(nodes[P.preds[nd.pred_begin + q]].col_end - 1) * pb.L;
                    for (int k = 0; k < R; ++k) {
                        const int32_t r = r0 + k;
                        if (r < L) {
                            const int32_t ph = H[pc + r], pe = E[pc + r];
                            const int32_t a = ph - go, b = pe - ge; int32_t x = a > b ? a : b; x = x > 0 ? x : 0;
                            if (x > e[k]) e[k] = x;
                        }
                        if (r >= 1 && r - 1 < L) { const int32_t ph = H[pc + r - 1]; if (ph > dg[k]) dg[k] = ph; }
                    }
                }
            }
            const uint32_t ref = gr[c];
            int32_t ht[R], pre[R], run = MNEG;
            for (int k = 0; k < R; ++k) {
                const int32_t r = r0 + k;
                int32_t h = (r == 0 ? 0 : dg[k]) + prof[k][ref];
                if (e[k] > h) h = e[k];
                ht[k] = h;
                pre[k] = run;
                const int32_t gk = h + r * ge;
                run = gk > run ? gk : run;
            }
            const int32_t excl = xl.scan_excl(run);                // max over the rows of the lanes above (MNEG for lane 0)
            for (int k = 0; k < R; ++k) {
                const int32_t r = r0 + k;
                const int32_t pm = excl > pre[k] ? excl : pre[k];
                int32_t f = pm - go - (r - 1) * ge; f = (r > 0 && f > 0) ? f : 0;
                const int32_t h = ht[k] > f ? ht[k] : f;
                if (h >= 32767) overflow = 1;                       // gssw's int16 limit
                if (r < L) { const uint64_t at = (uint64_t)c * pb.L + r; H[at] = h; E[at] = e[k]; F[at] = f; }
                Hp[k] = h; Ep[k] = e[k];
            }
        }
        xl.fence();                                                // successors read this node's last column through memory
    }
    if (xl.any(overflow) && lane == 0) pb_out.status = VGK_EOVERFLOW;
    else if (lane == 0) pb_out.status = VGK_OK;
}

// ---- X-drop with dozeu's band (vgk_xdrop_band_align; the rules are stated in include/vgk.h and, identically, in oracle/vgo_xdrop.c) ----
// Same shape as the wavefront fill above with R = 8: a lane IS one of dozeu's 8-cell vectors.  Rows i = 0 .. L count consumed read
// bases (L + 1 rows, cell (c, i) at c * (L + 1) + i); no zero floor; source nodes start from the root column.  A column is computed
// whole — cells without a live input come out unreachable by themselves — then one ballot over "my vector holds a cell >= best - xt"
// gives the first and last live vector, lanes outside store (and keep) unreachable cells, and the best score of the front is updated
// by a wave maximum.  A node whose incoming fronts are all empty costs no arithmetic worth mentioning: its cells stay unreachable.
template <class CT, class XL>
VGK_HD void xdrop_band_wave_lane_t(const GsswMatrixParams& P, uint32_t pi, uint32_t lane, XL& xl) {
    constexpr int R = 8;
    MProb& pb_out = P.probs[pi];
    const MProb pb = pb_out;                                   // a copy in registers: read through the reference, every field would be loaded again after each store (and wait for it)
    const int32_t L = (int32_t)pb.L, rows = L + 1, go = P.go, ge = P.ge;
    const int32_t stride = (rows + 7) & ~7;                     // a column's cells in memory: whole 8-row vectors, so that a lane stores its vector as two 16-byte words per plane
    const uint64_t plane = (uint64_t)pb.R * (uint64_t)stride;
    // two planes: H and E.  F is not kept — the traceback prefers the diagonal, then the deletion, so a cell that neither explains is
    // an insertion, and the insertion state itself walks on H alone (a third plane was a third of the kernel's HBM writes)
    const XbPlane<CT> H{reinterpret_cast<CT*>(P.cells) + pb.mat_off}, E{H.p + plane};
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    int32_t* node_fmax = P.node_fmax + pb.node_off;
    uint16_t* front = P.xb_front + pb.graph_off;            // what of a column is in memory: vectors [front & 255, front >> 8)
    const int32_t i0 = (int32_t)lane * R;
    // "Unreachable" is any value at or below MNEG / 2.  Stored cells that nothing reaches hold MNEG (or MNEG plus a few hundred: an unreachable
    // input plus scores stays unreachable for as many steps as a diagonal is long), live ones are above -2^16: the column arithmetic below
    // needs no test of its inputs, only the front's ballot and the traceback's equalities look at liveness.  Rows beyond L inside the last
    // vector get a profile and a vertical-gap cost of XB_OUT, which sends whatever reaches them below MNEG / 2.
    constexpr int32_t XB_OUT = (1 << 27) + (1 << 16);
    int32_t prof[R][5];                                        // row i consumes read base i - 1; the bonus rides on the last one
    int32_t fsub[R];                                           // F(i) = (max over rows j < i of H(j) + j ge) - fsub(i)
    const int32_t i0ge = i0 * ge;
    for (int k = 0; k < R; ++k) {
        const int32_t i = i0 + k;
        for (int g = 0; g < 5; ++g)
            prof[k][g] = (i >= 1 && i <= L) ? (int32_t)(ql ? P.mat[25 * ql[i - 1] + 5 * g + rd[i - 1]] : P.mat[5 * g + rd[i - 1]]) + (i == L ? pb.start_bonus : 0) : -XB_OUT;
        fsub[k] = (i >= 1 && i <= L) ? go + (i - 1) * ge : XB_OUT;
    }
    int32_t Hp[R], Ep[R];
    unsigned long long in_band = 0;
    int32_t best = 0, best_c = -1, best_v = 0; uint32_t best_sb = 0, best_eb = 0;
    // The graph's bases come through LDS, a stretch of XL::stage_cap() columns at a time.  A global load per column would do more than
    // cost its own latency: vmcnt counts loads and stores in one order, so waiting for ANY load waits for the previous column's six
    // 16-byte stores to be acknowledged first — measured as 75 % of the wavefronts' cycles in s_waitcnt before this (SQ_WAIT_ANY).
    uint8_t* const stage = xl.stage(); const uint32_t stage_cap = xl.stage_cap();
    uint32_t stage_base = 0;
    auto refill = [&](uint32_t from) {
        xl.stage_sync();
        for (uint32_t j = lane; j < stage_cap && from + j < pb.R; j += xl.width()) stage[j] = gr[from + j];
        xl.stage_sync();
        stage_base = from;
    };
    refill(0);
    const int32_t band_cells = i0 < stride ? (L < i0 + 7 ? L : i0 + 7) - i0 + 1 : 0;
    // ONE loop over the problem's columns, node after node: the four problems of a wavefront (xdrop_band_kernel16) step through their
    // columns side by side whatever their node boundaries — as two nested loops, a problem that had finished a 3-column node waited for
    // its neighbour's 60-column one before starting its next.
    {
    uint32_t v = 0, c = 0;
    MNode nd{}; nd.col_start = nd.col_end = 0;
    bool front_live = false, entered = false, fenced = true;
    int32_t fmax = 0;
    int32_t* const cache = xl.col_cache(); const uint32_t cache_w = xl.width() * 8u;      // two slots of H | E, one int per row of the wavefront (row of lanes)
    int32_t tag0 = -1, tag1 = -1, cfm0 = MNEG, cfm1 = MNEG;                               // the node each slot holds and its node_fmax
    for (;;) {
        while (v < pb.n_nodes && (!entered || c == nd.col_end)) {
            if (entered) {
                const int32_t nf = front_live ? fmax : MNEG;                 // an empty front is not merged into its successors (src/dozeu_interface.cpp:261-269)
                if (lane == 0) node_fmax[v] = nf;
                // The node's last column (the masked Hp / Ep) also goes into the LDS slot v & 1: the two nodes before a node are the
                // predecessors of nearly every node of a variation graph (both sides of a SNP or an indel), and reading them back from
                // memory meant waiting for this wavefront's own stores to be acknowledged and then for the loads.
                { int32_t* ch = cache + (v & 1u) * 2u * cache_w;
                  for (int k = 0; k < R; ++k) { ch[i0 + k] = Hp[k]; ch[cache_w + i0 + k] = Ep[k]; }
                  if (v & 1u) { tag1 = (int32_t)v; cfm1 = nf; } else { tag0 = (int32_t)v; cfm0 = nf; } }
                xl.lds_sync();
                fenced = false;
                ++v;
                if (v >= pb.n_nodes) break;
            }
            entered = true;
            nd = nodes[v]; c = nd.col_start; front_live = false;
            fmax = 0;                                          // a source node: the root's best is "nothing consumed", 0
            if (nd.n_pred) {
                fmax = MNEG;
                for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const int32_t pr = (int32_t)P.preds[nd.pred_begin + q];
                    int32_t f;
                    if (pr == tag0) f = cfm0; else if (pr == tag1) f = cfm1;
                    else { if (!fenced) { xl.fence(); fenced = true; } f = node_fmax[pr]; }       // an older node: through memory, once this wavefront's stores are there
                    fmax = f > fmax ? f : fmax;
                }
            }
        }
        if (v >= pb.n_nodes) break;
        {
            const bool first = c == nd.col_start;
            int32_t e[R], dg[R];
            if (!first) {
                int32_t up = xl.down(Hp[R - 1]);
                if (lane == 0) up = MNEG;
                for (int k = 0; k < R; ++k) {
                    const int32_t a = Hp[k] - go, b = Ep[k] - ge; e[k] = a > b ? a : b;
                    dg[k] = k ? Hp[k - 1] : up;
                }
            } else if (nd.n_pred == 0) {                       // dozeu's root column (dz_align_init): i leading inserted bases cost go + (i - 1) ge
                for (int k = 0; k < R; ++k) {
                    const int32_t i = i0 + k;
                    const int32_t hr = i == 0 ? 0 : (i <= pb.gap_cells && i <= L ? -(go + (i - 1) * ge) : MNEG);
                    e[k] = hr > MNEG / 2 && i <= L ? hr - go : MNEG;                          // E of the column after the root
                    const int32_t im = i - 1;
                    dg[k] = i >= 1 ? (im == 0 ? 0 : (im <= pb.gap_cells ? -(go + (im - 1) * ge) : MNEG)) : MNEG;
                }
            } else {
                for (int k = 0; k < R; ++k) { e[k] = MNEG; dg[k] = MNEG; }
                // a predecessor's last column as whole vectors (columns are stored padded to vectors, rows beyond L unreachable): five loads
                // in flight and one wait, where row-by-row reads were two dozen round trips
                if (i0 < stride) for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const int32_t pr = (int32_t)P.preds[nd.pred_begin + q];
                    MVec8 ph, pe; int32_t above;
                    if (pr == tag0 || pr == tag1) {                  // one of the two nodes before this one: its last column is in LDS
                        const int32_t* ch = cache + (pr == tag1 ? 2u * cache_w : 0u);
                        for (int k = 0; k < R; ++k) { ph.v[k] = ch[i0 + k]; pe.v[k] = ch[cache_w + i0 + k]; }
                        above = i0 >= 1 ? ch[i0 - 1] : MNEG;
                    } else {
                        const uint32_t pcol = nodes[pr].col_end - 1;
                        const uint64_t pc = (uint64_t)pcol * (uint64_t)stride + (uint64_t)i0;
                        const uint32_t fr = front[pcol], fb = fr & 255u, fe = fr >> 8;       // vectors outside the stored front are unreachable
                        if (lane >= fb && lane < fe) { H.load8(pc, ph.v); E.load8(pc, pe.v); }
                        else for (int k = 0; k < R; ++k) { ph.v[k] = MNEG; pe.v[k] = MNEG; }
                        above = (lane >= fb + 1 && lane < fe + 1) ? H.get(pc - 1) : MNEG;
                    }
                    for (int k = 0; k < R; ++k) {
                        const int32_t a = ph.v[k] - go, b = pe.v[k] - ge; int32_t x = a > b ? a : b; x = x > MNEG / 2 ? x : MNEG;
                        if (x > e[k]) e[k] = x;
                        const int32_t d = k ? ph.v[k - 1] : above;
                        if (d > dg[k]) dg[k] = d;
                    }
                }
            }
            if (c - stage_base >= stage_cap) refill(c);       // (the columns of a problem are visited in ascending order)
            const uint32_t ref = stage[c - stage_base];
            int32_t ht[R], pre[R], run = MNEG;
            for (int k = 0; k < R; ++k) {
                // the row's score against this column's base, by selects: indexing prof[k][ref] with a run-time ref would put the table in scratch memory
                const int32_t sc = ref == 0 ? prof[k][0] : ref == 1 ? prof[k][1] : ref == 2 ? prof[k][2] : ref == 3 ? prof[k][3] : prof[k][4];
                int32_t h = dg[k] + sc;                           // (row 0 has no diagonal: its dg is MNEG)
                if (e[k] > h) h = e[k];
                ht[k] = h;
                pre[k] = run;
                const int32_t gk = h + i0ge + k * ge;
                run = gk > run ? gk : run;
            }
            const int32_t excl = xl.scan_excl(run);
            const int32_t keep = fmax - pb.xt > MNEG / 2 ? fmax - pb.xt : MNEG / 2 + 1;      // live and within xt of the best so far
            int32_t hh[R]; bool alive = false; int32_t lane_max = MNEG;
            for (int k = 0; k < R; ++k) {
                const int32_t pm = excl > pre[k] ? excl : pre[k];
                const int32_t f = pm - fsub[k];
                const int32_t h = ht[k] > f ? ht[k] : f;
                hh[k] = h;
                alive = alive || h >= keep;
                if (h > lane_max) lane_max = h;
            }
            // the front: vectors from the first to the last live one
            const unsigned long long live = xl.ballot(alive);
            uint32_t sb = 64, eb = 0;
            if (live) { sb = (uint32_t)__builtin_ctzll(live); eb = 64u - (uint32_t)__builtin_clzll(live); }       // first and last set bit: one instruction each
            const bool inside = lane >= sb && lane < eb;
            for (int k = 0; k < R; ++k) {
                const bool cell = inside && i0 + k <= L;         // (rows beyond L inside the last vector hold "unreachable")
                if (!cell) { hh[k] = MNEG; e[k] = MNEG; }
                Hp[k] = hh[k]; Ep[k] = e[k];
            }
            if (inside && i0 < stride) {                         // only the front goes to memory; its extent beside it
                const uint64_t at = (uint64_t)c * (uint64_t)stride + (uint64_t)i0;
                H.store8(at, hh); E.store8(at, e);
                in_band += (unsigned long long)band_cells;
            }
            if (lane == 0) front[c] = (uint16_t)(live ? (sb | (eb << 8)) : 0u);
            const int32_t colmax = xl.reduce_max(inside ? lane_max : MNEG);
            fmax = colmax > fmax ? colmax : fmax;
            front_live = live != 0;
            if (colmax > best) { best = colmax; best_c = (int32_t)c; best_v = (int32_t)v; best_sb = sb; best_eb = eb; }      // the end cell's column: the first one that beats every earlier one
            ++c;
        }
    }
    }
    const unsigned long long tot = xl.reduce_add(in_band);
    if (lane == 0) { pb_out.status = VGK_OK; bump_stat(P.stats, tot); }
    if (!P.xb_results) return;
    // ---- end cell (round 3): first node in order / first column / smallest row with the best score.  The traceback from it is
    // xdrop_band_walk_one's, a kernel of its own with one LANE per problem: walked by lane 0 of this wavefront, four walks (one per row of
    // 16 lanes) kept a 166-VGPR wavefront resident for a quarter of the launch
    int32_t best_i = 0x7fffffff;
    xl.fence();                                                // (the end column is read back from memory, across lanes)
    if (best_c >= 0) for (int32_t i = (int32_t)(8u * best_sb + lane); i < rows && i < (int32_t)(8u * best_eb); i += (int32_t)xl.width()) if (H.get((uint64_t)best_c * (uint64_t)stride + i) == best && i < best_i) best_i = i;
    best_i = -xl.reduce_max(-best_i);
    if (lane == 0) { pb_out.best = best; pb_out.best_c = best_c; pb_out.best_v = best_v; pb_out.best_i = best_i; }
}
// ---- the same fill with TWO ROWS TO A REGISTER (round 4; GsswMatrixParams::xb_cell16 == 2) -------------------------------------------
// The int32 column above is 207 VALU instructions for 8 rows per lane; maxima, compares and permutes issue at 4 cycles on gfx950 whether
// they carry one 32-bit cell or two 16-bit ones.  Here a lane's 8 rows are 4 words (rows 2k | 2k + 1 in the low | high half), every
// quantity UNSIGNED and biased: x' = x + XBP_OFF for a reachable cell, and anything below XBP_LIVE is unreachable (0 where it is made).
//   * subtractions saturate at 0 (v_pk_sub_u16 clamp): an unreachable cell stays unreachable without a test;
//   * what an unreachable cell can gain on its way down the rows is at most L steps of the largest score, below XBP_LIVE by the caller's
//     bound ((read + graph + 10) x (largest |score| + gap_open + gap_extend + bonus) < 16 000 — the bound that also keeps every reachable
//     cell inside (XBP_LIVE, XBP_OFF + 16 000)): the normalisations of the int32 form have nothing to do;
//   * a row's score against the column's base is ONE byte permute per pair (the four biased scores of a row are the bytes of a word, the
//     selector carries the base); N columns — rare — select a fifth word;
//   * the serial prefix maximum over the lane's rows becomes 4 + 3 packed maxima whose op_sel reads the half it needs (pk16.hpp);
//   * the rows that do not exist (0 for the diagonal, beyond L) die through a saturating subtrahend of 0xffff instead of a compare.
// Cells leave as int16 (x' ^ 0x8000 = x, two's complement), the layout of xb_cell16 == 1: the walk kernel reads either.  Exactness: the map
// x -> x' is monotone on reachable cells and every unreachable one lies below every reachable one in both forms, so maxima, the front's
// ballot, the end cell and the stored reachable cells are the int32 form's.
constexpr uint32_t XBP_OFF = 32768u, XBP_LIVE = XBP_OFF - 16000u;
VGK_HD uint32_t xbp_bias(int32_t v) { return v > MNEG / 2 ? (uint32_t)(v + (int32_t)XBP_OFF) : 0u; }
template <class XL>
VGK_HD void xdrop_band_pk_lane(const GsswMatrixParams& P, uint32_t pi, uint32_t lane, XL& xl) {
    MProb& pb_out = P.probs[pi];
    const MProb pb = pb_out;
    const int32_t L = (int32_t)pb.L, rows = L + 1, go = P.go, ge = P.ge;
    const int32_t stride = (rows + 7) & ~7;
    const uint64_t plane = (uint64_t)pb.R * (uint64_t)stride;
    uint16_t* const Hm = reinterpret_cast<uint16_t*>(P.cells) + pb.mat_off; uint16_t* const Em = Hm + plane;      // biased cells as they are (XbPlane<uint16_t>)
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    int32_t* node_fmax = P.node_fmax + pb.node_off;            // biased here (0: an empty front)
    uint16_t* front = P.xb_front + pb.graph_off;
    const int32_t i0 = (int32_t)lane * 8;
    const uint32_t SB = (uint32_t)P.xb_sb;                      // what makes every score a byte: score + SB in [0, 255]
    const uint32_t go2 = (uint32_t)go * 0x00010001u, ge2 = (uint32_t)ge * 0x00010001u;
    uint32_t prof_lo[4], prof_hi[4], prof_n[4], sbrow[4], fsub[4], ige[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
        uint32_t by[2] = {0, 0}, nn[2] = {0, 0}, sbr[2], fs[2], ig[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int h = 0; h < 2; ++h) {
            const int32_t i = i0 + 2 * k + h;
            const bool valid = i >= 1 && i <= L;
            if (valid) {
                const int8_t* row = ql ? P.mat + 25 * ql[i - 1] : P.mat; const uint32_t r = rd[i - 1];
                const int32_t bon = i == L ? pb.start_bonus : 0;
                for (int g = 0; g < 4; ++g) by[h] |= ((uint32_t)((int32_t)row[5 * g + r] + bon + (int32_t)SB) & 0xffu) << (8 * g);
                nn[h] = (uint32_t)((int32_t)row[20 + r] + bon + (int32_t)SB);
            }
            sbr[h] = valid ? SB : 0xffffu;
            fs[h] = valid ? (uint32_t)(go + (i - 1) * ge) : 0xffffu;
            ig[h] = (uint32_t)(i * ge);
        }
        prof_lo[k] = by[0]; prof_hi[k] = by[1]; prof_n[k] = nn[0] | (nn[1] << 16);
        sbrow[k] = sbr[0] | (sbr[1] << 16); fsub[k] = fs[0] | (fs[1] << 16); ige[k] = ig[0] | (ig[1] << 16);
    }
    uint32_t Hp[4] = {0, 0, 0, 0}, Ep[4] = {0, 0, 0, 0};
    unsigned long long in_band = 0;
    int32_t best = (int32_t)XBP_OFF, best_c = -1, best_v = 0; uint32_t best_sb = 0, best_eb = 0;
    uint8_t* const stage = xl.stage(); const uint32_t stage_cap = xl.stage_cap();
    uint32_t stage_base = 0; bool stage_n = false;             // stage_n: the staged stretch holds an N
    auto refill = [&](uint32_t from) {
        xl.stage_sync();
        int32_t seen = 0;
        for (uint32_t j = lane; j < stage_cap && from + j < pb.R; j += xl.width()) { const uint8_t b = gr[from + j]; stage[j] = b; seen |= b == 4; }
        stage_n = xl.any(seen);
        xl.stage_sync();
        stage_base = from;
    };
    refill(0);
    uint32_t ref_cur = stage[0];
    uint32_t col_at = (uint32_t)i0;                                 // c * stride + i0 (a problem's planes hold fewer than 2^28 cells each)
    const int32_t band_cells = i0 < stride ? (L < i0 + 7 ? L : i0 + 7) - i0 + 1 : 0;
    {
    uint32_t v = 0, c = 0;
    MNode nd{}; nd.col_start = nd.col_end = 0;
    MNode nd_next = nodes[0];
    bool front_live = false, entered = false, fenced = true;
    int32_t fmax = 0;
    uint32_t* const cache = reinterpret_cast<uint32_t*>(xl.col_cache()); const uint32_t cache_w = xl.width() * 4u;      // two slots of H | E, a word per row pair
    int32_t tag0 = -1, tag1 = -1, cfm0 = 0, cfm1 = 0;
    for (;;) {
        while (v < pb.n_nodes && (!entered || c == nd.col_end)) {
            if (entered) {
                const int32_t nf = front_live ? fmax : 0;
                if (lane == 0) node_fmax[v] = nf;
                { uint32_t* ch = cache + (v & 1u) * 2u * cache_w;
                  for (int k = 0; k < 4; ++k) { ch[lane * 4u + k] = Hp[k]; ch[cache_w + lane * 4u + k] = Ep[k]; }
                  if (v & 1u) { tag1 = (int32_t)v; cfm1 = nf; } else { tag0 = (int32_t)v; cfm0 = nf; } }
                xl.lds_sync();
                fenced = false;
                ++v;
                if (v >= pb.n_nodes) break;
            }
            entered = true;
            // (the node's record was asked for when the node before it began)
            nd = nd_next; if (v + 1 < pb.n_nodes) nd_next = nodes[v + 1];
            c = nd.col_start; front_live = false;
            fmax = (int32_t)XBP_OFF;                           // a source node: the root's best is "nothing consumed", 0
            if (nd.n_pred) {
                fmax = 0;
                for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const int32_t pr = (int32_t)P.preds[nd.pred_begin + q];
                    int32_t f;
                    if (pr == tag0) f = cfm0; else if (pr == tag1) f = cfm1;
                    else { if (!fenced) { xl.fence(); fenced = true; } f = node_fmax[pr]; }
                    fmax = f > fmax ? f : fmax;
                }
            }
        }
        if (v >= pb.n_nodes) break;
        {
            const bool first = c == nd.col_start;
            uint32_t e[4], dg[4];
            if (!first) {
                uint32_t up = (uint32_t)xl.down((int32_t)Hp[3]);
                if (lane == 0) up = 0;
                dg[0] = align16(Hp[0], up);
                for (int k = 1; k < 4; ++k) dg[k] = align16(Hp[k], Hp[k - 1]);
                for (int k = 0; k < 4; ++k) e[k] = pk_max(pk_subs(Hp[k], go2), pk_subs(Ep[k], ge2));
            } else if (nd.n_pred == 0) {                       // dozeu's root column (dz_align_init)
                for (int k = 0; k < 4; ++k) {
                    uint32_t ev[2], dv[2];
                    for (int h = 0; h < 2; ++h) {
                        const int32_t i = i0 + 2 * k + h;
                        const int32_t hr = i == 0 ? 0 : (i <= pb.gap_cells && i <= L ? -(go + (i - 1) * ge) : MNEG);
                        ev[h] = xbp_bias(hr > MNEG / 2 && i <= L ? hr - go : MNEG);
                        const int32_t im = i - 1;
                        dv[h] = xbp_bias(i >= 1 ? (im == 0 ? 0 : (im <= pb.gap_cells ? -(go + (im - 1) * ge) : MNEG)) : MNEG);
                    }
                    e[k] = ev[0] | (ev[1] << 16); dg[k] = dv[0] | (dv[1] << 16);
                }
            } else {
                for (int k = 0; k < 4; ++k) { e[k] = 0; dg[k] = 0; }
                if (i0 < stride) for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const int32_t pr = (int32_t)P.preds[nd.pred_begin + q];
                    uint32_t ph[4], pe[4], above;
                    if (pr == tag0 || pr == tag1) {
                        const uint32_t* ch = cache + (pr == tag1 ? 2u * cache_w : 0u);
                        for (int k = 0; k < 4; ++k) { ph[k] = ch[lane * 4u + k]; pe[k] = ch[cache_w + lane * 4u + k]; }
                        above = lane >= 1 ? ch[lane * 4u - 1u] >> 16 : 0u;
                    } else {
                        const uint32_t pcol = nodes[pr].col_end - 1;
                        const uint64_t pc = (uint64_t)pcol * (uint64_t)stride + (uint64_t)i0;
                        const uint32_t fr = front[pcol], fb = fr & 255u, fe = fr >> 8;
                        if (lane >= fb && lane < fe) {
                            const MVec8h xh = *reinterpret_cast<const MVec8h*>(Hm + pc), xe = *reinterpret_cast<const MVec8h*>(Em + pc);
                            for (int k = 0; k < 4; ++k) { ph[k] = xh.w[k]; pe[k] = xe.w[k]; }
                        } else for (int k = 0; k < 4; ++k) { ph[k] = 0; pe[k] = 0; }
                        above = (lane >= fb + 1 && lane < fe + 1) ? (uint32_t)Hm[pc - 1] : 0u;
                    }
                    for (int k = 0; k < 4; ++k) {
                        e[k] = pk_max(e[k], pk_max(pk_subs(ph[k], go2), pk_subs(pe[k], ge2)));
                        const uint32_t d = k ? align16(ph[k], ph[k - 1]) : ((ph[0] << 16) | above);
                        dg[k] = pk_max(dg[k], d);
                    }
                }
            }
            const uint32_t ref = ref_cur;                         // (read from the stage while the column before this one finished)
            const uint32_t sel = 0x0c040c00u + ref * 0x00010001u;      // byte `ref` of the low row's word | byte `ref` of the high row's
            uint32_t ht[4], q2[4];
            for (int k = 0; k < 4; ++k) {
                uint32_t sc = byte_perm(prof_hi[k], prof_lo[k], sel);
                if (stage_n) sc = ref == 4u ? prof_n[k] : sc;
                ht[k] = pk_max(pk_subs(dg[k] + sc, sbrow[k]), e[k]);          // (the sum cannot carry: a cell is below 2^16 - 256)
                q2[k] = pk_max_lo_into_hi(ht[k] + ige[k]);
            }
            uint32_t incl[4], pre[4];
            incl[0] = q2[0];
            for (int k = 1; k < 4; ++k) incl[k] = pk_max_bhi(q2[k], incl[k - 1]);
            pre[0] = incl[0] << 16;
            for (int k = 1; k < 4; ++k) pre[k] = align16(incl[k], incl[k - 1]);
            int32_t excl = xl.scan_excl((int32_t)(incl[3] >> 16));
            excl = excl < 0 ? 0 : excl;                           // (the first lane: nothing above)
            const uint32_t excl2 = (uint32_t)excl | ((uint32_t)excl << 16);
            const int32_t keep = fmax - pb.xt > (int32_t)XBP_LIVE ? fmax - pb.xt : (int32_t)XBP_LIVE;
            uint32_t hh[4];
            for (int k = 0; k < 4; ++k) hh[k] = pk_max(ht[k], pk_subs(pk_max(excl2, pre[k]), fsub[k]));
            const uint32_t m2 = pk_max(pk_max(hh[0], hh[1]), pk_max(hh[2], hh[3]));
            const uint32_t mlo = m2 & 0xffffu, mhi = m2 >> 16;
            const int32_t lane_max = (int32_t)(mlo > mhi ? mlo : mhi);
            const unsigned long long live = xl.ballot(lane_max >= keep);
            uint32_t sb = 64, eb = 0;
            if (live) { sb = (uint32_t)__builtin_ctzll(live); eb = 64u - (uint32_t)__builtin_clzll(live); }
            const bool inside = lane >= sb && lane < eb;
            const uint32_t keep_mask = inside ? 0xffffffffu : 0u;      // (rows beyond L inside the last vector are 0 by themselves)
            for (int k = 0; k < 4; ++k) { Hp[k] = hh[k] & keep_mask; Ep[k] = e[k] & keep_mask; }
            if (inside && i0 < stride) {
                MVec8h xh, xe;
                for (int k = 0; k < 4; ++k) { xh.w[k] = Hp[k]; xe.w[k] = Ep[k]; }
                *reinterpret_cast<MVec8h*>(Hm + col_at) = xh; *reinterpret_cast<MVec8h*>(Em + col_at) = xe;      // (col_at: this lane's vector of column c, kept as a running sum)
                in_band += (unsigned long long)band_cells;
            }
            if (lane == 0) front[c] = (uint16_t)(live ? (sb | (eb << 8)) : 0u);
            const int32_t colmax = xl.reduce_max(inside ? lane_max : 0);
            fmax = colmax > fmax ? colmax : fmax;
            front_live = live != 0;
            if (colmax > best) { best = colmax; best_c = (int32_t)c; best_v = (int32_t)v; best_sb = sb; best_eb = eb; }
            ++c; col_at += (uint32_t)stride;
            if (c < pb.R) { if (c - stage_base >= stage_cap) refill(c); ref_cur = stage[c - stage_base]; }      // (the columns of a problem are visited in ascending order)
        }
    }
    }
    const unsigned long long tot = xl.reduce_add(in_band);
    if (lane == 0) { pb_out.status = VGK_OK; bump_stat(P.stats, tot); }
    if (!P.xb_results) return;
    const int32_t best_true = best - (int32_t)XBP_OFF;
    int32_t best_i = 0x7fffffff;
    xl.fence();
    const XbPlane<uint16_t> H{Hm};
    if (best_c >= 0) for (int32_t i = (int32_t)(8u * best_sb + lane); i < rows && i < (int32_t)(8u * best_eb); i += (int32_t)xl.width()) if (H.get((uint64_t)best_c * (uint64_t)stride + i) == best_true && i < best_i) best_i = i;
    best_i = -xl.reduce_max(-best_i);
    if (lane == 0) { pb_out.best = best_true; pb_out.best_c = best_c; pb_out.best_v = best_v; pb_out.best_i = best_i; }
}
template <class XL>
VGK_HD void xdrop_band_wave_lane(const GsswMatrixParams& P, uint32_t pi, uint32_t lane, XL& xl) {
    if (P.xb_cell16 == 2) xdrop_band_pk_lane(P, pi, lane, xl);
    else if (P.xb_cell16) xdrop_band_wave_lane_t<int16_t>(P, pi, lane, xl); else xdrop_band_wave_lane_t<int32_t>(P, pi, lane, xl);
}

// The traceback of one problem over the H / E columns its fill left in HBM (the host's BandTracer of round 2 stated the same rules):
// diagonal > deletion > insertion, gap open before extend, first explaining predecessor; the walk ends at the root.
template <class CT>
VGK_HD void xdrop_band_walk_one_t(const GsswMatrixParams& P, uint32_t pi) {
    const MProb pb = P.probs[pi];
    const int32_t L = (int32_t)pb.L, rows = L + 1, go = P.go, ge = P.ge;
    const int32_t stride = (rows + 7) & ~7;
    const uint64_t plane = (uint64_t)pb.R * (uint64_t)stride;
    const XbPlane<CT> H{reinterpret_cast<CT*>(P.cells) + pb.mat_off}, E{H.p + plane};
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    const int32_t best = pb.best, best_c = pb.best_c, best_v = pb.best_v, best_i = pb.best_i;
    vgk_result res{};
    res.end_node = -1; res.end_offset = -1; res.end_read = -1; res.status = VGK_OK; res.ops_begin = (uint32_t)P.xb_ops_off[pi];
    if (best >= 32767) { res.status = VGK_EOVERFLOW; P.xb_results[pi] = res; return; }
    if (best <= 0 || best_c < 0) { P.xb_results[pi] = res; return; }                       // the root wins: the caller writes the full-length insertion
    res.score = best; res.end_node = best_v; res.end_offset = best_c - (int32_t)nodes[best_v].col_start; res.end_read = best_i - 1;
    if (!P.xb_want_tb[pi]) { P.xb_results[pi] = res; return; }
    vgk_op* ops = P.xb_ops + P.xb_ops_off[pi];
    uint32_t n_ops = 0;
    auto live = [](int32_t x) { return x > MNEG / 2; };
    auto push = [&](int32_t node, int op, uint32_t len) {
        if (!len) return;
        if (n_ops && ops[n_ops - 1].node == (uint32_t)node && ops[n_ops - 1].op == (uint8_t)op) { ops[n_ops - 1].len = (uint16_t)(ops[n_ops - 1].len + len); return; }
        vgk_op x{}; x.node = (uint32_t)node; x.op = (uint8_t)op; x.len = (uint16_t)len; ops[n_ops++] = x;
    };
    const uint16_t* front = P.xb_front + pb.graph_off;
    auto stored = [&](int32_t c, int32_t i) { const uint32_t fr = front[c], vec = (uint32_t)i >> 3; return vec >= (fr & 255u) && vec < (fr >> 8); };      // else: unreachable, and not in memory
    auto hc = [&](int32_t c, int32_t i) { return stored(c, i) ? H.get((uint64_t)c * (uint64_t)stride + i) : MNEG; };
    auto ec = [&](int32_t c, int32_t i) { return stored(c, i) ? E.get((uint64_t)c * (uint64_t)stride + i) : MNEG; };
    auto e_next = [&](int32_t c, int32_t i) { const int32_t a = live(hc(c, i)) ? hc(c, i) - go : MNEG, b = live(ec(c, i)) ? ec(c, i) - ge : MNEG; return a > b ? a : b; };     // E of the column after c
    auto root_h = [&](int32_t i) { return i == 0 ? 0 : (i <= pb.gap_cells && i <= L ? -(go + (i - 1) * ge) : MNEG); };
    // plain contexts: the 5 x 5 table in registers, a row per reference code in a 64-bit word — read from memory, the score was a third
    // dependent round trip of every diagonal step (after the cell and the two codes)
    uint64_t mrow[5] = {0, 0, 0, 0, 0};
    if (!ql) for (int gq = 0; gq < 5; ++gq) for (int rq = 0; rq < 5; ++rq) mrow[gq] |= (uint64_t)(uint8_t)P.mat[5 * gq + rq] << (8 * rq);
    auto score = [&](int32_t i, int32_t c) {
        const uint32_t gc = gr[c], rc = rd[i - 1];
        int32_t sc;
        if (ql) sc = (int32_t)P.mat[25 * ql[i - 1] + 5 * gc + rc];
        else { const uint64_t rw = gc == 0 ? mrow[0] : gc == 1 ? mrow[1] : gc == 2 ? mrow[2] : gc == 3 ? mrow[3] : mrow[4]; sc = (int32_t)(int8_t)(uint8_t)(rw >> (8u * rc)); }
        return sc + (i == L ? pb.start_bonus : 0);
    };
    int32_t c = best_c, i = best_i, n = best_v, cur = best;
    int status = VGK_OK;
    push(n, VGK_OP_S, (uint32_t)(L - i));
    enum { ST_H, ST_E, ST_F } st = ST_H;
    const uint32_t cap = pb.L + pb.R + 3u;
    MNode nd = nodes[n]; int32_t nd_of = n;
    for (uint32_t guard = 0; status == VGK_OK; ++guard) {
        if (guard > 2u * cap + 8u || n_ops + 2u > cap) { status = VGK_EINVAL; break; }     // (never: every step consumes a base or changes state once)
        if (n != nd_of) { nd = nodes[n]; nd_of = n; }              // (the node's record only when the walk enters another node)
        // A tail's traceback is mostly diagonal runs inside a node, and every step of one was a round trip to HBM (the column's extent,
        // then the cell; the two codes beside them) on which the next step's addresses depended.  Here the inputs of the next four diagonal
        // steps are asked for together — cells, extents and codes at (c - 1 - t, i - 1 - t) — and consumed as long as each step IS the
        // diagonal one; whatever else happens falls through to the general step below with nothing changed.  (One lane per problem: the
        // kernel's time is its longest chain of such round trips.)
        if (st == ST_H && i > 0 && (uint32_t)c > nd.col_start) {
            constexpr int K = 4;
            uint32_t fr[K]; decltype(H.raw(0)) raw[K]; uint32_t gcode[K], rcode[K], qv[K];
            for (int t = 0; t < K; ++t) {
                const int32_t ct = c - 1 - t > 0 ? c - 1 - t : 0, it = i - 1 - t > 0 ? i - 1 - t : 0;      // (kept inside the planes; what lies beyond the run is not used)
                fr[t] = front[ct]; raw[t] = H.raw((uint64_t)ct * (uint64_t)stride + (uint64_t)it);
                gcode[t] = gr[c - t > 0 ? c - t : 0]; rcode[t] = rd[it]; qv[t] = ql ? ql[it] : 0u;
            }
            int32_t room = i < K ? i : K;
            if (c - (int32_t)nd.col_start < room) room = c - (int32_t)nd.col_start;
            int steps = 0;
            for (int t = 0; t < room; ++t) {
                const uint32_t vec = (uint32_t)(i - 1) >> 3;
                const int32_t d = (vec >= (fr[t] & 255u) && vec < (fr[t] >> 8)) ? H.conv(raw[t]) : MNEG;
                int32_t sc;
                if (ql) sc = (int32_t)P.mat[25 * qv[t] + 5 * gcode[t] + rcode[t]];
                else { const uint32_t gc = gcode[t]; const uint64_t rw = gc == 0 ? mrow[0] : gc == 1 ? mrow[1] : gc == 2 ? mrow[2] : gc == 3 ? mrow[3] : mrow[4]; sc = (int32_t)(int8_t)(uint8_t)(rw >> (8u * rcode[t])); }
                if (i == L) sc += pb.start_bonus;
                if (!(live(d) && cur == d + sc)) break;
                push(n, VGK_OP_M, 1);
                cur = d; i -= 1; c -= 1; ++steps;
            }
            if (steps) { guard += (uint32_t)steps - 1u; continue; }
        }
        const bool first = (uint32_t)c == nd.col_start;
        const uint32_t* pr = P.preds + nd.pred_begin;
        if (st == ST_H) {
            bool moved = false;
            if (i > 0) {
                int32_t d = first ? (nd.n_pred == 0 ? root_h(i - 1) : MNEG) : hc(c - 1, i - 1);
                if (first) for (uint32_t k = 0; k < nd.n_pred; ++k) { const int32_t x = hc((int32_t)nodes[pr[k]].col_end - 1, i - 1); d = x > d ? x : d; }
                if (live(d) && cur == d + score(i, c)) {
                    push(n, VGK_OP_M, 1);
                    cur = d; i -= 1; moved = true;
                    if (!first) c -= 1;
                    else if (nd.n_pred == 0) { push(n, VGK_OP_I, (uint32_t)i); break; }            // back at the root: leading insertion
                    else {
                        int32_t found = -1;
                        for (uint32_t k = 0; k < nd.n_pred; ++k) { const int32_t q = (int32_t)nodes[pr[k]].col_end - 1; if (hc(q, i) == cur) { found = (int32_t)pr[k]; break; } }
                        if (found < 0) { status = VGK_EINVAL; break; }
                        n = found; c = (int32_t)nodes[n].col_end - 1;
                    }
                }
            }
            if (!moved) {
                if (cur == ec(c, i)) st = ST_E;
                else if (i > 0) st = ST_F;                           // neither the diagonal nor the deletion: the vertical gap
                else { status = VGK_EINVAL; break; }
            }
        } else if (st == ST_E) {
            push(n, VGK_OP_D, 1);
            if (first && nd.n_pred == 0) {                           // deletion opened straight from the root column
                if (!live(root_h(i)) || root_h(i) - go != cur) { status = VGK_EINVAL; break; }
                push(n, VGK_OP_I, (uint32_t)i); break;
            }
            int32_t q = c - 1, qn = n;
            if (first) {
                q = -1;
                for (uint32_t k = 0; k < nd.n_pred; ++k) { const int32_t x = (int32_t)nodes[pr[k]].col_end - 1; if (e_next(x, i) == cur) { q = x; qn = (int32_t)pr[k]; break; } }
                if (q < 0) { status = VGK_EINVAL; break; }
            }
            if (live(hc(q, i)) && hc(q, i) - go == cur) { st = ST_H; cur += go; } else cur += ge;
            c = q; n = qn;
        } else {
            push(n, VGK_OP_I, 1);
            if (i == 0) { status = VGK_EINVAL; break; }
            if (live(hc(c, i - 1)) && hc(c, i - 1) - go == cur) { st = ST_H; cur += go; } else cur += ge;
            i -= 1;
        }
    }
    for (uint32_t a = 0, b = n_ops; a + 1 < b; ++a, --b) { const vgk_op t = ops[a]; ops[a] = ops[b - 1]; ops[b - 1] = t; }       // found back to front
    res.status = status; res.n_ops = status == VGK_OK ? n_ops : 0; res.first_offset = 0;
    P.xb_results[pi] = res;
}
VGK_HD void xdrop_band_walk_one(const GsswMatrixParams& P, uint32_t pi) {
    if (P.xb_cell16 == 2) xdrop_band_walk_one_t<uint16_t>(P, pi); else if (P.xb_cell16) xdrop_band_walk_one_t<int16_t>(P, pi); else xdrop_band_walk_one_t<int32_t>(P, pi);
}

}  // namespace vgk
