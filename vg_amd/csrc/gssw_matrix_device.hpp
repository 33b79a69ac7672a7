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
VGK_HD void bump_stat(unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
template <int R, class XL>
VGK_HD void gssw_matrix_wave_lane(const GsswMatrixParams& P, uint32_t i, uint32_t lane, XL& xl) {
    MProb& pb = P.probs[i];
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
    if (xl.any(overflow) && lane == 0) pb.status = VGK_EOVERFLOW;
    else if (lane == 0) pb.status = VGK_OK;
}

// ---- X-drop with dozeu's band (vgk_xdrop_band_align; the rules are stated in include/vgk.h and, identically, in oracle/vgo_xdrop.c) ----
// Same shape as the wavefront fill above with R = 8: a lane IS one of dozeu's 8-cell vectors.  Rows i = 0 .. L count consumed read
// bases (L + 1 rows, cell (c, i) at c * (L + 1) + i); no zero floor; source nodes start from the root column.  A column is computed
// whole — cells without a live input come out unreachable by themselves — then one ballot over "my vector holds a cell >= best - xt"
// gives the first and last live vector, lanes outside store (and keep) unreachable cells, and the best score of the front is updated
// by a wave maximum.  A node whose incoming fronts are all empty costs no arithmetic worth mentioning: its cells stay unreachable.
template <class XL>
VGK_HD void xdrop_band_wave_lane(const GsswMatrixParams& P, uint32_t pi, uint32_t lane, XL& xl) {
    constexpr int R = 8;
    MProb& pb = P.probs[pi];
    const int32_t L = (int32_t)pb.L, rows = L + 1, go = P.go, ge = P.ge;
    const uint64_t plane = (uint64_t)pb.R * (uint64_t)rows;
    int32_t* H = P.cells + pb.mat_off; int32_t* E = H + plane; int32_t* F = E + plane;
    const uint8_t* rd = P.reads + pb.read_off; const uint8_t* ql = P.quals ? P.quals + pb.read_off : nullptr;
    const uint8_t* gr = P.graph + pb.graph_off;
    const MNode* nodes = P.nodes + pb.node_off;
    int32_t* node_fmax = P.node_fmax + pb.node_off;
    const int32_t i0 = (int32_t)lane * R;
    int32_t prof[R][5];                                        // row i consumes read base i - 1; the bonus rides on the last one
    for (int k = 0; k < R; ++k) {
        const int32_t i = i0 + k;
        for (int g = 0; g < 5; ++g)
            prof[k][g] = (i >= 1 && i <= L) ? (int32_t)(ql ? P.mat[25 * ql[i - 1] + 5 * g + rd[i - 1]] : P.mat[5 * g + rd[i - 1]]) + (i == L ? pb.start_bonus : 0) : 0;
    }
    int32_t Hp[R], Ep[R];
    unsigned long long in_band = 0;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        const MNode nd = nodes[v];
        bool front_live = false;
        int32_t fmax = 0;                                      // a source node: the root's best is "nothing consumed", 0
        if (nd.n_pred) { fmax = MNEG; for (uint32_t q = 0; q < nd.n_pred; ++q) { const int32_t f = node_fmax[P.preds[nd.pred_begin + q]]; fmax = f > fmax ? f : fmax; } }
        for (uint32_t c = nd.col_start; c < nd.col_end; ++c) {
            const bool first = c == nd.col_start;
            int32_t e[R], dg[R];
            if (!first) {
                int32_t up = xl.down(Hp[R - 1]);
                if (lane == 0) up = MNEG;
                for (int k = 0; k < R; ++k) {
                    const int32_t a = Hp[k] - go, b = Ep[k] - ge; int32_t x = a > b ? a : b; e[k] = x > MNEG / 2 ? x : MNEG;
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
                for (uint32_t q = 0; q < nd.n_pred; ++q) {
                    const uint64_t pc = (uint64_t)(nodes[P.preds[nd.pred_begin + q]].col_end - 1) * (uint64_t)rows;
                    for (int k = 0; k < R; ++k) {
                        const int32_t i = i0 + k;
                        if (i <= L) {
                            const int32_t ph = H[pc + i], pe = E[pc + i];
                            const int32_t a = ph - go, b = pe - ge; int32_t x = a > b ? a : b; x = x > MNEG / 2 ? x : MNEG;
                            if (x > e[k]) e[k] = x;
                        }
                        if (i >= 1 && i - 1 <= L) { const int32_t ph = H[pc + i - 1]; if (ph > dg[k]) dg[k] = ph; }
                    }
                }
            }
            const uint32_t ref = gr[c];
            int32_t ht[R], pre[R], run = MNEG;
            for (int k = 0; k < R; ++k) {
                const int32_t i = i0 + k;
                int32_t h = (i >= 1 && dg[k] > MNEG / 2) ? dg[k] + prof[k][ref] : MNEG;
                if (e[k] > h) h = e[k];
                ht[k] = h;
                pre[k] = run;
                const int32_t gk = h > MNEG / 2 ? h + i * ge : MNEG;
                run = gk > run ? gk : run;
            }
            const int32_t excl = xl.scan_excl(run);
            int32_t hh[R], ff[R]; bool alive = false; int32_t lane_max = MNEG;
            for (int k = 0; k < R; ++k) {
                const int32_t i = i0 + k;
                const int32_t pm = excl > pre[k] ? excl : pre[k];
                int32_t f = (i >= 1 && pm > MNEG / 2) ? pm - go - (i - 1) * ge : MNEG;
                if (f < MNEG / 2) f = MNEG;
                int32_t h = ht[k] > f ? ht[k] : f;
                if (i > L) { h = MNEG; f = MNEG; e[k] = MNEG; }
                hh[k] = h; ff[k] = f;
                if (h > MNEG / 2 && h >= fmax - pb.xt) alive = true;
                if (h > lane_max) lane_max = h;
            }
            // the front: vectors from the first to the last live one
            const unsigned long long live = xl.ballot(alive);
            uint32_t sb = 64, eb = 0;
            if (live) { sb = 0; while (!((live >> sb) & 1ull)) ++sb; eb = 64; while (!((live >> (eb - 1)) & 1ull)) --eb; }
            const bool inside = lane >= sb && lane < eb;
            for (int k = 0; k < R; ++k) {
                const int32_t i = i0 + k;
                if (!inside) { hh[k] = MNEG; ff[k] = MNEG; e[k] = MNEG; }
                if (i <= L) { const uint64_t at = (uint64_t)c * (uint64_t)rows + i; H[at] = hh[k]; E[at] = e[k]; F[at] = ff[k]; if (inside) ++in_band; }
                Hp[k] = hh[k]; Ep[k] = e[k];
            }
            const int32_t colmax = xl.reduce_max(inside ? lane_max : MNEG);
            fmax = colmax > fmax ? colmax : fmax;
            front_live = live != 0;
        }
        if (lane == 0) node_fmax[v] = front_live ? fmax : MNEG;      // an empty front is not merged into its successors (src/dozeu_interface.cpp:261-269)
        xl.fence();
    }
    const unsigned long long tot = xl.reduce_add(in_band);
    if (lane == 0) { pb.status = VGK_OK; bump_stat(P.stats, tot); }
}

}  // namespace vgk
