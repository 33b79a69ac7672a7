// gssw_matrix_device.hpp — the pinned gssw fill with every cell's H / E / F kept, for the k-best tracebacks of
// vgk_gssw_align_multi (Aligner::align_pinned_multi, reference src/aligner.cpp:423-435).
//
// The packed fill kernel of gssw_device.hpp streams 4-bit traceback codes: enough for the one best walk, not for ranking
// alternates, which needs the score lost at every source not taken.  This path is mpmap's (max_alt_alns > 1) and low-volume, so
// the fill here is the plain recurrence, one thread per problem, int32 cells written to HBM column by column:
//   E[r][c] = max(0, H[r][c-1] - go, E[r][c-1] - ge)   F[r][c] = max(0, H[r-1][c] - go, F[r-1][c] - ge)
//   H[r][c] = max(H[r-1][c-1] + s(r, c), E[r][c], F[r][c]);   "c-1" at a node's first column = element-wise max over the
//   predecessors' last columns (gssw_create_seed_*); bonus at read row 0 only (pinned: the other end is the pinned one).
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
    int32_t  start_bonus;
    int32_t  status;                  // out: VGK_OK or VGK_EOVERFLOW
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

}  // namespace vgk
