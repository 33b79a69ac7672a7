// banded_device.hpp — banded global graph alignment on one wavefront per problem
// (replaces BAMatrix::fill_matrix / traceback / traceback_over_edge of the reference's
// src/banded_global_aligner.cpp:251-742, :756-1126, :1129-1780).
//
// Mapping.  The band of a node is the set of diagonals d = r - j in [top, bot] (r = read row, j = node
// column).  Lane l of the wave owns the R consecutive band rows k = l*R .. l*R+R-1 (k = d - top), so
//   * the match state M(r,j) = s + max(M,Ic,Ir)(r-1,j-1) reads the lane's own previous column,
//   * the column gap Ic(r,j) (graph base against a gap) reads row k+1 of the previous column — the next
//     register of the lane, or lane l+1 through one DPP wave_shl:1,
//   * the row gap Ir(r,j) (read base against a gap) runs down the column: a max-plus prefix scan over
//     the band rows (serial inside a lane, DPP row_shr / row_bcast across lanes),
// and the whole wave walks the node's columns together, which keeps node boundaries wave-uniform.
// Per cell one byte of traceback goes to HBM (2 bits per state: which state the optimum came from, in
// the reference's preference order match, insert-column, insert-row :812); per node the last column of
// the three matrices (int32) stays in HBM for the successors and for the traceback across edges.
//
// The same code is stepped on the CPU by tests/emu (one thread per lane, cross-lane primitives through a
// barrier) — test infrastructure only.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "pk16.hpp"
#include "../../include/vgk.h"

namespace vgk {

constexpr int32_t BNEG = -(1 << 28);
VGK_HD bool blive(int32_t v) { return v > BNEG / 2; }
enum : uint32_t { BM = 0, BIC = 1, BIR = 2 };

struct BNode {                 // one per graph node of a problem
    int32_t  top, bot;         // inclusive diagonals (undefined when masked)
    int32_t  len, cum;         // bases; shortest sequence from a source to the node's left edge (:2271-2293)
    uint32_t seq_off;          // node bases, relative to the problem's graph_off
    uint32_t seed_off;         // first flattened predecessor, relative to the problem's seed_base
    uint16_t n_seeds;
    uint8_t  as_source;        // no predecessor, or joined to a source through empty nodes (:301, :313-318)
    uint8_t  masked;
    uint32_t tb_off;           // bytes, relative to the problem's tb_base: len columns of `stride` bytes
    uint32_t last_off;         // int32 elements, relative to the problem's last_base: last column M | Ic | Ir, first column M | Ic,
                               // `stride` each (only for nodes with keep_last / without chain)
    uint32_t stride;           // band height rounded up to the store granule (4, or the rows per lane)
    uint32_t src_path_off;     // empty nodes between this node and the source it is joined to (relative to pool_base)
    uint32_t src_path_len;
    uint32_t keep_last;        // 1: some traceback may examine this node's last column (it is a predecessor of a non-chain node,
                               // or a candidate end node): store it
    uint32_t chain;            // 1: the only predecessor is the previous non-empty unmasked node in the order, reached directly, and the
                               // band is that node's band carried over: column 0 continues from the registers / traceback bytes
};
struct BSeed {                 // a non-empty, unmasked predecessor reached through path[] of empty nodes, in the LIFO
    uint32_t node;             // order the reference pops them (:305-330, :1226-1262, :1311-1330)
    uint32_t path_off, path_len;
};
struct BStart { uint32_t node; };      // candidate end nodes (non-empty), in the order the reference inserts them (:2442-2556)
struct BProb {
    uint32_t L, n_nodes;
    uint32_t node_base, seed_base, pool_base, start_base, n_starts;
    uint32_t read_off, graph_off;      // codes 0..4; qualities share read_off
    uint32_t Hpad;                     // 64 * rows per lane
    uint64_t tb_base, last_base;
    uint64_t ops_off; uint32_t ops_cap;
    uint32_t graph_len;                // bases of the whole graph
};
struct BResult { int32_t score; int32_t status; uint32_t start; uint32_t n_ops; uint32_t ops_begin; uint32_t pad[3]; };

// rows per lane of a fill launch: 1, 2, 4, 8 in registers; 16, 32 as 2 / 4 blocks of 8 with their state in LDS; 64 ... 512 as 8 ... 64 blocks with
// their state in an HBM slab per wavefront (banded_fill_lane_blocks<0>): bands of up to 32 768 diagonals
constexpr uint32_t B_MAX_ROWS_PER_LANE = 512, B_ROW_CLASSES = 10;
struct BandedParams {
    const BProb*   probs;
    const uint32_t* order;             // problem indices grouped by rows-per-lane class (one fill launch each)
    uint32_t       n;                  // problems
    const BNode*   nodes;
    const BSeed*   seeds;
    const uint32_t* pool;
    const BStart*  starts;
    const uint8_t* reads;              // codes
    const uint8_t* quals;              // raw phred (quality-adjusted contexts only)
    const uint8_t* graph;              // codes
    const int8_t*  mat;                // 25, or 256*25 when quals != nullptr
    int32_t        go, ge;
    uint8_t*       tb;
    int32_t*       scores;            // k-best mode only: every cell's M | Ic | Ir (int32, `stride` each per column), 3 x the traceback layout
    int32_t*       last;
    vgk_op*        ops;               // per-problem slots the traceback writes back to front
    vgk_op*        dense;             // the finished op lists, packed (BResult::ops_begin indexes this)
    unsigned long long* dense_count;  // ops in `dense` so far
    BResult*       results;
};

VGK_HD int32_t bmax(int32_t a, int32_t b) { return a > b ? a : b; }
VGK_HD unsigned long long bump(unsigned long long* counter, unsigned long long n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(counter, n);
#else
    const unsigned long long at = *counter; *counter += n; return at;      // the CPU emulation walks one problem at a time
#endif
}
VGK_HD int32_t bsub(const BandedParams& P, const BProb& pb, uint32_t g, int64_t r) {
    const uint32_t rd = P.reads[pb.read_off + r];
    return P.quals ? P.mat[25u * P.quals[pb.read_off + r] + 5u * g + rd] : P.mat[5u * g + rd];
}

// the R traceback bytes of a lane are contiguous (and R-aligned): dword stores when R >= 4
template <int R> VGK_HD void store_codes(uint8_t* dst, const uint8_t (&codes)[R]) {
    if (R >= 4) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        for (int q = 0; q < R / 4; ++q)
            d[q] = (uint32_t)codes[4 * q] | ((uint32_t)codes[4 * q + 1] << 8) | ((uint32_t)codes[4 * q + 2] << 16) | ((uint32_t)codes[4 * q + 3] << 24);
    } else for (int i = 0; i < R; ++i) dst[i] = codes[i];
}

// Where the lane code reads the problem's read codes, qualities, graph codes and score table from: LDS copies made by
// the kernel's prologue when they fit, the HBM arenas otherwise (and in the CPU emulation).
struct BSrc { const uint8_t* rd; const uint8_t* q; const uint8_t* graph; const int8_t* mat; const uint64_t* rows; };      // rows: FAST only
constexpr uint32_t BMAT_ROWS_AT = 32, BMAT_BYTES = 72;    // plain contexts: the 25 table bytes, then at byte 32 row g as one 64-bit word (byte c = score of read code c)
template <bool QA> VGK_HD int32_t bsub(const BSrc& s, uint32_t g, int32_t r) {
    const uint32_t rd = s.rd[r];
    return QA ? s.mat[25u * s.q[r] + 5u * g + rd] : s.mat[5u * g + rd];
}
// ---- the FAST score path (plain 5 x 5 table, <= 4 rows per lane): no table or read lookups per cell.  Every four columns a lane
// fetches the eight read codes from its first row on (a column later the lane's rows lie one base further) and the wave the four graph
// bases; per column ONE byte permute turns the lane's four read codes into their four scores against the column's base — the table
// row of that base is one 64-bit word fetched with a wave-uniform address (BMAT_ROWS_AT).  On the device the codes come out of padded LDS copies (kernel prologue) by
// aligned dword reads and a byte funnel; the CPU emulation reads the arenas byte by byte.  Rows outside the read get arbitrary
// scores: the lane code masks them.
VGK_HD uint32_t b_perm(uint32_t hi, uint32_t lo, uint32_t sel) {        // byte i of the result = byte sel.byte[i] of {hi, lo} (selectors 0..7)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t t = ((uint64_t)hi << 32) | lo; uint32_t out = 0;
    for (int i = 0; i < 4; ++i) { const uint32_t b = (sel >> (8 * i)) & 0xffu; out |= (b < 8 ? (uint32_t)((t >> (8 * b)) & 0xffu) : 0u) << (8 * i); }
    return out;
#endif
}
VGK_HD uint32_t b_bytes_from(uint64_t win, uint32_t c) {                  // bytes c .. c + 3 of the window, c < 4
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte((uint32_t)(win >> 32), (uint32_t)win, c);
#else
    return (uint32_t)(win >> (8 * c));
#endif
}
VGK_HD uint64_t b_read_window(const BSrc& s, int32_t r0, int32_t L) {    // read codes of rows r0 .. r0 + 7
#if defined(__HIP_DEVICE_COMPILE__)
    const int32_t rc = r0 < -8 ? -8 : r0 > L - 1 ? L - 1 : r0;           // (all of the lane's rows outside the read: any window will do)
    const uint32_t* w = reinterpret_cast<const uint32_t*>(s.rd + (rc & ~3));
    const uint32_t a = w[0], b = w[1], c = w[2], sh = (uint32_t)rc & 3u;
    return ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32) | __builtin_amdgcn_alignbyte(b, a, sh);
#else
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) { const int32_t r = r0 + b; if (r >= 0 && r < L) v |= (uint64_t)s.rd[r] << (8 * b); }
    return v;
#endif
}
VGK_HD uint32_t b_graph_word(const BSrc& s, uint32_t at, uint32_t graph_len) {      // graph codes at .. at + 3 (wave-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t* w = reinterpret_cast<const uint32_t*>(s.graph + (at & ~3u));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_alignbyte(w[1], w[0], at & 3u));
#else
    uint32_t v = 0;
    for (uint32_t b = 0; b < 4; ++b) if (at + b < graph_len) v |= (uint32_t)s.graph[at + b] << (8 * b);
    return v;
#endif
}

// ---- fill: lane code.  XL supplies the cross-lane primitives:
//   int32 up(int32 v)        value of lane+1 (BNEG for the last lane)
//   int32 down(int32 v)      value of lane-1 (BNEG for lane 0)
//   int32 up_sub(int32 old, int32 v, int32 s)     value of lane+1 minus s; the last lane returns `old`
//   int32 down_sub(int32 old, int32 v, int32 s)   value of lane-1 minus s; lane 0 returns `old`
//   int32 scan_excl_keep(int32 old, int32 v)      max over lanes < this one; lane 0 returns `old`
//   int32 scalar(int32 s)    s, wave-uniform, opaque to the optimiser
//   int32 add3(int32 a, int32 b, int32 s)   a + b + s with s wave-uniform, as one instruction
//   int32 in_lanes(int32 s)  s, as a per-lane value (the DPP instructions take no scalar operand)
//   int32 scan_excl(int32 v) max over lanes < this one (BNEG for lane 0)
template <int R, bool QA, bool FAST, class XL>
VGK_HD void banded_fill_lane(const BandedParams& P, const BProb& pb, const BSrc& src, uint32_t lane, XL& xl) {
    static_assert(!FAST || (!QA && R <= 4), "the FAST score path: plain table, at most four rows per lane");
    const int32_t go = P.go, ge = P.ge, L = (int32_t)pb.L;
    const BNode* nodes = P.nodes + pb.node_base;
    int32_t* last = P.last + pb.last_base;
    uint8_t* tb = P.tb + pb.tb_base;
    int32_t M[R], Ic[R], Ir[R];
    for (int i = 0; i < R; ++i) { M[i] = BNEG; Ic[i] = BNEG; Ir[i] = BNEG; }
    // what the lanes without a neighbour (the last one looking at lane + 1, the first one looking at lane - 1) see: -inf, kept in the
    // destination registers of the cross-lane subtractions from column to column (XL::up_sub / down_sub leave those lanes alone)
    int32_t keep_nxM = BNEG, keep_nxIc = BNEG, keep_nxIr = BNEG, keep_upM = BNEG, keep_upIc = BNEG, keep_excl = BNEG;
    const int32_t vgo = xl.in_lanes(go), vge = xl.in_lanes(ge);
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        const BNode nd = nodes[v];
        if (nd.masked || nd.len == 0) continue;
        const int32_t H = nd.bot - nd.top + 1;
        const uint8_t* seq = src.graph + nd.seq_off;
        uint8_t* tbn = tb + nd.tb_off;
        const int32_t k0 = (int32_t)lane * R;
        int32_t kge[R];                                     // (row of the lane's i-th band row in column 0) * ge
        for (int i = 0; i < R; ++i) kge[i] = (k0 + i + nd.top) * ge;

        // the row gaps of a column and its traceback bytes, given the column's M / Ic and the lead-gap seeds ir0 of Ir:
        //   Ir(r) = max(ir0(r), max_{r'<r} Y(r') - (r-1)*ge),  Y(r') = max(max(M,Ic)(r') - go, ir0(r') - ge) + r'*ge
        // EDGE = false: a column all of whose band rows lie strictly inside the read (0 < r < L): no lead gaps (ir0 = -inf everywhere),
        // no row tests — the node decides that once for all its columns.
        auto finish_column = [&](auto edge, int32_t j, const int32_t (&nM)[R], const int32_t (&nIc)[R], const int32_t (&ir0)[R], const uint32_t (&code_mc)[R]) {
            constexpr bool EDGE = decltype(edge)::value;
            // (-inf is any value below BNEG / 2: nothing here is clamped back to BNEG — a dead cell drifts by a gap or substitution score per
            // column, far from the threshold for any graph the 16-bit run lengths admit — so the first row of a lane and the first lane need no max)
            const int32_t jge = xl.scalar(j * ge);             // (kept a scalar: as a per-lane running sum it costs two VALU per column)
            int32_t run = BNEG, pre[R];
            for (int i = 0; i < R; ++i) {
                pre[i] = run;
                const int32_t open = bmax(nM[i], nIc[i]) - go;
                const int32_t y = (EDGE ? bmax(open, ir0[i] - ge) : open) + kge[i] + jge;      // + r * ge
                run = i == 0 ? y : bmax(run, y);
            }
            const int32_t excl = keep_excl = xl.scan_excl_keep(keep_excl, run);
            // the cells above, a gap opening away: the previous lane's last row, then the lane's own rows
            int32_t upM = keep_upM = xl.down_sub(keep_upM, nM[R - 1], vgo), upIc = keep_upIc = xl.down_sub(keep_upIc, nIc[R - 1], vgo);
            uint8_t codes[R];
            for (int i = 0; i < R; ++i) {
                const int32_t k = k0 + i, r = k + nd.top + j;
                const bool valid = EDGE ? (k < H && r >= 0 && r < L) : k < H;
                const int32_t from_above = xl.add3(i == 0 ? excl : bmax(excl, pre[i]), ge - kge[i], -jge);      // - (r - 1) * ge
                int32_t ir = EDGE ? bmax(ir0[i], from_above) : from_above;
                const uint32_t cr = ir == upM ? BM : ir == upIc ? BIC : BIR;
                if (!valid) ir = BNEG;
                if (i + 1 < R) { upM = nM[i] - go; upIc = nIc[i] - go; }
                M[i] = nM[i]; Ic[i] = nIc[i]; Ir[i] = ir;
                codes[i] = (uint8_t)(code_mc[i] | (cr << 2));
            }
            if (k0 < (int32_t)nd.stride) store_codes<R>(tbn + (size_t)j * nd.stride + (uint32_t)k0, codes);
            if (P.scores) {         // the alternate tracebacks need score differences, not just sources (AltTracebackStack)
                int32_t* sc = P.scores + 3 * (pb.tb_base + nd.tb_off + (size_t)j * nd.stride);
                for (int i = 0; i < R; ++i) if (k0 + i < (int32_t)nd.stride) { sc[k0 + i] = M[i]; sc[nd.stride + k0 + i] = Ic[i]; sc[2 * nd.stride + k0 + i] = Ir[i]; }
            }
        };
        // a column whose left neighbours are the registers: columns 1.. of a node (:492-590), and column 0 of a node whose
        // only predecessor is the node this wave has just finished, with the band carried straight over (BNode::chain)
        auto column = [&](auto edge, int32_t j, const int32_t (&msv)[R]) {  // msv: the rows' substitution scores against the column's base
            constexpr bool EDGE = decltype(edge)::value;
            // row k+1 of the previous column, a gap opening / extension away (the next lane's first row; the lane's own rows below)
            const int32_t nxM = keep_nxM = xl.up_sub(keep_nxM, M[0], vgo), nxIc = keep_nxIc = xl.up_sub(keep_nxIc, Ic[0], vge), nxIr = keep_nxIr = xl.up_sub(keep_nxIr, Ir[0], vgo);
            int32_t nM[R], nIc[R], ir0[R];
            uint32_t code_mc[R];
            const int32_t lead_m = -go - (nd.cum + j - 1) * ge;                              // implied lead gap along the top edge (:507-526, :352-366)
            const int32_t lead_ir = nd.top + j < 0 ? -2 * go - (nd.cum + j) * ge : BNEG;
            for (int i = 0; i < R; ++i) {
                const int32_t k = k0 + i, r = k + nd.top + j;
                const bool valid = EDGE ? (k < H && r >= 0 && r < L) : k < H;
                const int32_t oM = i + 1 < R ? M[i + 1 < R ? i + 1 : 0] - go : nxM, oIc = i + 1 < R ? Ic[i + 1 < R ? i + 1 : 0] - ge : nxIc,
                              oIr = i + 1 < R ? Ir[i + 1 < R ? i + 1 : 0] - go : nxIr;
                const int32_t ms = msv[i];
                const int32_t b3 = bmax(bmax(M[i], Ic[i]), Ir[i]);
                const uint32_t cm = b3 == M[i] ? BM : b3 == Ic[i] ? BIC : BIR;
                const int32_t icv = bmax(bmax(oM, oIr), oIc);
                const uint32_t cc = icv == oM ? BM : icv == oIc ? BIC : BIR;
                const bool top_row = EDGE && r == 0;
                // (inside the read the rows behind the band's last one need no mask here: they hold -inf — the node's first column and the
                // Ir mask of finish_column see to that — and -inf plus a score stays -inf)
                nM[i] = !EDGE ? ms + b3 : valid ? (top_row ? ms + lead_m : ms + b3) : BNEG;
                nIc[i] = !EDGE || valid ? icv : BNEG;
                ir0[i] = EDGE && valid && top_row ? lead_ir : BNEG;
                code_mc[i] = cm | (cc << 4);
            }
            finish_column(edge, j, nM, nIc, ir0, code_mc);
        };

        // columns [j, len) of the node.  Scores looked up cell by cell (quality-adjusted tables, tall lanes, the emulation of those), or
        // the FAST path above.  The loop exists once per kind of node (interior / touching an edge of the read): no per-column choice.
        auto columns_of = [&](auto edge, int32_t j, const int32_t jend) {
            if (!FAST) {
                for (; j < jend; ++j) {
                    const uint32_t g = seq[j];
                    int32_t msv[R];
                    for (int i = 0; i < R; ++i) {
                        const int32_t r = k0 + i + nd.top + j;
                        const int32_t rc = r < 0 ? 0 : r >= L ? L - 1 : r;       // branch-free: out-of-matrix rows read a clamped read base and are masked
                        msv[i] = bsub<QA>(src, g, rc);
                    }
                    column(edge, j, msv);
                }
                return;
            }
            uint64_t win = 0; uint32_t g4 = 0;
            for (uint32_t c = 0; j < jend; ++j, c = (c + 1) & 3u) {       // (not unrolled: four columns in flight cost 25 VGPRs and three waves per SIMD)
                if (c == 0) { win = b_read_window(src, k0 + nd.top + j, L); g4 = b_graph_word(src, nd.seq_off + (uint32_t)j, pb.graph_len); }
                const uint32_t g = (g4 >> (8 * c)) & 0xffu;
                const uint64_t row = src.rows[g];
                const uint32_t sc4 = b_perm((uint32_t)(row >> 32), (uint32_t)row, b_bytes_from(win, c));
                int32_t msv[R];
                for (int i = 0; i < R; ++i) msv[i] = (int32_t)(int8_t)(uint8_t)(sc4 >> (8 * i));
                column(edge, j, msv);
            }
        };
        // columns [j, len): those whose band rows all lie strictly inside the read (0 < top + j, bot + j < L) run without the row tests
        auto columns_from = [&](int32_t j) {
            const int32_t a = j > 1 - nd.top ? j : 1 - nd.top, b = nd.len < L - nd.bot ? nd.len : L - nd.bot;
            if (a >= b) { columns_of(std::true_type(), j, nd.len); return; }
            if (j < a) columns_of(std::true_type(), j, a);
            columns_of(std::false_type(), a, b);
            if (b < nd.len) columns_of(std::true_type(), b, nd.len);
        };

        if (nd.chain) columns_from(0);
        else {
            // ---- column 0: gather from the predecessors' last columns (:333-430) and the implied lead gaps (:433-476)
            const uint32_t g = seq[0];
            const int32_t hi0 = nd.bot >= L ? L - 1 : nd.bot;
            int32_t nM[R], nIc[R], ir0[R];
            uint32_t code_mc[R];
            for (int i = 0; i < R; ++i) {
                const int32_t k = k0 + i, r = k + nd.top;
                const bool valid = k < H && r >= 0 && r < L;
                int32_t m = BNEG, ic = BNEG, ir = BNEG;
                if (valid) {
                    const int32_t ms = bsub<QA>(src, g, r);
                    for (uint32_t si = 0; si < nd.n_seeds; ++si) {
                        const BNode sd = nodes[P.seeds[pb.seed_base + nd.seed_off + si].node];
                        const int32_t snt = sd.top + sd.len, snb = sd.bot + sd.len;
                        const int32_t lo = snt < 0 ? 0 : snt, hi = snb >= L ? L - 1 : snb;
                        if (r < lo || r > hi) continue;
                        const int32_t ext = sd.cum + sd.len;
                        const int32_t* sl = last + sd.last_off;
                        if (r == lo && snt <= 0) {
                            m = bmax(m, ms - go - (ext - 1) * ge);
                            if (snt < 0) ir = bmax(ir, -2 * go - ext * ge);
                        } else {
                            const int32_t ks = r - snt;
                            m = bmax(m, ms + bmax(bmax(sl[ks], sl[2 * sd.stride + ks]), sl[sd.stride + ks]));
                        }
                        if (r <= snb - 1) {
                            const int32_t ks = r - snt + 1;
                            ic = bmax(ic, bmax(bmax(sl[ks] - go, sl[2 * sd.stride + ks] - go), sl[sd.stride + ks] - ge));
                        }
                    }
                    if (nd.as_source) {
                        if (r == 0) { m = QA ? bmax(m, ms) : ms; ir = bmax(ir, -2 * go); ic = bmax(ic, -2 * go); }
                        else { m = bmax(m, ms - go - (r - 1) * ge); ic = bmax(ic, -2 * go - r * ge); }
                        if (r == hi0) ic = BNEG;
                    }
                }
                nM[i] = m; nIc[i] = ic; ir0[i] = ir; code_mc[i] = 0;
            }
            finish_column(std::true_type(), 0, nM, nIc, ir0, code_mc);
            // the traceback re-examines the predecessors from this column (traceback_over_edge): keep its M and Ic
            int32_t* nf = last + nd.last_off + 3 * nd.stride;
            for (int i = 0; i < R; ++i) if (k0 + i < (int32_t)nd.stride) { nf[k0 + i] = M[i]; nf[nd.stride + k0 + i] = Ic[i]; }
        }
        if (!nd.chain) columns_from(1);
        // ---- keep the last column for the successors and the traceback
        if (nd.keep_last) {
            int32_t* nl = last + nd.last_off;
            for (int i = 0; i < R; ++i) {
                const uint32_t k = lane * R + i;
                if (k < nd.stride) { nl[k] = M[i]; nl[nd.stride + k] = Ic[i]; nl[2 * nd.stride + k] = Ir[i]; }
            }
        }
        xl.fence();        // successors read these through memory
    }
}

// ---- fill, wide bands: the band as B BLOCKS of 64 lanes x 8 rows, one after the other per column ------------------------------------------------
// A band of more than 512 diagonals used to run as 16 or 32 rows per lane: 399-512 VGPRs and up to 1.6 KB of scratch per lane (the row arrays of
// banded_fill_lane live in registers only up to 8 rows).  Here a lane keeps 8 rows of ONE block in registers and the M / Ic / Ir of its rows in
// the other blocks in `st` (LDS on the device, lane-interleaved: st[e * st_stride]); per column the blocks run top to bottom: block b's last lane
// takes "the next row of the previous column" from block b + 1's first row, which that block has not overwritten yet (kept wave-uniform in
// `first`), and block b's first lane takes the row above — the running maximum of the row-gap scan and the gap-opening sources — from what
// block b - 1 has just computed for this column (wave-uniform carries).  Same cells, same codes, same last columns as banded_fill_lane<16 | 32>.
// XL additionally supplies first_lane(v) / last_lane(v): the value of the first / last lane, to every lane.
// B = 0: any number of blocks (`nb`, known at run time: bands of more than 2048 diagonals, round 6) — the blocks' first rows then live in `fst`
// (nb + 1 triples behind each other, element e at fst[e * st_stride]: wave-uniform values every lane writes alike and reads back itself), and both
// `st` and `fst` are the wavefront's stretch of an HBM slab (nb x 6 KB: too much for LDS beyond 8 blocks); the blocks of a column still run top to
// bottom through registers.  The reference aligns any band its max_cells admits (src/banded_global_aligner.cpp:2019-2043); so does this.
template <int B, bool QA, class XL>
VGK_HD void banded_fill_lane_blocks(const BandedParams& P, const BProb& pb, const BSrc& src, uint32_t lane, XL& xl, int32_t* st, uint32_t st_stride,
                                    uint32_t nb = (uint32_t)B, int32_t* fst = nullptr) {
    constexpr int R = 8;
    const int NB = B ? B : (int)nb;
    const int32_t go = P.go, ge = P.ge, L = (int32_t)pb.L;
    const uint32_t W = xl.width();
    const BNode* nodes = P.nodes + pb.node_base;
    int32_t* last = P.last + pb.last_base;
    uint8_t* tb = P.tb + pb.tb_base;
    auto S = [&](int b, int m, int i) -> int32_t& { return st[(uint32_t)((b * 3 + m) * R + i) * st_stride]; };
    for (int b = 0; b < NB; ++b) for (int m = 0; m < 3; ++m) for (int i = 0; i < R; ++i) S(b, m, i) = BNEG;
    // block b's first row (M, Ic, Ir) of the column computed last: wave-uniform values, chosen by selects (the block loops are NOT unrolled — four
    // copies of a column's code are 300 VGPRs — and a register array indexed by a loop counter would live in scratch)
    int32_t first[B ? B : 1][3];
    if constexpr (B != 0) { for (int b = 0; b < B; ++b) first[b][0] = first[b][1] = first[b][2] = BNEG; }
    else { for (int e = 0; e < 3 * (NB + 1); ++e) fst[(uint32_t)e * st_stride] = BNEG; first[0][0] = first[0][1] = first[0][2] = BNEG; }
    auto first_get = [&](int b, int m) -> int32_t {
        if constexpr (B != 0) { int32_t v = BNEG; for (int q = 0; q < B; ++q) v = q == b ? first[q][m] : v; return v; }
        else return fst[(uint32_t)(b * 3 + m) * st_stride];                  // (entry NB stays BNEG: past the last block)
    };
    auto first_set = [&](int b, int m, int32_t v) {
        if constexpr (B != 0) { for (int q = 0; q < B; ++q) first[q][m] = q == b ? v : first[q][m]; }
        else fst[(uint32_t)(b * 3 + m) * st_stride] = v;
    };
    const bool last_l = lane + 1 == W, first_l = lane == 0;
    for (uint32_t v = 0; v < pb.n_nodes; ++v) {
        const BNode nd = nodes[v];
        if (nd.masked || nd.len == 0) continue;
        const int32_t H = nd.bot - nd.top + 1;
        const uint8_t* seq = src.graph + nd.seq_off;
        uint8_t* tbn = tb + nd.tb_off;
        // the row gaps of block b's column and its traceback bytes (finish_column of banded_fill_lane, every column in its EDGE form); carries in
        // and out: the scan's running maximum over the rows above, and M / Ic of the row right above a gap opening away
        auto finish = [&](int b, int32_t j, const int32_t (&nM)[R], const int32_t (&nIc)[R], const int32_t (&ir0)[R], const uint32_t (&code_mc)[R],
                          int32_t& carry_excl, int32_t& carry_upM, int32_t& carry_upIc) {
            const int32_t k0 = (int32_t)((uint32_t)b * W + lane) * R, jge = j * ge;
            int32_t run = BNEG, pre[R], kge[R];
            for (int i = 0; i < R; ++i) {
                kge[i] = (k0 + i + nd.top) * ge;
                pre[i] = run;
                const int32_t y = bmax(bmax(nM[i], nIc[i]) - go, ir0[i] - ge) + kge[i] + jge;
                run = i == 0 ? y : bmax(run, y);
            }
            const int32_t excl = bmax(xl.scan_excl(run), carry_excl);
            int32_t upM = xl.down(nM[R - 1]), upIc = xl.down(nIc[R - 1]);
            upM = first_l ? carry_upM : upM - go; upIc = first_l ? carry_upIc : upIc - go;
            uint8_t codes[R]; int32_t nIr[R];
            for (int i = 0; i < R; ++i) {
                const int32_t k = k0 + i, r = k + nd.top + j;
                const bool valid = k < H && r >= 0 && r < L;
                const int32_t from_above = (i == 0 ? excl : bmax(excl, pre[i])) + ge - kge[i] - jge;
                int32_t ir = bmax(ir0[i], from_above);
                const uint32_t cr = ir == upM ? BM : ir == upIc ? BIC : BIR;
                if (!valid) ir = BNEG;
                upM = nM[i] - go; upIc = nIc[i] - go;
                nIr[i] = ir;
                codes[i] = (uint8_t)(code_mc[i] | (cr << 2));
            }
            for (int i = 0; i < R; ++i) { S(b, 0, i) = nM[i]; S(b, 1, i) = nIc[i]; S(b, 2, i) = nIr[i]; }
            if (k0 < (int32_t)nd.stride) store_codes<R>(tbn + (size_t)j * nd.stride + (uint32_t)k0, codes);
            if (P.scores) {
                int32_t* sc = P.scores + 3 * (pb.tb_base + nd.tb_off + (size_t)j * nd.stride);
                for (int i = 0; i < R; ++i) if (k0 + i < (int32_t)nd.stride) { sc[k0 + i] = nM[i]; sc[nd.stride + k0 + i] = nIc[i]; sc[2 * nd.stride + k0 + i] = nIr[i]; }
            }
            carry_excl = xl.last_lane(bmax(excl, run));
            carry_upM = xl.last_lane(nM[R - 1]) - go; carry_upIc = xl.last_lane(nIc[R - 1]) - go;
            first_set(b, 0, xl.first_lane(nM[0])); first_set(b, 1, xl.first_lane(nIc[0])); first_set(b, 2, xl.first_lane(nIr[0]));
        };
        // a column whose left neighbour is the stored state (columns 1.. of a node; column 0 of a chained node)
        auto column = [&](int32_t j) {
            const uint32_t g = seq[j];
            const int32_t lead_m = -go - (nd.cum + j - 1) * ge;
            const int32_t lead_ir = nd.top + j < 0 ? -2 * go - (nd.cum + j) * ge : BNEG;
            int32_t carry_excl = BNEG, carry_upM = BNEG, carry_upIc = BNEG;
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                const int32_t k0 = (int32_t)((uint32_t)b * W + lane) * R;
                int32_t M[R], Ic[R], Ir[R];
                for (int i = 0; i < R; ++i) { M[i] = S(b, 0, i); Ic[i] = S(b, 1, i); Ir[i] = S(b, 2, i); }
                // row k + 1 of the previous column for the lane's last row: the next lane's first row, or the next block's
                int32_t nxM = xl.up(M[0]), nxIc = xl.up(Ic[0]), nxIr = xl.up(Ir[0]);
                if (last_l) { nxM = first_get(b + 1, 0); nxIc = first_get(b + 1, 1); nxIr = first_get(b + 1, 2); }      // (past the last block: BNEG)
                int32_t nM[R], nIc[R], ir0[R]; uint32_t code_mc[R];
                for (int i = 0; i < R; ++i) {
                    const int32_t k = k0 + i, r = k + nd.top + j;
                    const bool valid = k < H && r >= 0 && r < L;
                    const int32_t oM = (i + 1 < R ? M[i + 1 < R ? i + 1 : 0] : nxM) - go, oIc = (i + 1 < R ? Ic[i + 1 < R ? i + 1 : 0] : nxIc) - ge,
                                  oIr = (i + 1 < R ? Ir[i + 1 < R ? i + 1 : 0] : nxIr) - go;
                    const int32_t rc = r < 0 ? 0 : r >= L ? L - 1 : r;
                    const int32_t ms = bsub<QA>(src, g, rc);
                    const int32_t b3 = bmax(bmax(M[i], Ic[i]), Ir[i]);
                    const uint32_t cm = b3 == M[i] ? BM : b3 == Ic[i] ? BIC : BIR;
                    const int32_t icv = bmax(bmax(oM, oIr), oIc);
                    const uint32_t cc = icv == oM ? BM : icv == oIc ? BIC : BIR;
                    const bool top_row = r == 0;
                    nM[i] = valid ? (top_row ? ms + lead_m : ms + b3) : BNEG;
                    nIc[i] = valid ? icv : BNEG;
                    ir0[i] = valid && top_row ? lead_ir : BNEG;
                    code_mc[i] = cm | (cc << 4);
                }
                finish(b, j, nM, nIc, ir0, code_mc, carry_excl, carry_upM, carry_upIc);
            }
        };
        int32_t j0 = 0;
        if (!nd.chain) {
            // ---- column 0: gather from the predecessors' last columns and the implied lead gaps (as banded_fill_lane)
            const uint32_t g = seq[0];
            const int32_t hi0 = nd.bot >= L ? L - 1 : nd.bot;
            int32_t carry_excl = BNEG, carry_upM = BNEG, carry_upIc = BNEG;
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                const int32_t k0 = (int32_t)((uint32_t)b * W + lane) * R;
                int32_t nM[R], nIc[R], ir0[R]; uint32_t code_mc[R];
                for (int i = 0; i < R; ++i) {
                    const int32_t k = k0 + i, r = k + nd.top;
                    const bool valid = k < H && r >= 0 && r < L;
                    int32_t m = BNEG, ic = BNEG, ir = BNEG;
                    if (valid) {
                        const int32_t ms = bsub<QA>(src, g, r);
                        for (uint32_t si = 0; si < nd.n_seeds; ++si) {
                            const BNode sd = nodes[P.seeds[pb.seed_base + nd.seed_off + si].node];
                            const int32_t snt = sd.top + sd.len, snb = sd.bot + sd.len;
                            const int32_t lo = snt < 0 ? 0 : snt, hi = snb >= L ? L - 1 : snb;
                            if (r < lo || r > hi) continue;
                            const int32_t ext = sd.cum + sd.len;
                            const int32_t* sl = last + sd.last_off;
                            if (r == lo && snt <= 0) {
                                m = bmax(m, ms - go - (ext - 1) * ge);
                                if (snt < 0) ir = bmax(ir, -2 * go - ext * ge);
                            } else {
                                const int32_t ks = r - snt;
                                m = bmax(m, ms + bmax(bmax(sl[ks], sl[2 * sd.stride + ks]), sl[sd.stride + ks]));
                            }
                            if (r <= snb - 1) {
                                const int32_t ks = r - snt + 1;
                                ic = bmax(ic, bmax(bmax(sl[ks] - go, sl[2 * sd.stride + ks] - go), sl[sd.stride + ks] - ge));
                            }
                        }
                        if (nd.as_source) {
                            if (r == 0) { m = QA ? bmax(m, ms) : ms; ir = bmax(ir, -2 * go); ic = bmax(ic, -2 * go); }
                            else { m = bmax(m, ms - go - (r - 1) * ge); ic = bmax(ic, -2 * go - r * ge); }
                            if (r == hi0) ic = BNEG;
                        }
                    }
                    nM[i] = m; nIc[i] = ic; ir0[i] = ir; code_mc[i] = 0;
                }
                finish(b, 0, nM, nIc, ir0, code_mc, carry_excl, carry_upM, carry_upIc);
                int32_t* nf = last + nd.last_off + 3 * nd.stride;          // the traceback re-examines the predecessors from this column: keep its M and Ic
                for (int i = 0; i < R; ++i) if (k0 + i < (int32_t)nd.stride) { nf[k0 + i] = S(b, 0, i); nf[nd.stride + k0 + i] = S(b, 1, i); }
            }
            j0 = 1;
        }
        for (int32_t j = j0; j < nd.len; ++j) column(j);
        if (nd.keep_last) {
            int32_t* nl = last + nd.last_off;
            for (int b = 0; b < NB; ++b) for (int i = 0; i < R; ++i) {
                const uint32_t k = ((uint32_t)b * W + lane) * R + (uint32_t)i;
                if (k < nd.stride) { nl[k] = S(b, 0, i); nl[nd.stride + k] = S(b, 1, i); nl[2 * nd.stride + k] = S(b, 2, i); }
            }
        }
        xl.fence();        // successors read these through memory
    }
}

// ---- traceback: one thread per problem (BAMatrix::traceback :756-1126, traceback_over_edge :1129-1780, BABuilder :44-205)
struct BWalker {
    vgk_op* end; vgk_op* cur; vgk_op* floor;      // finished runs are written back to front
    uint32_t node, op, len; bool open;            // the run being grown lives in registers
    bool overflow;
};
VGK_HD void bflush(BWalker& w) {
    if (!w.open) return;
    if (w.cur == w.floor) { w.overflow = true; return; }
    --w.cur;
    w.cur->node = w.node; w.cur->op = (uint8_t)w.op; w.cur->len = (uint16_t)w.len; w.cur->pad = 0;
}
VGK_HD void bemit(BWalker& w, const BNode* nodes, uint32_t node, uint32_t op, uint32_t inc) {
    if (w.open && w.node == node) {
        if (w.op == op) { w.len += inc; return; }
        if (w.len == 0 && nodes[node].len == 0) { w.op = op; w.len = inc; return; }   // an empty node's zero edit is replaced (:69-72)
    }
    bflush(w);
    w.node = node; w.op = op; w.len = inc; w.open = true;
}
VGK_HD uint32_t bop(uint32_t mat) { return mat == BM ? VGK_OP_M : mat == BIR ? VGK_OP_I : VGK_OP_D; }
// source state of a transition out of a predecessor's last column, in the reference's order
VGK_HD int bpick(const int32_t* sl, uint32_t Hpad, int32_t ks, int32_t cur, int32_t dm, int32_t dc, int32_t dr) {
    if (cur == sl[ks] + dm) return BM;
    if (blive(sl[Hpad + ks]) && cur == sl[Hpad + ks] + dc) return BIC;
    if (blive(sl[2 * Hpad + ks]) && cur == sl[2 * Hpad + ks] + dr) return BIR;
    return -1;
}

VGK_HD void banded_walk_one(const BandedParams& P, uint32_t pi) {
    const BProb pb = P.probs[pi];
    BResult& res = P.results[pi];
    const BNode* nodes = P.nodes + pb.node_base;
    const int32_t* last = P.last + pb.last_base;
    const uint8_t* tb = P.tb + pb.tb_base;
    const int32_t go = P.go, ge = P.ge, L = (int32_t)pb.L;
    // where the traceback starts (:2442-2556): first strictly better candidate wins, match before insert-row before insert-col
    bool have = false; int32_t best = 0; uint32_t bnode = 0, bmat = BM, bstart = 0;
    for (uint32_t c = 0; c < pb.n_starts; ++c) {
        const uint32_t u = P.starts[pb.start_base + c].node;
        const BNode n = nodes[u];
        const int32_t k = (L - 1) - (n.len - 1) - n.top;
        if (k < 0 || k > n.bot - n.top) continue;
        const int32_t* nl = last + n.last_off;
        const int32_t cand[3] = { nl[k], nl[2 * n.stride + k], nl[n.stride + k] };
        const uint32_t cmat[3] = { BM, BIR, BIC };
        for (int q = 0; q < 3; ++q) if (blive(cand[q]) && (!have || cand[q] > best)) { have = true; best = cand[q]; bnode = u; bmat = cmat[q]; bstart = c; }
    }
    res.start = bstart; res.n_ops = 0; res.ops_begin = 0;
    if (!have) { res.score = 0; res.status = VGK_ENOBAND; return; }
    res.score = best;
    BWalker w; w.end = P.ops + pb.ops_off + pb.ops_cap; w.cur = w.end; w.floor = P.ops + pb.ops_off; w.overflow = false;
    w.open = false; w.node = 0; w.op = 0; w.len = 0;
    uint32_t node = bnode, mat = bmat; int32_t r = L - 1, j = nodes[node].len - 1;
    bool lead = false; int status = VGK_OK;
    for (;;) {
        const BNode n = nodes[node];
        const uint8_t* seq = P.graph + pb.graph_off + n.seq_off;
        const uint8_t* tbn = tb + n.tb_off;
        // a run of matches walks up one band row: its bytes sit one stride apart, so four of them are fetched at once
        int32_t ck = -1, cj = -1; uint32_t c4[4] = {0, 0, 0, 0};
        auto code_at = [&](int32_t jj, int32_t kk) -> uint32_t {
            if (kk != ck || jj > cj || jj <= cj - 4) {
                ck = kk; cj = jj;
                for (int q = 0; q < 4; ++q) c4[q] = jj - q >= 0 ? tbn[(size_t)(jj - q) * n.stride + kk] : 0u;
            }
            return c4[cj - jj];
        };
        while ((j > 0 || mat == BIR) && !lead) {
            bemit(w, nodes, node, bop(mat), 1);
            const uint32_t code = code_at(j, r - j - n.top);
            if (mat == BM) {
                if (r == 0) { mat = BIC; --j; r = -1; lead = true; break; }
                mat = code & 3u; --r; --j;
            } else if (mat == BIR) {
                if (r == 0) { lead = true; r = -1; break; }
                mat = (code >> 2) & 3u; --r;
            } else {
                mat = (code >> 4) & 3u; --j;
            }
        }
        if (lead) { mat = BIC; while (j > 0) { bemit(w, nodes, node, VGK_OP_D, 1); --j; } }
        const BSeed* seeds = P.seeds + pb.seed_base + n.seed_off;
        if (n.chain && !lead) {       // the only predecessor continues this band: column 0 reads its traceback byte like any column
            bemit(w, nodes, node, bop(mat), 1);
            const uint32_t code = code_at(0, r - n.top);
            if (mat == BM) {
                if (r == 0) { mat = BIC; lead = true; r = -1; }
                else { mat = code & 3u; --r; }
            } else mat = (code >> 4) & 3u;
            node = seeds[0].node; j = nodes[node].len - 1;
            continue;
        }
        const uint32_t* pool = P.pool + pb.pool_base;
        int found = -1; uint32_t fmat = BM; bool flead = lead; int32_t ms = 0;
        if (lead) {
            bemit(w, nodes, node, VGK_OP_D, 1);
            for (uint32_t si = 0; si < n.n_seeds && found < 0; ++si) {
                const BNode s = nodes[seeds[si].node];
                if ((int64_t)ge * (s.cum + s.len - n.cum) == 0) found = (int)si;
            }
            if (found < 0) {
                if (!n.as_source) { status = VGK_EINVAL; break; }
                for (uint32_t q = 0; q < n.src_path_len; ++q) bemit(w, nodes, pool[n.src_path_off + q], VGK_OP_D, 0);
                break;
            }
        } else {
            bemit(w, nodes, node, bop(mat), 1);
            ms = mat == BM ? bsub(P, pb, seq[0], r) : 0;
            // the value of the cell we stand on: the fill kept column 0 of M and Ic for exactly this
            const int32_t cur = (last + n.last_off)[(mat == BM ? 3 : 4) * n.stride + (r - n.top)];
            for (uint32_t si = 0; si < n.n_seeds && found < 0; ++si) {
                const BNode s = nodes[seeds[si].node];
                const int32_t snt = s.top + s.len, snb = s.bot + s.len;
                if (r > snb - (mat == BIC ? 1 : 0) || r < snt) continue;
                const int32_t* sl = last + s.last_off;
                if (mat == BM) {
                    if (r == 0) { if (cur == -go - (s.cum + s.len - 1) * ge + ms) { found = (int)si; fmat = BIC; flead = true; } continue; }
                    const int src = bpick(sl, s.stride, r - snt, cur, ms, ms, ms);
                    if (src >= 0) { found = (int)si; fmat = (uint32_t)src; }
                } else {
                    const int src = bpick(sl, s.stride, r - snt + 1, cur, -go, -ge, -go);
                    if (src >= 0) { found = (int)si; fmat = (uint32_t)src; }
                }
            }
            if (found < 0) {
                if (!n.as_source) { status = VGK_EINVAL; break; }
                int32_t ins;
                if (mat == BM) { if (cur != (r > 0 ? -go - (r - 1) * ge : 0) + ms) { status = VGK_EINVAL; break; } ins = r; }
                else           { if (cur != -go - r * ge - go) { status = VGK_EINVAL; break; } ins = r + 1; }
                for (uint32_t q = 0; q < n.src_path_len; ++q) bemit(w, nodes, pool[n.src_path_off + q], VGK_OP_D, 0);
                const uint32_t end_node = n.src_path_len ? pool[n.src_path_off + n.src_path_len - 1] : node;
                for (int32_t q = 0; q < ins; ++q) bemit(w, nodes, end_node, VGK_OP_I, 1);
                break;
            }
        }
        const BSeed sr = seeds[found];
        for (uint32_t q = 0; q < sr.path_len; ++q) bemit(w, nodes, pool[sr.path_off + q], bop(mat), 0);
        if (!lead) { if (mat == BM) --r; mat = fmat; lead = flead; }
        node = sr.node; j = nodes[node].len - 1;
    }
    bflush(w);
    if (w.overflow && status == VGK_OK) status = VGK_EOPS;
    res.status = status;
    // pack the finished list behind the others: only what was written travels back to the host
    const uint32_t n_ops = status == VGK_OK ? (uint32_t)(w.end - w.cur) : 0u;
    const unsigned long long at = n_ops ? bump(P.dense_count, n_ops) : 0ull;
    for (uint32_t q = 0; q < n_ops; ++q) P.dense[at + q] = w.cur[q];
    res.n_ops = n_ops;
    res.ops_begin = (uint32_t)at;
}

}  // namespace vgk
