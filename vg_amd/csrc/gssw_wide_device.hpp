// gssw_wide_device.hpp — the graph Smith-Waterman / X-drop engine for problems OUTSIDE the packed kernel's range: reads of more than
// 1024 rows (dozeu has no such limit — its guard is the caller's cell budget, reference src/minimizer_mapper.cpp:5694-5701 and
// src/minimizer_mapper_from_chains.cpp:3787-3813; the reference's own "can align a long tail" test is a 4.4 kbp tail,
// src/unittest/minimizer_mapper.cpp:682-709) and scorings whose reachable scores leave 11 bits (gssw_device.hpp keeps two reads per
// register in unsigned 16-bit halves with the score x 32 in the end-cell key).
//
// Same recurrences, same tie rules, same outputs as gssw_device.hpp (LOCAL, PINNED and XDROP_PINNED; Aligner::align_internal,
// src/aligner.cpp:396-435, 537-557; DozeuInterface::align_pinned, src/dozeu_interface.cpp:210-307) — in plain signed 32-bit cells, one
// read per lane register, so nothing saturates before gssw's / dozeu's own int16 limit (reported as VGK_EOVERFLOW, like the oracles).
//
// MI355X mapping:
//   * one WORKGROUP of four wavefronts per problem: its 256 lanes form one skewed wavefront, lane l owning K consecutive read rows
//     (K = 8 or 16); at step t lane l computes graph column t - l.  H, F and the column byte of the row above come from lane l - 1
//     through one DPP wave_shr:1 each inside a wavefront and through three LDS words (double-buffered by step parity) from the last
//     lane of the wavefront before; one s_barrier per step keeps the four wavefronts in step.
//   * a read of more than 256 K rows runs in STRIPS of 256 K rows, one after the other: the last lane of a strip leaves H and F of its
//     last row per column in HBM (`carry`, 8 B per column), the first lane of the next strip takes them as its row above.
//   * node boundaries as in the packed kernel: chain links carry on in registers, every other boundary goes through `scratch`
//     (per saved node and read row: H of its last column and E for the column after it), which the traceback also reads.
//   * per cell a 4-bit traceback code, K / 8 dwords per (strip, step, lane): every wavefront store is one contiguous burst.
//   * the traceback is one lane per problem over the codes, the H / E / F state machine of gssw_device.hpp's walk_one.
//
// Plain C++ so that tests/emu can step the identical lane code on the CPU.
#pragma once
#include <stdint.h>
#include "../../include/vgk.h"
#include "gssw_device.hpp"

namespace vgk {

constexpr int32_t  WNEG = -(1 << 28);          // "unreachable" (X-drop has no zero floor)
constexpr uint32_t WIDE_LANES = 256;           // lanes of one problem's skewed wavefront (four wavefronts)

struct WPair { int32_t h, x; };                // scratch: H, E for the next column; carry: H, F of a strip's last row

struct WideProb {
    uint32_t col_off, R;                       // column-info stream (gssw_device.hpp CI_*), graph columns
    uint32_t L;                                // read rows (X-drop: read length + 1, row 0 = nothing consumed yet)
    uint32_t prof_off;                         // first per-row profile word: 4 bytes = score + bias against ref A, C, G, T, bonuses folded in
    uint32_t node_off, n_nodes;                // NodeRec (gssw_device.hpp)
    uint32_t flags;                            // VGK_GSSW_*
    uint32_t ops_off, ops_cap;
    uint32_t max_gap;                          // X-drop: leading-insertion cells of the root column (multiple of 8)
    int32_t  bonus_start, bonus_end;           // full-length bonuses in force (a graph N scores 0 + these)
    uint32_t K, n_strips, Lpad, n_slots;       // rows per lane; strips of 256 K rows; Lpad = n_strips * 256 * K rows per scratch slot
    uint64_t scratch_off;                      // in WPair
    uint64_t tb_off;                           // in dwords: strip-major, then step, lane, K / 8 dwords
    uint64_t carry_off;                        // in WPair: R entries
    uint64_t strip_dwords;                     // traceback dwords of one strip = (R + 255) * 256 * K / 8
};

struct WideParams {
    const WideProb* probs; uint32_t n;
    const uint32_t* order; uint32_t order_begin, order_count;     // this launch: problems order[order_begin ..) (those of one K)
    const uint8_t*  colinfo;
    const uint32_t* prof;
    const NodeRec*  nodes;
    const uint32_t* preds;
    WPair* scratch; WPair* carry;
    uint32_t* tb;
    unsigned long long* best;                  // LOCAL / XDROP: max over cells of key64(score, col, row)
    vgk_result* results; vgk_op* ops;
    int32_t bias, go, ge;
};

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ WPair wide_load(const WPair* p) {      // written earlier in this kernel by another lane / wavefront: bypass the CU's L1
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    WPair r; r.h = (int32_t)(uint32_t)v; r.x = (int32_t)(uint32_t)(v >> 32); return r;
}
#else
static inline WPair wide_load(const WPair* p) { return *p; }
#endif

VGK_HD int32_t wmax(int32_t a, int32_t b) { return a > b ? a : b; }

template <int K>
struct WLane {
    int32_t  H[K], E[K];          // H of the last processed column; E for the next one
    uint32_t P[K];                // profile words of this lane's rows (0 beyond the read)
    int32_t  out_h, out_f; uint32_t info;      // handed to lane + 1 at the next step
    int32_t  prev_rh;             // H of the row above this lane's block, previous column
    int32_t  best; uint32_t best_step, best_row;
    uint32_t node, row0, n_valid; // current node; first row; rows of this lane inside the read
    uint32_t nx_info; WPair nx_up; // lane 0: the next step's column byte and row above, fetched one step ahead
};

// what the row above the first strip holds: gssw's zero border, nothing reachable for X-drop
VGK_HD int32_t wide_floor(const WideProb& d) { return (d.flags & 15u) == VGK_XDROP_PINNED ? WNEG : 0; }

// lane 0 reads the column byte of step t from the stream and, in every strip but the first, the row above from the strip before —
// one step ahead, so that the load's latency hides behind a step of DP
template <int K>
VGK_HD void wide_lane_fetch(WLane<K>& s, const WideParams& P, const WideProb& d, uint32_t strip, uint32_t t) {
    const int32_t fl = wide_floor(d);
    s.nx_info = CI_INVALID; s.nx_up.h = fl; s.nx_up.x = fl;
    if (t < d.R) {
        s.nx_info = P.colinfo[d.col_off + t];
        if (strip) s.nx_up = wide_load(P.carry + d.carry_off + t);
    }
}

template <int K>
VGK_HD void wide_lane_init(WLane<K>& s, const WideParams& P, const WideProb& d, uint32_t strip, uint32_t lane) {
    s.row0 = (strip * WIDE_LANES + lane) * (uint32_t)K;
    s.n_valid = s.row0 >= d.L ? 0u : (d.L - s.row0 < (uint32_t)K ? d.L - s.row0 : (uint32_t)K);
    const int32_t fl = wide_floor(d);
#pragma unroll
    for (int m = 0; m < K; ++m) {
        s.P[m] = (uint32_t)m < s.n_valid ? P.prof[d.prof_off + s.row0 + m] : 0u;
        s.H[m] = fl; s.E[m] = fl;
    }
    s.out_h = fl; s.out_f = fl; s.info = CI_INVALID; s.prev_rh = fl;
    s.best = 0; s.best_step = 0; s.best_row = 0;
    s.node = 0xffffffffu;
    s.nx_info = CI_INVALID; s.nx_up.h = fl; s.nx_up.x = fl;
    if (lane == 0) wide_lane_fetch(s, P, d, strip, 0);
}

// SEED_SLOW: the column before a node's first = element-wise max over the predecessors' saved last columns (gssw_create_seed_*;
// dozeu merges incoming fronts the same way, src/dozeu_interface.cpp:261-283); a source node of an X-drop problem starts from
// dozeu's root column (dz_align_init).
template <int K>
VGK_HD void wide_seed(WLane<K>& s, const WideParams& P, const WideProb& d, int32_t& diag0) {
    const NodeRec& nr = P.nodes[d.node_off + s.node];
    const int32_t fl = wide_floor(d);
    if (nr.n_pred == 0 && fl != 0) {
#pragma unroll
        for (int m = 0; m <= K; ++m) {
            int32_t h = WNEG;
            if (m > 0 || s.row0 > 0) {
                const uint32_t row = s.row0 + (uint32_t)m - 1u;
                if (row == 0) h = 0;
                else if (row <= d.max_gap) h = -(P.go + (int32_t)(row - 1) * P.ge);
            }
            if (m == 0) diag0 = h;
            else { s.H[m - 1] = h; s.E[m - 1] = h > WNEG ? h - P.go : WNEG; }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < K; ++m) { s.H[m] = fl; s.E[m] = fl; }
    diag0 = fl;
    for (uint32_t k = 0; k < nr.n_pred; ++k) {
        const NodeRec& pr = P.nodes[d.node_off + P.preds[nr.pred_begin + k]];
        const WPair* base = P.scratch + d.scratch_off + (uint64_t)(uint32_t)pr.slot * d.Lpad + s.row0;
#pragma unroll
        for (int m = 0; m < K; ++m) { const WPair v = wide_load(base + m); s.H[m] = wmax(s.H[m], v.h); s.E[m] = wmax(s.E[m], v.x); }
        if (s.row0 > 0) diag0 = wmax(diag0, wide_load(base - 1).h);
    }
}

// One step of one lane.  rh / rf / rinfo: lane - 1's out_h / out_f / info of the previous step (lane 0 takes the column byte from the
// stream and the row above from the strip before).  tb: this (strip, step, lane)'s K / 8 dwords, or nullptr.
template <int K>
VGK_HD void wide_lane_step(WLane<K>& s, const WideParams& P, const WideProb& d, uint32_t strip, uint32_t lane, uint32_t t,
                           int32_t rh, int32_t rf, uint32_t rinfo, uint32_t* tb) {
    const int32_t fl = wide_floor(d);
    if (lane == 0) { rh = s.nx_up.h; rf = s.nx_up.x; rinfo = s.nx_info; wide_lane_fetch(s, P, d, strip, t + 1); }
    s.info = rinfo;
    if (!(rinfo & CI_INVALID) && s.n_valid) {
        int32_t diag0 = s.prev_rh;
        if (rinfo & CI_NODE_START) s.node += 1;
        if (rinfo & CI_SEED_SLOW) wide_seed<K>(s, P, d, diag0);
        const uint32_t base = rinfo & CI_BASE_MASK;
        const int32_t go = P.go, ge = P.ge, bias = P.bias;
        int32_t f = rf, dg = diag0;
        uint32_t acc = 0;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            int32_t sc;
            if (base < 4u) sc = (int32_t)((s.P[m] >> (8u * base)) & 0xffu) - bias;
            else sc = (int32_t)row_bonus((uint32_t)d.bonus_start, (uint32_t)d.bonus_end, s.row0 + (uint32_t)m, d.L);      // N scores 0 (+ bonus)
            const int32_t t4 = dg + sc, e = s.E[m];
            const int32_t h = wmax(wmax(t4, e), wmax(f, fl));
            const int32_t old = s.H[m];
            const int32_t gg = h - go;
            const int32_t en = wmax(wmax(gg, e - ge), fl), fn = wmax(wmax(gg, f - ge), fl);
            // bit0 = H not from the diagonal, bit1 = H not from E (then F), bit2 = next-column E is an extension, bit3 = next-row F is one
            // (ties: diagonal > E > F, open > extend — the packed kernel's rules)
            const uint32_t code = (h > t4 ? 1u : 0u) | (h > e ? 2u : 0u) | (en > gg ? 4u : 0u) | (fn > gg ? 8u : 0u);
            acc |= code << (4u * ((uint32_t)m & 7u));
            if (((uint32_t)m & 7u) == 7u) { if (tb) tb[m >> 3] = acc; acc = 0; }
            // end cell: first column with the best score, smallest row (SSW's rule; dozeu: first node / column, smallest query position)
            if ((uint32_t)m < s.n_valid && h > s.best) { s.best = h; s.best_step = t; s.best_row = s.row0 + (uint32_t)m; }
            s.H[m] = h; s.E[m] = en; f = fn; dg = old;
        }
        s.out_h = s.H[K - 1]; s.out_f = f;
        if (rinfo & CI_STORE_END) {
            const NodeRec& nr = P.nodes[d.node_off + s.node];
            WPair* out = P.scratch + d.scratch_off + (uint64_t)(uint32_t)nr.slot * d.Lpad + s.row0;
#pragma unroll
            for (int m = 0; m < K; ++m) { WPair v; v.h = s.H[m]; v.x = s.E[m]; out[m] = v; }
        }
        if (lane == WIDE_LANES - 1u && strip + 1u < d.n_strips) { WPair v; v.h = s.out_h; v.x = s.out_f; P.carry[d.carry_off + (t - lane)] = v; }
    } else {
        s.out_h = fl; s.out_f = fl;
    }
    s.prev_rh = rh;
}

// after a strip's last step: this lane's best cell (LOCAL and X-drop; the pinned end cell is read off the scratch by the walk)
template <int K>
VGK_HD bool wide_lane_best(const WLane<K>& s, const WideProb& d, uint32_t lane, unsigned long long& key) {
    if ((d.flags & 15u) == VGK_GSSW_PINNED || s.best <= 0) return false;
    const uint32_t sc = (uint32_t)s.best < 0xffffffu ? (uint32_t)s.best : 0xffffffu;
    key = key64(sc, s.best_step - lane, s.best_row);
    return true;
}

// ---------------------------------------------------------------------------
// traceback: one lane per problem — gssw_device.hpp's walk_one over int32 scratch and this layout of the codes
// ---------------------------------------------------------------------------
struct WideWalker {
    const WideParams& P; const WideProb& d;
    VGK_HD uint32_t code(uint32_t r, uint32_t c) const {
        const uint32_t per = WIDE_LANES * d.K, strip = r / per, rr = r - strip * per, l = rr / d.K, m = rr - l * d.K;
        const uint64_t at = d.tb_off + (uint64_t)strip * d.strip_dwords + ((uint64_t)(c + l) * WIDE_LANES + l) * (d.K >> 3) + (m >> 3);
        return (P.tb[at] >> (4u * (m & 7u))) & 15u;
    }
    VGK_HD int32_t score(uint32_t r, uint32_t c) const {
        const uint32_t base = P.colinfo[d.col_off + c] & CI_BASE_MASK;
        if (base >= 4u) return (int32_t)row_bonus((uint32_t)d.bonus_start, (uint32_t)d.bonus_end, r, d.L);
        return (int32_t)((P.prof[d.prof_off + r] >> (8u * base)) & 0xffu) - P.bias;
    }
    VGK_HD WPair saved(const NodeRec& n, uint32_t r) const { return P.scratch[d.scratch_off + (uint64_t)(uint32_t)n.slot * d.Lpad + r]; }
};

VGK_HD void wide_walk_one(const WideParams& P, uint32_t i) {
    const WideProb d = P.probs[i];
    vgk_result res;
    res.score = 0; res.status = VGK_OK; res.end_node = -1; res.end_offset = -1; res.end_read = -1;
    res.first_offset = 0; res.n_ops = 0; res.ops_begin = d.ops_off;
    const WideWalker w{P, d};
    const NodeRec* nodes = P.nodes + d.node_off;
    const bool pinned = (d.flags & 15u) == VGK_GSSW_PINNED;
    const bool xdrop = (d.flags & 15u) == VGK_XDROP_PINNED;
    const int32_t go = P.go, ge = P.ge;

    int32_t cur = 0; uint32_t c = 0, node = 0; int32_t r = 0;
    bool have = false;
    if (pinned) {
        r = (int32_t)d.L - 1;
        for (uint32_t n = 0; n < d.n_nodes; ++n) {
            if (!nodes[n].pinning) continue;
            const int32_t v = w.saved(nodes[n], (uint32_t)r).h;
            if (!have || v > cur) { cur = v; node = n; have = true; }
        }
        if (have) c = nodes[node].col_end - 1;
    } else {
        const unsigned long long k = P.best[i];
        cur = (int32_t)(k >> 40);
        if (cur > 0) {
            have = true;
            c = 0xFFFFFu - (uint32_t)((k >> 20) & 0xFFFFFu);
            r = (int32_t)(0xFFFFFu - (uint32_t)(k & 0xFFFFFu));
            uint32_t lo = 0, hi = d.n_nodes;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nodes[mid].col_start <= c) lo = mid; else hi = mid; }
            node = lo;
        }
    }
    if (pinned && !have) { res.status = VGK_EINVAL; P.results[i] = res; return; }
    if (cur >= 32767) { res.status = VGK_EOVERFLOW; P.results[i] = res; return; }      // gssw's word mode / dozeu's int16 cells
    if (!have || cur <= 0) { P.results[i] = res; return; }
    res.score = cur; res.end_node = (int32_t)node; res.end_offset = (int32_t)(c - nodes[node].col_start);
    res.end_read = xdrop ? r - 1 : r;
    if (!(d.flags & VGK_GSSW_TRACEBACK)) { P.results[i] = res; return; }

    vgk_op* ops = P.ops + d.ops_off;
    uint32_t pos = d.ops_cap;
    int32_t status = VGK_OK;
    uint32_t rn = 0, ro = 0xffu, rl = 0;
    // (a run longer than vgk_op.len's 16 bits is split: an insertion of a whole long read)
#define VGW_FLUSH() do { if (rl) { if (pos == 0) status = VGK_EOPS; else { --pos; \
        ops[pos].node = rn; ops[pos].len = (uint16_t)rl; ops[pos].op = (uint8_t)ro; ops[pos].pad = 0; } rl = 0; } } while (0)
#define VGW_PUSH(NODE, OP, LEN) do { uint32_t len_ = (LEN); \
        while (len_) { \
            if (!(rl && rn == (uint32_t)(NODE) && ro == (uint32_t)(OP))) { VGW_FLUSH(); rn = (uint32_t)(NODE); ro = (uint32_t)(OP); } \
            const uint32_t take_ = len_ < 65535u - rl ? len_ : 65535u - rl; rl += take_; len_ -= take_; \
            if (len_) VGW_FLUSH(); } } while (0)

    if (r < (int32_t)d.L - 1) VGW_PUSH(node, VGK_OP_S, d.L - 1 - (uint32_t)r);
    enum { ST_H, ST_E, ST_F } st = ST_H;
    bool at_root = false;
    uint32_t first_c = c;
    uint32_t node_start = nodes[node].col_start;
    for (uint64_t guard = 0; guard < 2ull * ((uint64_t)d.L + d.R) + 4 && status == VGK_OK; ++guard) {
        const bool first = (c == node_start);
        if (st == ST_H) {
            if (!xdrop && cur == 0) break;
            const uint32_t fl = w.code((uint32_t)r, c);
            if (fl & 1u) { st = (fl & 2u) ? ST_F : ST_E; continue; }
            VGW_PUSH(node, VGK_OP_M, 1); first_c = c;
            cur -= w.score((uint32_t)r, c); r -= 1;
            if (r < 0 || (!xdrop && cur == 0)) break;
            if (!first) c -= 1;
            else {
                const NodeRec& nr = nodes[node];
                if (xdrop && nr.n_pred == 0) {           // back at the dozeu root: r read bases are a leading insertion
                    if (r > 0) VGW_PUSH(node, VGK_OP_I, (uint32_t)r);
                    at_root = true; break;
                }
                int32_t found = -1;
                if (nr.n_pred == 1) found = (int32_t)P.preds[nr.pred_begin];
                else for (uint32_t kk = 0; kk < nr.n_pred; ++kk) {
                    const uint32_t p = P.preds[nr.pred_begin + kk];
                    if (w.saved(nodes[p], (uint32_t)r).h == cur) { found = (int32_t)p; break; } }
                if (found < 0) { status = VGK_EINVAL; break; }
                node = (uint32_t)found; c = nodes[node].col_end - 1; node_start = nodes[node].col_start;
            }
        } else if (st == ST_E) {
            VGW_PUSH(node, VGK_OP_D, 1); first_c = c;
            uint32_t pnode = node, pc = c - 1;
            if (first) {
                const NodeRec& nr = nodes[node];
                if (xdrop && nr.n_pred == 0) {           // deletion opened straight from the root column
                    if (r > 0) VGW_PUSH(node, VGK_OP_I, (uint32_t)r);
                    at_root = true; break;
                }
                int32_t found = -1;
                if (nr.n_pred == 1) found = (int32_t)P.preds[nr.pred_begin];
                else for (uint32_t k = 0; k < nr.n_pred; ++k) {
                    const uint32_t p = P.preds[nr.pred_begin + k];
                    if (w.saved(nodes[p], (uint32_t)r).x == cur) { found = (int32_t)p; break; } }
                if (found < 0) { status = VGK_EINVAL; break; }
                pnode = (uint32_t)found; pc = nodes[pnode].col_end - 1; node_start = nodes[pnode].col_start;
            }
            if (!(w.code((uint32_t)r, pc) & 4u)) { st = ST_H; cur += go; } else cur += ge;
            c = pc; node = pnode;
        } else {
            VGW_PUSH(node, VGK_OP_I, 1);
            if (r == 0) { status = VGK_EINVAL; break; }
            if (!(w.code((uint32_t)r - 1, c) & 8u)) { st = ST_H; cur += go; } else cur += ge;
            r -= 1;
        }
    }
    if (xdrop && status == VGK_OK && !at_root) status = VGK_EINVAL;     // a dozeu path always ends at the root
    if (!xdrop && status == VGK_OK && r >= 0) VGW_PUSH(node, VGK_OP_S, (uint32_t)r + 1);
    if (status == VGK_OK) VGW_FLUSH();
#undef VGW_PUSH
#undef VGW_FLUSH
    res.status = status;
    if (status == VGK_OK) {
        res.n_ops = d.ops_cap - pos; res.ops_begin = d.ops_off + pos;
        res.first_offset = (int32_t)(first_c - node_start);
    }
    P.results[i] = res;
}

}  // namespace vgk
