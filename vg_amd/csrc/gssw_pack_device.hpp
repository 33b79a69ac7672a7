// gssw_pack_device.hpp — packing of gssw / X-drop problems ON THE DEVICE, for callers whose reads all align against windows
// of ONE graph that stays resident in HBM (vgk_graph_create + vgk_gssw_pack_windows).
//
// What the CPU does per read in vg — cut a subgraph around the seed cluster (src/mapper.cpp:2445-2518, the id range around
// the MEMs), convert it node by node with create_gssw_graph (src/aligner.cpp:30-85: one malloc'd gssw_node per node, an
// unordered_map, one gssw_nodes_add_edge per edge) — becomes: the whole graph is encoded once (column info bytes, node and
// predecessor tables), a problem is {read, first node, node count} = the induced subgraph on a run of consecutive nodes of
// the topological order, and four small kernels derive every per-problem arena the fill / traceback kernels read
// (gssw_device.hpp: ProbDesc, NodeRec, preds, colinfo, read codes, the wave order) from the resident tables.  The host
// touches no problem: it copies two flat buffers to HBM and sizes the arenas from totals the kernels report.
//
// The arenas come out exactly as vgk_gssw_pack (vgk_api.cpp) would encode the induced subgraphs, with two harmless
// supersets: a node whose predecessors are partly outside the window keeps its "seed from scratch" flag even when the only
// one left is the previous node, and a node keeps its scratch slot when the successor that needed it lies outside the window.
// Neither changes a DP value (the slow seed path is the general case of the chain path).
//
// Plain C++ over VGK_HD like the lane code, so tests/emu runs the same functions on the CPU.
#pragma once
#include <stdint.h>
#include "../../include/vgk.h"
#include "gssw_device.hpp"

namespace vgk {

// Lane geometry for a read of `rows` DP rows: rows per lane K (16, 19, 20, 24) and lanes per pair G = ceil(rows/K) — the
// instantiation that spends the fewest issued instructions per useful cell: (25 K + 60) per step buys floor(64/G) * 2 * rows
// cells.  Exact integer comparison (the host packer and the device packer must agree).  19 rows per lane fit a 150 bp read into
// 8 lanes with 2 padding rows instead of 10; for short reads a fourth launch bucket costs more than the rows it saves.
VGK_HD void lane_geometry(uint32_t rows, uint32_t forced, uint32_t& K, uint32_t& G) {
    const uint32_t ks[4] = {16u, 19u, 20u, 24u};
    uint64_t best_num = 0, best_den = 0; K = 16;
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = ks[j], g = (rows + k - 1) / k;
        if (g > 64) continue;
        if (k == 19 && rows < 128 && forced != k) continue;
        if (forced == k) { K = k; break; }
        const uint64_t num = k * 25ull + 60ull, den = (64 / g) * 2ull * rows;      // cost = num / den
        if (!forced && (best_den == 0 || num * best_den < best_num * den)) { best_num = num; best_den = den; K = k; }
    }
    G = (rows + K - 1) / K;
}
VGK_HD uint32_t geometry_index(uint32_t K) { return K == 16 ? 0u : K == 19 ? 1u : K == 20 ? 2u : 3u; }
constexpr uint32_t WIN_BUCKETS = 4 * 65;          // (K index, G) pairs
constexpr uint32_t WIN_COLS = 6;                  // size columns that get prefix sums
enum { WS_COLS = 0, WS_NODES = 1, WS_PREDS = 2, WS_READS = 3, WS_SCRATCH = 4, WS_OPS = 5 };

struct WinGraph {                // the resident graph (device pointers)
    const uint32_t* col;         // [n_nodes + 1] first column of node v (col[n_nodes] = all columns)
    const uint8_t*  info;        // [n_cols + 8] column info bytes of the whole graph: base code | CI_NODE_START | CI_SEED_SLOW | CI_STORE_END
    const uint32_t* pred_off;    // [n_nodes + 1]
    const uint32_t* pred_idx;
    const uint32_t* slot;        // [n_nodes + 1] stored nodes before node v (a node is stored when a successor seeds from scratch)
    uint32_t n_nodes, n_cols;
    const uint32_t* succ_off;    // [n_nodes + 1] successors as CSR (ascending per node): what a LEFTWARD extension window walks; null for graphs made
    const uint32_t* succ_idx;    //               on the device (tail forests), which are never extended leftwards
};

// An EXTENSION window (vgk_rescue_align, rescue_api.cpp): a pinned X-drop extension from a position INSIDE a window of the resident graph — one of
// the two passes of Aligner::align_xdrop (DozeuInterface::align, src/dozeu_interface.cpp:608-685), as Aligner::xdrop_extend_prepare builds it on
// the host for one subgraph at a time (vg_amd/host/aligner.cpp): the part of the start node that lies in the direction of the extension, then the
// window's nodes REACHABLE from it in that direction, in extension order (a leftward pass: descending, every node's bases reversed, successors as
// predecessors), with the read part on that side of `query_offset` (a leftward pass: reversed).  A start exactly at the node's end is not part
// of the problem: its neighbours start from the root column.  Nodes the start cannot reach are NOT part of the problem (in dozeu's pinned mode
// every source would be a root), so the problem's node k is kept[kept_off + k], which stage 1 leaves in `kept` for the caller.
struct WinExt {
    uint32_t start_node;         // index in the resident graph; inside the window
    uint32_t start_offset;       // rightward: the node's bases [start_offset, length) belong to the problem; leftward: its bases [0, start_offset), reversed
    uint32_t query_offset;       // rightward: read[query_offset, read_len); leftward: read[0, query_offset) reversed
    uint32_t leftward;
    uint32_t kept_off;           // first entry of this problem in the per-node temporaries (a prefix sum of the windows' n_nodes)
    uint32_t pad;
};
struct WinKept {                 // per node of an extension window's problem, written by stage 1 (one lane per problem), read by stage 2 (a wavefront per problem)
    uint32_t node;               // index in the resident graph
    uint32_t col;                // its first column in the problem
    uint32_t pred_at;            // predecessors (inside the problem) of the nodes before it
    uint32_t slot_flags;         // stored nodes before it << 2 | store << 1 | slow
};
constexpr uint32_t WIN_EXT_DUMMY = 0x80000000u;       // ext_count[i]: nothing lies in the extension's direction — the problem is one N column that scores nothing; the caller takes "did not run"

struct WinBucket { uint32_t s0, s1, K, G, pair0, pair1, wave0, pad; };

struct WinTotals {               // one per pack, device memory, zeroed before the first kernel
    unsigned long long tot[WIN_COLS];      // arena sizes (entries)
    unsigned long long cells, tb_cells, in_bytes;
    unsigned long long first_bad;          // min over failing problems of (index << 8 | -status), ~0 = none
    unsigned long long tb_dwords;
    unsigned long long wave_steps;         // sum over wavefronts of their steps (one step = one column for every lane)
    uint32_t max_rows, want_tb;
    uint32_t n_pairs, n_waves, n_buckets, n_launches;
    uint32_t launch_K[4], launch_begin[4], launch_count[4];
    uint32_t max_steps, first_G;           // the longest wavefront's steps; lanes per pair of wavefront 0 (the speculative fill, window_api.cpp: with ONE bucket that is every wavefront's)
};

struct WinParams {
    WinGraph g;
    const vgk_window_problem* problems; uint32_t n;
    const uint8_t* raw_reads; unsigned long long raw_bytes;     // the caller's reads, ASCII
    uint32_t ops_per_problem, forced_k;
    int32_t  max_score, max_bonus; uint32_t scale; int32_t bonus;
    int32_t  want_tb;            // stage 2: any problem wants a traceback (from totals)
    uint32_t n_waves_cap;        // entries allocated for waves / wave_tb
    // stage 1 (sizes): per problem, written by win_size_one
    uint32_t* sizes;             // [WIN_COLS][n + 1]
    uint32_t* offs;              // [WIN_COLS][n + 1]   exclusive prefix sums of `sizes`
    uint32_t* key; uint32_t* idx;                   // sort key (bucket << 16 | 0xffff - min(R, 0xffff)) and problem index
    uint32_t* key_sorted; uint32_t* idx_sorted;
    WinTotals* totals;
    uint32_t* bucket_first;      // [WIN_BUCKETS] first sorted position of the bucket, 0xffffffff = empty
    WinBucket* buckets;          // [WIN_BUCKETS] the non-empty ones, in launch order
    unsigned long long* wave_tb; // [n_waves_cap + 1] traceback dwords per wave, then their exclusive sums
    // extension windows (null: plain windows): per problem its start; per window node the temporaries; per problem the nodes kept
    const WinExt* ext; WinKept* kept; uint32_t* ext_count; uint32_t* kept_node;      // kept_node[kept_off + k] = node k of the problem, counted from the window's first node (what the caller downloads)
    // stage 2 outputs: the arenas of GsswParams
    ProbDesc* probs; uint8_t* colinfo; uint8_t* reads; NodeRec* nodes; uint32_t* preds; WaveDesc* waves; uint32_t* order;
};

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ void acc_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
static __device__ __forceinline__ void acc_min(unsigned long long* p, unsigned long long v) { atomicMin(p, v); }
static __device__ __forceinline__ void acc_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
static __device__ __forceinline__ void acc_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
#else
static inline void acc_add(unsigned long long* p, unsigned long long v) { *p += v; }
static inline void acc_min(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
static inline void acc_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
static inline void acc_or(uint32_t* p, uint32_t v) { *p |= v; }
#endif

VGK_HD uint32_t win_nt_read(uint32_t ch) {   // gssw_create_nt_table: case-insensitive ACGT, else N
    ch &= ~0x20u;
    return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
}

// per-problem sums a caller accumulates (per thread, per block, ...) before adding them to the totals once
struct WinAcc { unsigned long long tot[WIN_COLS], cells, tb_cells, in_bytes; uint32_t max_rows, want_tb; };
VGK_HD void win_acc_clear(WinAcc& a) { for (uint32_t k = 0; k < WIN_COLS; ++k) a.tot[k] = 0; a.cells = a.tb_cells = a.in_bytes = 0; a.max_rows = 0; a.want_tb = 0; }
VGK_HD void win_acc_flush(const WinParams& P, const WinAcc& a) {
    for (uint32_t k = 0; k < WIN_COLS; ++k) if (a.tot[k]) acc_add(&P.totals->tot[k], a.tot[k]);
    if (a.cells) acc_add(&P.totals->cells, a.cells);
    if (a.tb_cells) acc_add(&P.totals->tb_cells, a.tb_cells);
    if (a.in_bytes) acc_add(&P.totals->in_bytes, a.in_bytes);
    if (a.max_rows) acc_max(&P.totals->max_rows, a.max_rows);
    if (a.want_tb) acc_or(&P.totals->want_tb, 1u);
}

// Stage 1, one call per problem: validate, size, choose the lane geometry, fill the fields of ProbDesc that need no offsets.
// The checks are those of vgk_gssw_pack (vgk_api.cpp); a failing problem reports (index, status) and sizes to nothing.
// position of graph node v among the nodes kept so far (ascending for a rightward extension, descending for a leftward one), or 0xffffffff
VGK_HD uint32_t ext_find(const WinKept* kept, uint32_t count, uint32_t v, bool leftward) {
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1, u = kept[mid].node;
        if (u == v) return mid;
        if ((u < v) != leftward) lo = mid + 1; else hi = mid;
    }
    return 0xffffffffu;
}

// Stage 1 for an extension window, one call per problem: which nodes the start reaches inside the window, their columns, predecessor counts,
// stored / slow flags — and from those the sizes, exactly as win_size_one sizes a plain window.
VGK_HD void ext_size_one(const WinParams& P, uint32_t i, WinAcc& acc) {
    const vgk_window_problem p = P.problems[i];
    const WinExt x = P.ext[i];
    const uint32_t n1 = P.n + 1;
    ProbDesc d;
    d.col_off = d.R = d.L = d.read_off = d.node_off = d.n_nodes = d.scratch_off = d.n_slots = d.flags = d.ops_off = d.ops_cap = 0;
    d.max_gap = d.wave = d.lane0 = d.geom = d.Lpad = d.bonus_start = d.bonus_end = d.pad = 0; d.prof_off = 0xffffffffu;
    int status = VGK_OK;
    const bool left = x.leftward != 0;
    const uint32_t qlen = left ? x.query_offset : (x.query_offset <= p.read_len ? p.read_len - x.query_offset : 0u);
    const uint32_t rows = qlen + 1u;
    if (p.read_len == 0 || p.n_nodes == 0 || (unsigned long long)p.first_node + p.n_nodes > P.g.n_nodes ||
        p.read_off + p.read_len > P.raw_bytes || p.read_off + p.read_len < p.read_off) status = VGK_EINVAL;
    else if ((p.flags & 15u) != VGK_XDROP_PINNED || x.query_offset > p.read_len || qlen == 0) status = VGK_EINVAL;
    else if (x.start_node < p.first_node || x.start_node >= p.first_node + p.n_nodes || (left && !P.g.succ_off)) status = VGK_EINVAL;
    else if (x.start_offset > P.g.col[x.start_node + 1] - P.g.col[x.start_node]) status = VGK_EINVAL;
    else if (rows > 1024) status = VGK_ETOOLONG;
    else if ((long long)rows * (P.max_score > 0 ? P.max_score : 0) + 2ll * P.max_bonus > 2046) status = VGK_EUNSUPPORTED;
    else if ((long long)qlen * (P.max_score > 0 ? P.max_score : 0) + P.max_bonus >= (long long)XOFF) status = VGK_EUNSUPPORTED;
    uint32_t R = 0, slots = 0, n_preds = 0, count = 0;
    WinKept* kept = P.kept + x.kept_off;
    if (status == VGK_OK) {
        const uint32_t a = p.first_node, b = p.first_node + p.n_nodes, s = x.start_node;
        const uint32_t start_len = left ? x.start_offset : (P.g.col[s + 1] - P.g.col[s]) - x.start_offset;
        const bool skip_start = start_len == 0;
        if (!skip_start) { kept[0].node = s; kept[0].col = 0; kept[0].pred_at = 0; kept[0].slot_flags = 1u; count = 1; R = start_len; }      // (the start: the root column, no predecessors)
        // the window's other nodes in extension order: reached when a neighbour on the start's side is (the start itself counts, kept or not)
        for (uint32_t step = 1; left ? s >= a + step : s + step < b; ++step) {
            const uint32_t v = left ? s - step : s + step;
            const uint32_t eb = left ? P.g.succ_off[v] : P.g.pred_off[v], ee = left ? P.g.succ_off[v + 1] : P.g.pred_off[v + 1];
            const uint32_t* nb = left ? P.g.succ_idx : P.g.pred_idx;
            bool reach = false; uint32_t np = 0, only = 0xffffffffu;
            for (uint32_t e = eb; e < ee; ++e) {
                const uint32_t q = nb[e];
                if (left ? (q > s || q < a) : (q < s || q >= b)) continue;             // beyond the start / outside the window
                if (q == s) { reach = true; if (!skip_start) { ++np; only = 0; } continue; }
                const uint32_t at = ext_find(kept, count, q, left);
                if (at != 0xffffffffu) { reach = true; ++np; only = at; }
            }
            if (!reach) continue;
            const bool chain = np == 1 && only + 1 == count;
            WinKept k; k.node = v; k.col = R; k.pred_at = n_preds; k.slot_flags = chain ? 0u : 1u;
            kept[count] = k;
            if (!chain)                                                                // its predecessors' last columns are saved
                for (uint32_t e = eb; e < ee; ++e) {
                    const uint32_t q = nb[e];
                    if (left ? (q > s || q < a) : (q < s || q >= b)) continue;
                    const uint32_t at = (q == s) ? (skip_start ? 0xffffffffu : 0u) : ext_find(kept, count, q, left);
                    if (at != 0xffffffffu) kept[at].slot_flags |= 2u;
                }
            n_preds += np; R += P.g.col[v + 1] - P.g.col[v]; ++count;
        }
        for (uint32_t k = 0; k < count; ++k) { const uint32_t f = kept[k].slot_flags & 3u; kept[k].slot_flags = (slots << 2) | f; slots += (f >> 1) & 1u; P.kept_node[x.kept_off + k] = kept[k].node - a; }
        if (count == 0) { R = 1; }                                                       // the dummy column (WIN_EXT_DUMMY)
        if (R >= (1u << 20)) status = VGK_ETOOBIG;
    }
    uint32_t key = 0xffffffffu;
    if (status != VGK_OK) {
        acc_min(&P.totals->first_bad, ((unsigned long long)i << 8) | (unsigned long long)(uint32_t)(-status));
        for (uint32_t k = 0; k < WIN_COLS; ++k) P.sizes[k * n1 + i] = 0;
        key = ((WIN_BUCKETS - 1) << 16) | 0xffffu;
        P.ext_count[i] = 0;
    } else {
        uint32_t K, G; lane_geometry(rows, P.forced_k, K, G);
        const uint32_t nn = count ? count : 1u;
        d.R = R; d.L = rows; d.n_nodes = nn; d.n_slots = slots; d.flags = p.flags;
        d.node_off = p.first_node;
        d.max_gap = ((p.max_gap_length > 1u ? p.max_gap_length : 1u) + 7u) & ~7u;
        d.geom = K | (G << 8); d.Lpad = G * K;
        d.bonus_start = 0u; d.bonus_end = (uint32_t)P.bonus * P.scale;
        const bool tb = (p.flags & VGK_GSSW_TRACEBACK) != 0;
        d.ops_cap = tb ? (P.ops_per_problem ? P.ops_per_problem : qlen + R + 2u) : 0u;
        const uint32_t sz[WIN_COLS] = {(R + 3u) & ~3u, nn, n_preds, (rows + 3u) & ~3u, slots * d.Lpad, d.ops_cap};
        for (uint32_t k = 0; k < WIN_COLS; ++k) { P.sizes[k * n1 + i] = sz[k]; acc.tot[k] += sz[k]; }
        acc.cells += (unsigned long long)R * rows;
        if (tb) { acc.tb_cells += (unsigned long long)R * rows; acc.want_tb = 1; }
        acc.in_bytes += (unsigned long long)qlen + R + 8ull * nn + 4ull * n_preds;
        acc.max_rows = acc.max_rows > rows ? acc.max_rows : rows;
        key = ((geometry_index(K) * 65u + G) << 16) | (0xffffu - (R < 0xffffu ? R : 0xffffu));
        P.ext_count[i] = count ? count : WIN_EXT_DUMMY;
    }
    P.probs[i] = d;
    P.key[i] = key; P.idx[i] = i;
}

VGK_HD void win_size_one(const WinParams& P, uint32_t i, WinAcc& acc) {
    if (P.ext) { ext_size_one(P, i, acc); return; }
    const vgk_window_problem p = P.problems[i];
    const uint32_t n1 = P.n + 1;
    ProbDesc d;
    d.col_off = d.R = d.L = d.read_off = d.node_off = d.n_nodes = d.scratch_off = d.n_slots = d.flags = d.ops_off = d.ops_cap = 0;
    d.max_gap = d.wave = d.lane0 = d.geom = d.Lpad = d.bonus_start = d.bonus_end = d.pad = 0; d.prof_off = 0xffffffffu;
    int status = VGK_OK;
    const uint32_t mode = p.flags & 15u;
    const bool xdrop = mode == VGK_XDROP_PINNED;
    const uint32_t rows = p.read_len + (xdrop ? 1u : 0u);
    if (p.read_len == 0 || p.n_nodes == 0 || (unsigned long long)p.first_node + p.n_nodes > P.g.n_nodes ||
        p.read_off + p.read_len > P.raw_bytes || p.read_off + p.read_len < p.read_off) status = VGK_EINVAL;
    else if (mode != VGK_GSSW_LOCAL && mode != VGK_XDROP_PINNED) status = VGK_EINVAL;      // pinned windows: not offered (the pinning nodes depend on where the window ends)
    else if (rows > 1024) status = VGK_ETOOLONG;
    else if ((long long)rows * (P.max_score > 0 ? P.max_score : 0) + 2ll * P.max_bonus > 2046) status = VGK_EUNSUPPORTED;
    else if (xdrop && (long long)p.read_len * (P.max_score > 0 ? P.max_score : 0) + P.max_bonus >= (long long)XOFF) status = VGK_EUNSUPPORTED;
    uint32_t R = 0, slots = 0, n_preds = 0;
    if (status == VGK_OK) {
        const uint32_t a = p.first_node, b = p.first_node + p.n_nodes;
        R = P.g.col[b] - P.g.col[a]; slots = P.g.slot[b] - P.g.slot[a]; n_preds = P.g.pred_off[b] - P.g.pred_off[a];
        if (R >= (1u << 20)) status = VGK_ETOOBIG;
    }
    uint32_t key = 0xffffffffu;
    if (status != VGK_OK) {
        acc_min(&P.totals->first_bad, ((unsigned long long)i << 8) | (unsigned long long)(uint32_t)(-status));
        for (uint32_t k = 0; k < WIN_COLS; ++k) P.sizes[k * n1 + i] = 0;
        key = ((WIN_BUCKETS - 1) << 16) | 0xffffu;
    } else {
        uint32_t K, G; lane_geometry(rows, P.forced_k, K, G);
        d.R = R; d.L = rows; d.n_nodes = p.n_nodes; d.n_slots = slots; d.flags = p.flags;
        d.node_off = p.first_node;                      // until win_emit_one places the problem's NodeRecs
        d.max_gap = xdrop ? (((p.max_gap_length > 1u ? p.max_gap_length : 1u) + 7u) & ~7u) : 0u;
        d.geom = K | (G << 8); d.Lpad = G * K;
        d.bonus_start = xdrop ? 0u : (uint32_t)P.bonus * P.scale;
        d.bonus_end = (uint32_t)P.bonus * P.scale;      // (pinned windows are not offered: the end bonus always applies)
        const bool tb = (p.flags & VGK_GSSW_TRACEBACK) != 0;
        d.ops_cap = tb ? (P.ops_per_problem ? P.ops_per_problem : p.read_len + R + 2u) : 0u;
        const uint32_t s[WIN_COLS] = {(R + 3u) & ~3u, p.n_nodes, n_preds, (rows + 3u) & ~3u, slots * d.Lpad, d.ops_cap};
        for (uint32_t k = 0; k < WIN_COLS; ++k) { P.sizes[k * n1 + i] = s[k]; acc.tot[k] += s[k]; }
        acc.cells += (unsigned long long)R * rows;
        if (tb) { acc.tb_cells += (unsigned long long)R * rows; acc.want_tb = 1; }
        acc.in_bytes += (unsigned long long)p.read_len + R + 8ull * p.n_nodes + 4ull * n_preds;
        acc.max_rows = acc.max_rows > rows ? acc.max_rows : rows;
        key = ((geometry_index(K) * 65u + G) << 16) | (0xffffu - (R < 0xffffu ? R : 0xffffu));
    }
    P.probs[i] = d;
    P.key[i] = key; P.idx[i] = i;
}

// after the sort: position j starts a bucket when its (K, G) differs from its left neighbour's
VGK_HD void win_bucket_first_one(const WinParams& P, uint32_t j) {
    const uint32_t b = P.key_sorted[j] >> 16;
    if (j == 0 || (P.key_sorted[j - 1] >> 16) != b) P.bucket_first[b] = j;
}

// one thread: the non-empty buckets in launch order, their pairs and wavefronts, one fill launch per K
VGK_HD void win_buckets(const WinParams& P) {
    WinTotals& T = *P.totals;
    uint32_t nb = 0, n_pairs = 0, n_waves = 0, n_launches = 0;
    for (uint32_t b = 0; b + 1 < WIN_BUCKETS; ++b) {             // the last bucket id holds the failed problems (none when a pack goes ahead)
        const uint32_t s0 = P.bucket_first[b];
        if (s0 == 0xffffffffu) continue;
        uint32_t s1 = P.n;
        for (uint32_t c = b + 1; c < WIN_BUCKETS; ++c) if (P.bucket_first[c] != 0xffffffffu) { s1 = P.bucket_first[c]; break; }
        const uint32_t ks[4] = {16u, 19u, 20u, 24u};
        const uint32_t K = ks[b / 65u], G = b % 65u, gpw = 64u / G;
        WinBucket bk; bk.s0 = s0; bk.s1 = s1; bk.K = K; bk.G = G; bk.pair0 = n_pairs; bk.pair1 = n_pairs + (s1 - s0 + 1) / 2; bk.wave0 = n_waves; bk.pad = 0;
        if (n_launches == 0 || T.launch_K[n_launches - 1] != K) { T.launch_K[n_launches] = K; T.launch_begin[n_launches] = n_waves; T.launch_count[n_launches] = 0; ++n_launches; }
        n_pairs = bk.pair1; n_waves += (bk.pair1 - bk.pair0 + gpw - 1) / gpw;
        T.launch_count[n_launches - 1] = n_waves - T.launch_begin[n_launches - 1];
        P.buckets[nb++] = bk;
    }
    T.n_buckets = nb; T.n_pairs = n_pairs; T.n_waves = n_waves; T.n_launches = n_launches;
}

// one call per wavefront: its read pairs, where each of its reads sits, how many steps it runs, its traceback dwords
VGK_HD void win_wave_one(const WinParams& P, uint32_t w) {
    const WinTotals& T = *P.totals;
    if (w >= T.n_waves || w >= P.n_waves_cap) return;
    uint32_t bi = 0;
    while (bi + 1 < T.n_buckets && P.buckets[bi + 1].wave0 <= w) ++bi;
    const WinBucket bk = P.buckets[bi];
    const uint32_t gpw = 64u / bk.G, pw = bk.pair0 + (w - bk.wave0) * gpw;
    WaveDesc wd; wd.tb_off = 0; wd.first_pair = pw; wd.G = bk.G; wd.pair_end = bk.pair1;
    uint32_t rmax = 0;
    for (uint32_t q = 0; q < gpw && pw + q < bk.pair1; ++q)
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t k = bk.s0 + 2 * (pw + q - bk.pair0) + h;
            const uint32_t i = k < bk.s1 ? P.idx_sorted[k] : 0xffffffffu;
            P.order[2 * (size_t)(pw + q) + h] = i;
            if (i == 0xffffffffu) continue;
            ProbDesc& d = P.probs[i];
            rmax = rmax > d.R ? rmax : d.R;
            d.wave = w; d.lane0 = q * bk.G; d.geom = bk.K | (bk.G << 8) | (h << 16);
        }
    wd.n_steps = rmax ? rmax + bk.G - 1 : 0;
    if (wd.n_steps) { acc_add(&P.totals->wave_steps, wd.n_steps); acc_max(&P.totals->max_steps, wd.n_steps); }
    if (w == 0) P.totals->first_G = bk.G;
    P.waves[w] = wd;
    P.wave_tb[w] = P.want_tb ? (unsigned long long)tb_wave_dwords(wd.n_steps, bk.K) : 0ull;
}
VGK_HD void win_wave_tb_one(const WinParams& P, uint32_t w) {       // after the prefix sums over wave_tb
    if (w < P.totals->n_waves && w < P.n_waves_cap) P.waves[w].tb_off = P.wave_tb[w];
    if (w == 0) P.totals->tb_dwords = P.wave_tb[P.totals->n_waves < P.n_waves_cap ? P.totals->n_waves : P.n_waves_cap];
}

// Stage 2, per problem, `lane` of `lanes` cooperating callers (a wavefront on the device): the offsets into the shared arenas,
// the node records and predecessor lists of the induced subgraph, its column info stream, its read codes.
// Stage 2 for an extension window: the arenas of its problem from the per-node temporaries stage 1 left (every lane works for itself: nothing
// written here is read here).
VGK_HD void ext_emit_one(const WinParams& P, uint32_t i, uint32_t lane, uint32_t lanes) {
    const uint32_t n1 = P.n + 1;
    const vgk_window_problem p = P.problems[i];
    const WinExt x = P.ext[i];
    ProbDesc& d = P.probs[i];
    const bool left = x.leftward != 0;
    const uint32_t col_off = P.offs[WS_COLS * n1 + i], node_off = P.offs[WS_NODES * n1 + i], pred_at = P.offs[WS_PREDS * n1 + i],
                   read_off = P.offs[WS_READS * n1 + i];
    const uint32_t cnt = P.ext_count[i];
    const bool dummy = cnt == WIN_EXT_DUMMY;
    const uint32_t count = dummy ? 0u : cnt;
    const WinKept* kept = P.kept + x.kept_off;
    const uint32_t a = p.first_node, b = p.first_node + p.n_nodes, s = x.start_node;
    const uint32_t R = d.R, R4 = (R + 3u) & ~3u;
    const bool start_kept = count && kept[0].node == s;
    if (dummy) {
        if (lane == 0) { NodeRec nr; nr.col_start = 0; nr.col_end = 1; nr.pred_begin = pred_at; nr.n_pred = 0; nr.slot = -1; nr.pinning = 0; P.nodes[node_off] = nr; }
    } else {
        for (uint32_t k = lane; k < count; k += lanes) {
            const WinKept kk = kept[k];
            const uint32_t v = kk.node;
            NodeRec nr;
            nr.col_start = kk.col; nr.col_end = (k + 1 < count) ? kept[k + 1].col : R;
            nr.pred_begin = pred_at + kk.pred_at;
            uint32_t np = 0;
            if (v != s) {
                const uint32_t eb = left ? P.g.succ_off[v] : P.g.pred_off[v], ee = left ? P.g.succ_off[v + 1] : P.g.pred_off[v + 1];
                const uint32_t* nb = left ? P.g.succ_idx : P.g.pred_idx;
                for (uint32_t e = eb; e < ee; ++e) {
                    const uint32_t q = nb[e];
                    if (left ? (q > s || q < a) : (q < s || q >= b)) continue;
                    const uint32_t at = (q == s) ? (start_kept ? 0u : 0xffffffffu) : ext_find(kept, k, q, left);
                    if (at != 0xffffffffu) P.preds[nr.pred_begin + np++] = at;
                }
            }
            nr.n_pred = np;
            nr.slot = (kk.slot_flags & 2u) ? (int32_t)(kk.slot_flags >> 2) : -1;
            nr.pinning = 0;
            P.nodes[node_off + k] = nr;
        }
    }
    // the column stream: every column finds its node (the columns of a problem are a few hundred, its nodes a few dozen)
    for (uint32_t c = 4u * lane; c < R4; c += 4u * lanes) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            uint32_t ci = (uint32_t)CI_INVALID;
            const uint32_t cc = c + j;
            if (cc < R) {
                if (dummy) ci = 4u | CI_NODE_START | CI_SEED_SLOW;
                else {
                    uint32_t lo = 0, hi = count;                                       // the last node whose first column is <= cc
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (kept[mid].col <= cc) lo = mid; else hi = mid; }
                    const WinKept kk = kept[lo];
                    const uint32_t v = kk.node, off = cc - kk.col, end = (lo + 1 < count) ? kept[lo + 1].col : R, len = end - kk.col;
                    const uint32_t c0 = P.g.col[v];
                    uint32_t src;
                    if (!left) src = c0 + (v == s ? x.start_offset : 0u) + off;
                    else src = c0 + (v == s ? x.start_offset : P.g.col[v + 1] - c0) - 1u - off;
                    ci = P.g.info[src] & (uint32_t)CI_BASE_MASK;
                    if (off == 0) ci |= CI_NODE_START | ((kk.slot_flags & 1u) ? (uint32_t)CI_SEED_SLOW : 0u);
                    if (off + 1 == len && (kk.slot_flags & 2u)) ci |= CI_STORE_END;
                }
            }
            w |= ci << (8 * j);
        }
        *(uint32_t*)(P.colinfo + col_off + c) = w;
    }
    // read codes: row 0 = "no read base consumed yet", then the read part on the extension's side, in extension order
    const uint32_t rows4 = (d.L + 3u) & ~3u;
    for (uint32_t r = 4u * lane; r < rows4; r += 4u * lanes) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t row = r + j;
            uint32_t code = 0;
            if (row == 0) code = 5;
            else if (row < d.L) code = win_nt_read(P.raw_reads[p.read_off + (left ? x.query_offset - row : x.query_offset + row - 1u)]);
            w |= code << (8 * j);
        }
        *(uint32_t*)(P.reads + read_off + r) = w;
    }
    if (lane == 0) {
        d.col_off = col_off; d.node_off = node_off; d.read_off = read_off;
        d.scratch_off = P.offs[WS_SCRATCH * n1 + i]; d.ops_off = P.offs[WS_OPS * n1 + i];
    }
}

VGK_HD void win_emit_one(const WinParams& P, uint32_t i, uint32_t lane, uint32_t lanes) {
    if (P.ext) { ext_emit_one(P, i, lane, lanes); return; }
    const uint32_t n1 = P.n + 1;
    const vgk_window_problem p = P.problems[i];
    ProbDesc& d = P.probs[i];
    const uint32_t first = p.first_node, nn = d.n_nodes;
    const uint32_t col_off = P.offs[WS_COLS * n1 + i], node_off = P.offs[WS_NODES * n1 + i], pred_at = P.offs[WS_PREDS * n1 + i],
                   read_off = P.offs[WS_READS * n1 + i];
    const bool xdrop = (d.flags & 15u) == VGK_XDROP_PINNED;
    const uint32_t col0 = P.g.col[first], slot0 = P.g.slot[first], pred0 = P.g.pred_off[first];
    // node records + predecessors inside the window (those before it are not part of the induced subgraph)
    for (uint32_t k = lane; k < nn; k += lanes) {
        const uint32_t v = first + k;
        NodeRec nr;
        nr.col_start = P.g.col[v] - col0; nr.col_end = P.g.col[v + 1] - col0;
        const uint32_t pb = P.g.pred_off[v], pe = P.g.pred_off[v + 1];
        nr.pred_begin = pred_at + (pb - pred0);
        uint32_t np = 0;
        for (uint32_t e = pb; e < pe; ++e) { const uint32_t q = P.g.pred_idx[e]; if (q >= first) P.preds[nr.pred_begin + np++] = q - first; }
        nr.n_pred = np;
        nr.slot = (P.g.slot[v + 1] != P.g.slot[v]) ? (int32_t)(P.g.slot[v] - slot0) : -1;
        nr.pinning = 0;
        P.nodes[node_off + k] = nr;
    }
    // column info: the resident bytes, four columns per lane and store (the arena is padded: reading a few bytes past the window
    // is safe); the window's first column starts a source node (gssw: fresh registers; dozeu: the root column)
    const uint32_t R = d.R, R4 = (R + 3u) & ~3u;
    for (uint32_t c = 4u * lane; c < R4; c += 4u * lanes) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            uint32_t ci = (uint32_t)CI_INVALID;
            if (c + j < R) {
                ci = P.g.info[col0 + c + j];
                if (c + j == 0) ci = (ci & ~(uint32_t)CI_SEED_SLOW) | CI_NODE_START | (xdrop ? (uint32_t)CI_SEED_SLOW : 0u);
            }
            w |= ci << (8 * j);
        }
        *(uint32_t*)(P.colinfo + col_off + c) = w;         // col_off is a multiple of 4
    }
    // read codes, four per lane and store; X-drop problems get row 0 = "no read base consumed yet"
    const uint32_t lead = xdrop ? 1u : 0u, rows4 = (d.L + 3u) & ~3u;
    for (uint32_t r = 4u * lane; r < rows4; r += 4u * lanes) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t row = r + j;
            uint32_t code = 0;
            if (row < lead) code = 5;
            else if (row < d.L) code = win_nt_read(P.raw_reads[p.read_off + row - lead]);
            w |= code << (8 * j);
        }
        *(uint32_t*)(P.reads + read_off + r) = w;          // read_off is a multiple of 4 (sizes are padded)
    }
    if (lane == 0) {
        d.col_off = col_off; d.node_off = node_off; d.read_off = read_off;
        d.scratch_off = P.offs[WS_SCRATCH * n1 + i]; d.ops_off = P.offs[WS_OPS * n1 + i];
    }
}

}  // namespace vgk
