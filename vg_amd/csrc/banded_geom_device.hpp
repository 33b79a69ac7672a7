// banded_geom_device.hpp — the band geometry of a batch of banded global alignments, one lane per problem, over the RAW graph arrays
// (replaces, for graphs without empty nodes, what banded_api.cpp's prepare() does on host threads: find_banded_paths
// src/banded_global_aligner.cpp:2174-2268, path_lengths_to_sinks :2122-2170, shortest_seq_paths :2271-2293, and the tables the fill and
// traceback kernels of banded_device.hpp read: node records with their flattened predecessor lists, candidate end nodes, arena offsets).
//
// Why on the device: the tables are 2.1 kB per problem (a BNode is 64 bytes) against 0.8 kB of graph arrays, reads and bases — the host only
// gathers the latter, and the 1 µs per problem and thread that prepare() takes is a lane's work of a few hundred dependent steps here.
//
// Only graphs WITHOUT empty nodes: there a node's flattened predecessors are its unmasked predecessors (no paths through empty nodes, no
// pool), the candidate end nodes are the unmasked sinks (no prefixes), and there is no source-to-sink walk of empty nodes.  A call with an
// empty node anywhere takes the host path.  The host has checked the arrays (offsets ascending, predecessors before their node, lengths
// <= 65535) while it gathered them.
//
// The same code is stepped on the CPU by tests/emu — test infrastructure only.
#pragma once
#include <stdint.h>
#include "banded_device.hpp"

namespace vgk {

struct BGeomProb {                 // one problem as the host gathered it
    uint32_t node_base;            // its node_len[] / BNode[] / BStart[] / tmp[] slots; pred_off[] lies at node_base + (problem index): n_nodes + 1 entries, counted from the problem's first edge
    uint32_t edge_base;            // its pred_idx[] / BSeed[] slots
    uint32_t n_nodes, L;
    uint32_t band_padding, permissive;
    unsigned long long max_cells;  // 0: no limit
};
struct BGeomOut {                  // what the host needs back to place the problem
    int32_t  status;               // VGK_OK, VGK_ETOOBIG, VGK_ENOBAND
    uint32_t R;                    // rows per lane (Hpad = 64 R)
    uint32_t order_key;
    uint32_t n_seeds, n_starts, last_elems;
    unsigned long long cells, tb_bytes;
};
struct BGeomParams {
    const BGeomProb* probs; uint32_t n;
    const uint32_t* node_len; const uint32_t* pred_off; const uint32_t* pred_idx;
    int32_t* tmp;                  // 3 per node: shortest, longest, has-successor
    BNode* nodes; BSeed* seeds; BStart* starts;
    BGeomOut* out;
};

VGK_HD void banded_geometry_one(const BGeomParams& P, uint32_t i) {
    const BGeomProb pb = P.probs[i];
    const uint32_t N = pb.n_nodes; const int64_t L = pb.L;
    const uint32_t* len = P.node_len + pb.node_base;
    const uint32_t* poff = P.pred_off + pb.node_base + i;
    const uint32_t* pidx = P.pred_idx + pb.edge_base;
    int32_t* shortest = P.tmp + 3ull * pb.node_base; int32_t* longest = shortest + N; int32_t* has_succ = longest + N;
    BNode* rec = P.nodes + pb.node_base;
    BGeomOut out; out.status = VGK_OK; out.R = 1; out.order_key = 0; out.n_seeds = 0; out.n_starts = 0; out.last_elems = 0; out.cells = 0; out.tb_bytes = 0;
    constexpr int32_t INF = 0x3fffffff;
    for (uint32_t v = 0; v < N; ++v) { has_succ[v] = 0; longest[v] = 0; }
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = poff[v]; e < poff[v + 1]; ++e) has_succ[pidx[e]] = 1;
    for (uint32_t v = 0; v < N; ++v) shortest[v] = has_succ[v] ? INF : 0;
    for (uint32_t v = N; v-- > 0;) {
        const int32_t lv = (int32_t)len[v], lo = longest[v] + lv, sh = shortest[v] + lv;
        for (uint32_t e = poff[v]; e < poff[v + 1]; ++e) {
            const uint32_t u = pidx[e];
            if (lo > longest[u]) longest[u] = lo;
            if (sh < shortest[u]) shortest[u] = sh;
        }
    }
    // bands, top to bottom: a node's band is the union of what its unmasked predecessors hand on (:2198-2262); the shortest sequence from
    // a source to its left edge beside it (:2271-2293, over every predecessor)
    const int64_t pad = pb.band_padding;
    unsigned long long cells = 0; int64_t max_h = 0;
    for (uint32_t v = 0; v < N; ++v) {
        BNode nd{};
        const int64_t lv = len[v];
        int64_t top = INF, bot = -(int64_t)INF, cum = INF;
        if (poff[v] == poff[v + 1]) {
            if (pb.permissive) {
                const int64_t a = L - (lv + longest[v]) - pad, b = L - (lv + shortest[v]) + pad;
                top = -pad < a ? -pad : a; bot = pad > b ? pad : b;
            } else { top = -pad; bot = pad; }
            cum = 0;
        } else for (uint32_t e = poff[v]; e < poff[v + 1]; ++e) {
            const BNode& pu = rec[pidx[e]];
            const int64_t c = (int64_t)pu.cum + pu.len; if (c < cum) cum = c;
            if (pu.masked) continue;
            const int64_t et = (int64_t)pu.top + pu.len, eb = (int64_t)pu.bot + pu.len;
            if (et < top) top = et;
            if (eb > bot) bot = eb;
        }
        bool masked = top > bot;
        if (!masked) masked = top + lv + shortest[v] > L || bot + lv + longest[v] < L;
        nd.len = (int32_t)lv; nd.cum = (int32_t)cum; nd.masked = masked ? 1 : 0;
        nd.top = masked ? 0 : (int32_t)top; nd.bot = masked ? -1 : (int32_t)bot;
        if (!masked) { cells += (unsigned long long)(bot - top + 1) * (unsigned long long)lv; if (bot - top + 1 > max_h) max_h = bot - top + 1; }
        rec[v] = nd;
    }
    // (a masked node's cum is kept while the bands are made — its successors read it — and cleared with the records below, as prepare() stores it)
    out.cells = cells;
    if (pb.max_cells && cells > pb.max_cells) { out.status = VGK_ETOOBIG; P.out[i] = out; return; }
    if (!pb.permissive) {
        bool any = false;
        for (uint32_t v = 0; v < N; ++v) if (!has_succ[v] && !rec[v].masked) any = true;
        if (!any) { out.status = VGK_ENOBAND; P.out[i] = out; return; }
    }
    uint32_t R = 1; while ((int64_t)R * 64 < max_h) R *= 2;
    if (R > B_MAX_ROWS_PER_LANE) { out.status = VGK_ETOOBIG; P.out[i] = out; return; }
    out.R = R;
    { uint32_t r = 0; while ((1u << r) < R) ++r;
      uint32_t lg = 0; while ((cells >> lg) > 1 && lg < 63) ++lg;
      out.order_key = r * 64 + (63 - lg); }
    // node records: flattened predecessors (the reference pops them off a stack: last predecessor first), chains, arena offsets
    unsigned long long tb_off = 0; uint32_t seq_off = 0, n_seeds = 0; int64_t prev_filled = -1;
    BSeed* seeds = P.seeds + pb.edge_base;
    for (uint32_t v = 0; v < N; ++v) {
        BNode nd = rec[v];
        const int32_t top = nd.top, bot = nd.bot;
        if (nd.masked) nd.cum = 0;
        nd.seq_off = seq_off; seq_off += (uint32_t)nd.len;
        nd.seed_off = n_seeds;
        if (!nd.masked) {
            nd.as_source = poff[v] == poff[v + 1] ? 1 : 0;
            uint32_t mine = 0;
            for (uint32_t e = poff[v + 1]; e-- > poff[v];) {
                const uint32_t u = pidx[e];
                if (rec[u].masked) continue;
                BSeed sd; sd.node = u; sd.path_off = 0; sd.path_len = 0;
                seeds[n_seeds++] = sd; ++mine;
            }
            if (mine > 0xffffu) { out.status = VGK_ETOOBIG; P.out[i] = out; return; }
            nd.n_seeds = (uint16_t)mine;
            if (mine == 1 && !nd.as_source) {
                const uint32_t u = seeds[n_seeds - 1].node;
                nd.chain = (int64_t)u == prev_filled && top == rec[u].top + rec[u].len && bot == rec[u].bot + rec[u].len ? 1u : 0u;
            }
            prev_filled = v;
            if (!nd.chain) for (uint32_t q = 0; q < mine; ++q) rec[seeds[n_seeds - 1 - q].node].keep_last = 1;
            const uint32_t granule = R > 4 ? R : 4, H = (uint32_t)(bot - top + 1);
            nd.stride = (H + granule - 1) / granule * granule;
            nd.tb_off = (uint32_t)tb_off;
            tb_off += (unsigned long long)nd.len * nd.stride;
            if (tb_off > 0xfffffff0ull) { out.status = VGK_ETOOBIG; P.out[i] = out; return; }
        }
        nd.keep_last = 0;                                  // (set by the nodes behind this one, below and in the loop's later turns)
        rec[v] = nd;
    }
    // where a traceback may start: every unmasked sink, in order (:2442-2556 without empty sinks)
    BStart* starts = P.starts + pb.node_base; uint32_t n_starts = 0;
    for (uint32_t v = 0; v < N; ++v) if (!has_succ[v] && !rec[v].masked) { starts[n_starts++].node = v; rec[v].keep_last = 1; }
    // last / first columns only where a traceback or a successor will read them
    uint32_t last_off = 0;
    for (uint32_t v = 0; v < N; ++v) {
        BNode& nd = rec[v];
        if (nd.masked || (nd.chain && !nd.keep_last)) continue;
        nd.last_off = last_off; last_off += 5u * nd.stride;
    }
    out.n_seeds = n_seeds; out.n_starts = n_starts; out.last_elems = last_off; out.tb_bytes = (tb_off + 255) & ~255ull;
    P.out[i] = out;
}

}  // namespace vgk
