// banded_api.cpp — vgk_banded_align: the host half of banded global alignment.
//
// What BandedGlobalAligner's constructor does per call on the CPU (reference:
// src/banded_global_aligner.cpp:1961-2110 — band ends, masking, cell budget, shortest lead
// sequences, one BAMatrix per node with its seed pointers) becomes flat tables for the wavefront
// kernels in banded_device.hpp; the predecessor lists are flattened here, once, in the LIFO order the
// reference's fill and traceback pop them, with the empty nodes each one is reached through.
//
// A batch is prepared in three passes: (1) per-problem geometry and tables, in parallel over host
// threads; (2) prefix sums that place every problem in the shared arenas and cut the batch into
// sub-batches that fit the device budget; (3) parallel copy into the arenas.  Device scratch (traceback
// bytes, last columns, op slots) is cached on the context between calls.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>
#include <emmintrin.h>
#include "ctx.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {

inline uint8_t nt_code(char ch) {      // gssw_create_nt_table: case-insensitive ACGT, everything else N
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct Span { uint64_t off = 0; uint32_t len = 0; };
struct Prep {                          // one problem after pass 1; its tables live in the preparing thread's Store
    int status = VGK_OK;
    uint32_t order_key = 0;
    bool on_device = false;
    uint32_t R = 1, Hpad = 64, thread = 0;
    uint64_t cells = 0, bases = 0, tb_bytes = 0, last_elems = 0, in_bytes = 0;      // in_bytes: the problem's share of the call's algorithmic bytes (all but its ops)
    uint32_t ops_cap = 0;
    Span nodes, seeds, pool, starts;   // starts: candidate end nodes; start_prefix[k] = the empty sink-side nodes of candidate k, sink first (:2455-2480)
    bool have_empty_walk = false;      // a source-to-sink chain of empty nodes (:2464-2472)
    Span empty_walk;                   // in Store::prefix, sink first (the first chain found)
    std::vector<Span> empty_walks;     // every chain, for the k-best mode (:2464-2472)
    uint32_t arena = 0;                // index inside its sub-batch
    uint32_t need = 0;                 // ops this problem hands back
    bool use_empty_walk = false;
};
// (Store and Scratch are written by one thread each while its neighbours in the array write theirs: a cache line of their own, or every
// push_back's size update bounces the line between cores)
struct alignas(128) Store {           // per-thread flat tables of pass 1
    std::vector<BNode> nodes; std::vector<BSeed> seeds; std::vector<uint32_t> pool, starts, prefix;
    std::vector<Span> start_prefix;    // parallel to starts
    void clear() { nodes.clear(); seeds.clear(); pool.clear(); starts.clear(); prefix.clear(); start_prefix.clear(); }
};

struct alignas(128) Scratch {         // per-thread reusable buffers of pass 1
    std::vector<int64_t> len, shortest, longest, top, bot, cum;
    std::vector<uint8_t> masked;
    std::vector<uint32_t> succ_off, succ, fill;
    struct Item { uint32_t node, path_off, path_len; };
    std::vector<Item> stack;
    std::vector<int64_t> st; std::vector<uint32_t> path;
};

// Band geometry of one problem (find_banded_paths :2174-2268, path_lengths_to_sinks :2122-2170, shortest_seq_paths :2271-2293)
// and the tables the kernels need.
// `out` (nullable): where the N node records go instead of T.nodes (the pipelined call writes them where they are uploaded from)
void prepare(const vgk_ctx* ctx, const vgk_banded_problem& p, Prep& hp, Scratch& S, Store& T, BNode* out = nullptr) {
    const vgk_graph& g = p.graph;
    const uint32_t N = g.n_nodes; const int64_t L = p.read_len;
    if (!N || !L || !p.read || !g.node_len || !g.pred_off || !g.seq || (ctx->has_qa && !p.qual)) { hp.status = VGK_EINVAL; return; }
    if (g.pred_off[N] > g.pred_off[0] && !g.pred_idx) { hp.status = VGK_EINVAL; return; }
    for (uint32_t v = 0; v < N; ++v) {
        if (g.pred_off[v + 1] < g.pred_off[v]) { hp.status = VGK_EINVAL; return; }                 // offsets must not decrease (they size the successor lists)
        for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) if (g.pred_idx[e] >= v) { hp.status = VGK_EINVAL; return; }
        // vgk_op.len is 16 bits: one run (a match inside a node, the insertion of a whole read on an empty walk) must fit.  vg chops
        // nodes to <= 1024 bp and calls this aligner on stretches between anchors; the caller keeps its CPU path beyond.
        if (g.node_len[v] > 65535u) { hp.status = VGK_ETOOBIG; return; }
    }
    if (L > 65535) { hp.status = VGK_ETOOBIG; return; }
    const int64_t inf = std::numeric_limits<int64_t>::max();
    S.succ_off.assign(N + 1, 0);
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) ++S.succ_off[g.pred_idx[e] + 1];
    for (uint32_t v = 0; v < N; ++v) S.succ_off[v + 1] += S.succ_off[v];
    S.succ.resize(S.succ_off[N] + 1);
    S.fill.assign(S.succ_off.begin(), S.succ_off.end() - 1);
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) S.succ[S.fill[g.pred_idx[e]]++] = v;
    auto is_source = [&](uint32_t v) { return g.pred_off[v] == g.pred_off[v + 1]; };
    auto is_sink = [&](uint32_t v) { return S.succ_off[v] == S.succ_off[v + 1]; };
    S.len.resize(N); S.shortest.resize(N); S.longest.assign(N, 0); S.top.assign(N, inf); S.bot.assign(N, std::numeric_limits<int64_t>::min());
    S.cum.resize(N); S.masked.assign(N, 0);
    auto &len = S.len, &shortest = S.shortest, &longest = S.longest, &top = S.top, &bot = S.bot, &cum = S.cum; auto& masked = S.masked;
    uint64_t total_bases = 0;
    for (uint32_t v = 0; v < N; ++v) { len[v] = g.node_len[v]; total_bases += g.node_len[v]; shortest[v] = is_sink(v) ? 0 : inf; }
    hp.bases = total_bases;
    for (uint32_t v = N; v-- > 0;)
        for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) {
            const uint32_t u = g.pred_idx[e];
            longest[u] = std::max(longest[u], longest[v] + len[v]);
            shortest[u] = std::min(shortest[u], shortest[v] + len[v]);
        }
    const bool permissive = (p.flags & VGK_BANDED_PERMISSIVE) != 0; const int64_t pad = p.band_padding;
    for (uint32_t v = 0; v < N; ++v) if (is_source(v)) {
        if (permissive) {
            top[v] = std::min<int64_t>(-pad, L - (len[v] + longest[v]) - pad);
            bot[v] = std::max<int64_t>(pad, L - (len[v] + shortest[v]) + pad);
        } else { top[v] = -pad; bot[v] = pad; }
    }
    uint64_t cells = 0; int64_t max_h = 0;
    for (uint32_t v = 0; v < N; ++v) {
        if (top[v] > bot[v]) { masked[v] = 1; continue; }                  // no unmasked walk reaches it
        const int64_t et = top[v] + len[v], eb = bot[v] + len[v];
        if (et + shortest[v] > L || eb + longest[v] < L) { masked[v] = 1; continue; }
        for (uint32_t e = S.succ_off[v]; e < S.succ_off[v + 1]; ++e) { const uint32_t w = S.succ[e]; top[w] = std::min(top[w], et); bot[w] = std::max(bot[w], eb); }
        cells += (uint64_t)(bot[v] - top[v] + 1) * (uint64_t)len[v];
        if (len[v]) max_h = std::max(max_h, bot[v] - top[v] + 1);
    }
    hp.cells = cells;
    if (p.max_cells && cells > p.max_cells) { hp.status = VGK_ETOOBIG; return; }
    for (uint32_t v = 0; v < N; ++v) cum[v] = is_source(v) ? 0 : inf;
    for (uint32_t v = 0; v < N; ++v) {
        if (cum[v] == inf) continue;
        for (uint32_t e = S.succ_off[v]; e < S.succ_off[v + 1]; ++e) cum[S.succ[e]] = std::min(cum[S.succ[e]], cum[v] + len[v]);
    }
    if (!permissive) {
        bool any = false;
        for (uint32_t v = 0; v < N; ++v) if (is_sink(v) && !masked[v]) any = true;
        if (!any) { hp.status = VGK_ENOBAND; return; }
    }
    uint32_t R = 1; while ((int64_t)R * 64 < max_h) R *= 2;
    if (R > B_MAX_ROWS_PER_LANE || L > (1 << 24) || total_bases > (1u << 24)) { hp.status = VGK_ETOOBIG; return; }      // engine limit: bands up to 32 768 diagonals (R = 512 rows per lane: 64 blocks of 8, their state in HBM beyond 4 blocks)
    hp.R = R; hp.Hpad = 64 * R;
    { uint32_t r = 0; while ((1u << r) < R) ++r;                  // launch order: rows-per-lane class, then most cells first (by log2)
      uint32_t lg = 0; while ((hp.cells >> lg) > 1 && lg < 63) ++lg;
      hp.order_key = r * 64 + (63 - lg); }

    // node records with the flattened predecessor lists
    uint64_t tb_off = 0, last_off = 0; uint32_t seq_off = 0;
    int64_t prev_filled = -1;              // the node the wave will have in its registers when it reaches v
    const size_t keep_nodes = T.nodes.size(), keep_seeds = T.seeds.size(), keep_pool = T.pool.size();
    auto fail = [&](int code) { if (!out) T.nodes.resize(keep_nodes); T.seeds.resize(keep_seeds); T.pool.resize(keep_pool); hp.status = code; };
    if (!out) T.nodes.resize(keep_nodes + N);
    BNode* const recs = out ? out : T.nodes.data() + keep_nodes;
    for (uint32_t v = 0; v < N; ++v) {
        BNode nd{};
        nd.top = masked[v] ? 0 : (int32_t)top[v]; nd.bot = masked[v] ? -1 : (int32_t)bot[v];
        nd.len = (int32_t)len[v]; nd.cum = masked[v] || cum[v] == inf ? 0 : (int32_t)cum[v];
        nd.seq_off = seq_off; seq_off += (uint32_t)len[v];
        nd.masked = masked[v];
        nd.seed_off = (uint32_t)(T.seeds.size() - keep_seeds);
        if (!masked[v] && len[v]) {
            nd.as_source = is_source(v);
            S.stack.clear();
            for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) S.stack.push_back({g.pred_idx[e], 0, 0});
            uint32_t n_seeds = 0;
            while (!S.stack.empty()) {
                const Scratch::Item it = S.stack.back(); S.stack.pop_back();
                if (masked[it.node]) continue;
                if (len[it.node] == 0) {
                    const uint32_t noff = (uint32_t)(T.pool.size() - keep_pool);
                    for (uint32_t q = 0; q < it.path_len; ++q) T.pool.push_back(T.pool[keep_pool + it.path_off + q]);
                    T.pool.push_back(it.node);
                    if (is_source(it.node)) { nd.as_source = 1; nd.src_path_off = noff; nd.src_path_len = it.path_len + 1; }
                    for (uint32_t e = g.pred_off[it.node]; e < g.pred_off[it.node + 1]; ++e) S.stack.push_back({g.pred_idx[e], noff, it.path_len + 1});
                    continue;
                }
                T.seeds.push_back({it.node, it.path_off, it.path_len}); ++n_seeds;
            }
            if (n_seeds > 0xffff) { fail(VGK_ETOOBIG); return; }
            nd.n_seeds = (uint16_t)n_seeds;
            if (n_seeds == 1 && !nd.as_source) {
                const BSeed& sd = T.seeds.back();
                nd.chain = sd.path_len == 0 && (int64_t)sd.node == prev_filled && top[v] == top[sd.node] + len[sd.node] && bot[v] == bot[sd.node] + len[sd.node];
            }
            prev_filled = v;
            if (!nd.chain) for (uint32_t q = 0; q < n_seeds; ++q) recs[T.seeds[T.seeds.size() - 1 - q].node].keep_last = 1;
            const uint32_t granule = std::max<uint32_t>(R, 4), H = (uint32_t)(bot[v] - top[v] + 1);
            nd.stride = (H + granule - 1) / granule * granule;
            nd.tb_off = (uint32_t)tb_off;
            tb_off += (uint64_t)len[v] * nd.stride;
            if (tb_off > 0xfffffff0ull) { fail(VGK_ETOOBIG); return; }
        }
        recs[v] = nd;
    }
    hp.nodes = {out ? 0 : keep_nodes, N}; hp.seeds = {keep_seeds, (uint32_t)(T.seeds.size() - keep_seeds)}; hp.pool = {keep_pool, (uint32_t)(T.pool.size() - keep_pool)};
    hp.starts.off = T.starts.size();
    // where a traceback may start (:2442-2556): every sink in topological order (PARITY-UNPINNED: the reference iterates an
    // unordered_set of matrix pointers), looking through empty sinks to their predecessors depth-first, last predecessor first
    for (uint32_t v = 0; v < N; ++v) {
        if (!is_sink(v) || masked[v]) continue;
        S.st.assign(1, v); S.path.clear();
        while (!S.st.empty()) {
            const int64_t u = S.st.back(); S.st.pop_back();
            if (u < 0) { S.path.pop_back(); continue; }
            if (masked[u]) continue;
            if (len[u] == 0) {
                S.path.push_back((uint32_t)u); S.st.push_back(-1);
                if (is_source((uint32_t)u)) {
                    hp.empty_walks.push_back({T.prefix.size(), (uint32_t)S.path.size()});
                    if (!hp.have_empty_walk) { hp.have_empty_walk = true; hp.empty_walk = hp.empty_walks.back(); }
                    T.prefix.insert(T.prefix.end(), S.path.begin(), S.path.end());
                    continue;
                }
                for (uint32_t e = g.pred_off[u]; e < g.pred_off[u + 1]; ++e) S.st.push_back(g.pred_idx[e]);
                continue;
            }
            T.starts.push_back((uint32_t)u); T.start_prefix.push_back({T.prefix.size(), (uint32_t)S.path.size()});
            T.prefix.insert(T.prefix.end(), S.path.begin(), S.path.end());
        }
    }
    hp.starts.len = (uint32_t)(T.starts.size() - hp.starts.off);
    for (uint32_t q = 0; q < hp.starts.len; ++q) recs[T.starts[hp.starts.off + q]].keep_last = 1;
    // last / first columns only where a traceback or a successor will read them
    for (uint32_t v = 0; v < N; ++v) {
        BNode& nd = recs[v];
        if (nd.masked || nd.len == 0 || (nd.chain && !nd.keep_last)) continue;
        nd.last_off = (uint32_t)last_off; last_off += 5ull * nd.stride;
    }
    hp.tb_bytes = (tb_off + 255) & ~255ull; hp.last_elems = last_off;
    hp.ops_cap = (uint32_t)(L + total_bases + 2ull * N + 8);
    hp.in_bytes = p.read_len + hp.bases + 8ull * N + 4ull * g.pred_off[N] + hp.cells + 16;       // (here, on the preparing thread: the serial loop over the results would miss the cache once per problem for pred_off[N])
    hp.on_device = true;
}

// device scratch cached on the context (grow-only)
inline void* ensure(vgk_ctx* ctx, int slot, uint64_t bytes) { return ctx->ensure_scratch(slot, bytes); }
template <class T> int stage(vgk_ctx* ctx, int slot, const T* v, size_t count, const T*& out) {
    void* d = ensure(ctx, slot, std::max<size_t>(count, 1) * sizeof(T));
    if (!d) return VGK_ENOMEM;
    if (count) { int rc = ctx->be->upload(d, v, count * sizeof(T)); if (rc) return rc; }
    out = (const T*)d;
    return VGK_OK;
}
// host arenas kept on the context between calls: uninitialised storage, so a warm call neither zero-fills nor page-faults
struct PinnedSet {
    PinnedBuf<BProb> probs, probs2; PinnedBuf<BNode> nodes; PinnedBuf<BSeed> seeds; PinnedBuf<uint32_t> pool, order; PinnedBuf<BStart> starts;
    PinnedBuf<uint8_t> reads, quals, graph; PinnedBuf<BResult> dres; PinnedBuf<vgk_op> dops; PinnedBuf<unsigned long long> count;
};
struct HostArenas {
    std::vector<Scratch> scratch; std::vector<Store> store;       // pass 1's per-thread tables: their memory stays mapped between calls
    PinnedBuf<BProb> probs; PinnedBuf<BNode> nodes; PinnedBuf<BSeed> seeds; PinnedBuf<uint32_t> pool, order; PinnedBuf<BStart> starts;
    PinnedBuf<uint8_t> reads, quals, graph; PinnedBuf<BResult> dres; PinnedBuf<vgk_op> dops; PinnedBuf<int32_t> scores;
    PinnedSet set[2]; void* ev[2] = {nullptr, nullptr}; Backend* be = nullptr;      // the pipelined path: two sub-batches in flight
    PinnedBuf<BNode> qnodes[2];        // ... and the node records of a quarter of the call, written by prepare() where they are uploaded from (quarters alternate)
    struct GeomSet {                   // the device-geometry path (banded_align_device_geometry): the raw arrays of a sub-batch, and what comes back
        PinnedBuf<BGeomProb> gprobs; PinnedBuf<uint32_t> node_len, pred_off, pred_idx; PinnedBuf<BGeomOut> gout;
    } gset[2];
    ~HostArenas() { if (be) for (void* e : ev) if (e) be->event_destroy(e); }
};
enum { S_PROBS, S_ORDER, S_NODES, S_SEEDS, S_POOL, S_STARTS, S_READS, S_QUALS, S_GRAPH, S_MAT, S_TB, S_LAST, S_OPS, S_DENSE, S_RESULTS, S_COUNT };
constexpr int S_SCORES = 31;          // k-best mode: the full score matrices (slots 15..30 belong to gapless_api.cpp)
constexpr int S_SET1 = 124;           // the pipelined path's second sub-batch in flight: slots S_SET1 + S_*
enum { G_PROBS, G_NODELEN, G_PREDOFF, G_PREDIDX, G_TMP, G_OUT, G_COUNT };
constexpr int G_SET0 = 140, G_SET1 = 146;     // the device-geometry path's raw arrays, per sub-batch in flight
constexpr int BANDED_NOT_HERE = 10000;        // banded_align_device_geometry: this call is for the host-geometry path (never leaves the library)
static_assert(S_COUNT <= 15 && S_SCORES < (int)(sizeof(vgk_ctx::scratch) / sizeof(vgk_ctx::DevBuf)) && S_SET1 + S_COUNT <= (int)(sizeof(vgk_ctx::scratch) / sizeof(vgk_ctx::DevBuf)), "scratch slots");



// ---- k-best alignments: the alternate-traceback stack of the reference (AltTracebackStack, src/banded_global_aligner.cpp:2426-2790)
// walked on the host over the score matrices the fill kernel left in HBM.  A traceback is the list of its deflections — the first
// names the start, every later one a cell where it leaves the optimal choice for a named predecessor state; the stack keeps up to
// `max` of them in descending score order, equal scores in the order they were found.
struct Defl { int32_t from_node; int64_t r, j; int32_t to_node, to_mat; };
struct Trace { std::vector<Defl> d; int32_t score; Span prefix; };

struct MultiTracer {
    const vgk_ctx* ctx; const vgk_banded_problem& p; const Prep& hp; const Store& T; const int32_t* sc;
    std::vector<Trace> stack; size_t cur = 0, cur_defl = 0; int64_t max = 1;
    std::vector<vgk_op> runs;                      // back to front
    int32_t go, ge; int64_t L;

    const BNode& nd(int32_t v) const { return T.nodes[hp.nodes.off + v]; }
    static bool live(int32_t v) { return v > BNEG / 2; }
    int32_t val(int32_t v, uint32_t state, int64_t r, int64_t j) const {
        const BNode& n = nd(v);
        return sc[3 * ((size_t)n.tb_off + (size_t)j * n.stride) + (size_t)state * n.stride + (size_t)(r - j - n.top)];
    }
    int32_t sub(int32_t v, int64_t r, int64_t j) const {
        const uint32_t g = nt_code(p.graph.seq[nd(v).seq_off + j]), q = nt_code(p.read[r]);
        return ctx->has_qa ? ctx->qmat[25u * p.qual[r] + 5u * g + q] : ctx->sc.matrix[5u * g + q];
    }
    void emit(int32_t node, uint32_t op, uint32_t inc) {                       // BABuilder (:44-100)
        if (!runs.empty() && runs.back().node == (uint32_t)node) {
            vgk_op& c = runs.back();
            if (c.op == op) { c.len = (uint16_t)(c.len + inc); return; }
            if (c.len == 0 && nd(node).len == 0) { c.op = (uint8_t)op; c.len = (uint16_t)inc; return; }
        }
        vgk_op c{}; c.node = (uint32_t)node; c.op = (uint8_t)op; c.len = (uint16_t)inc; runs.push_back(c);
    }
    static uint32_t op_of(uint32_t mat) { return mat == BM ? VGK_OP_M : mat == BIR ? VGK_OP_I : VGK_OP_D; }
    void insert(const std::vector<Defl>& prefix_d, int32_t score, const Defl& last, Span prefix) {          // insert_traceback (:2691-2740)
        size_t pos = stack.size();
        while (pos > 0 && score > stack[pos - 1].score) --pos;
        if (!stack.empty() && pos == stack.size() && (int64_t)stack.size() >= max) return;
        Trace t; t.d = prefix_d; t.d.push_back(last); t.score = score; t.prefix = prefix;
        stack.insert(stack.begin() + pos, std::move(t));
        if ((int64_t)stack.size() > max) stack.pop_back();
    }
    void propose(int32_t alt, int32_t from_node, int64_t r, int64_t j, int32_t to_node, uint32_t to_mat) {  // propose_deflection (:2671-2689)
        if (cur_defl != stack[cur].d.size()) return;
        if (alt <= stack.back().score && (int64_t)stack.size() >= max) return;
        const std::vector<Defl> d = stack[cur].d; const Span pre = stack[cur].prefix;
        insert(d, alt, Defl{from_node, r, j, to_node, (int32_t)to_mat}, pre);
    }
    bool at_deflection(int32_t node, int64_t r, int64_t j) const {
        const Trace& c = stack[cur];
        return cur_defl < c.d.size() && c.d[cur_defl].from_node == node && c.d[cur_defl].r == r && c.d[cur_defl].j == j;
    }
    // the three predecessor states in the reference's order; the first that explains `cur` is taken, every other live one proposed
    int pick(int32_t v, int64_t r, int64_t j, int32_t cur_val, int32_t dm, int32_t dc, int32_t dr, int32_t from_node, int64_t fr, int64_t fj) {
        const int32_t S = stack[cur].score;
        int found = -1;
        { const int32_t src = val(v, BM, r, j), diff = cur_val - (src + dm);
          if (diff == 0) found = BM; else if (live(src)) propose(S - diff, from_node, fr, fj, from_node, BM); }
        { const int32_t src = val(v, BIC, r, j); if (live(src)) { const int32_t diff = cur_val - (src + dc);
          if (found < 0 && diff == 0) found = BIC; else propose(S - diff, from_node, fr, fj, from_node, BIC); } }
        { const int32_t src = val(v, BIR, r, j); if (live(src)) { const int32_t diff = cur_val - (src + dr);
          if (found < 0 && diff == 0) found = BIR; else propose(S - diff, from_node, fr, fj, from_node, BIR); } }
        return found;
    }
    // where a deflection across an edge lands: predecessors in their own order, each through its empty nodes depth-first (:1163-1196)
    bool find_deflect_seed(int32_t node, int32_t target, std::vector<int32_t>& path) const {
        const vgk_graph& g = p.graph;
        std::vector<int64_t> st;
        for (uint32_t e0 = g.pred_off[node]; e0 < g.pred_off[node + 1]; ++e0) {
            st.assign(1, g.pred_idx[e0]); path.clear();
            while (!st.empty()) {
                const int64_t s = st.back(); st.pop_back();
                if (s < 0) { path.pop_back(); continue; }
                if (nd((int32_t)s).masked) continue;
                if (s == target) return true;
                if (nd((int32_t)s).len == 0) {
                    path.push_back((int32_t)s); st.push_back(-1);
                    for (uint32_t e = g.pred_off[s]; e < g.pred_off[s + 1]; ++e) st.push_back(g.pred_idx[e]);
                }
            }
        }
        return false;
    }
    // one traceback (BAMatrix::traceback :756-1126 + traceback_over_edge :1129-1780), following stack[cur]'s deflections
    int trace() {
        const int32_t S = stack[cur].score;
        int32_t node = stack[cur].d[0].from_node; uint32_t mat = (uint32_t)stack[cur].d[0].to_mat;
        int64_t r = L - 1, j = nd(node).len - 1;
        bool lead = false;
        std::vector<int32_t> dpath;
        for (;;) {
            const BNode& n = nd(node);
            while ((j > 0 || mat == BIR) && !lead) {
                emit(node, op_of(mat), 1);
                if (at_deflection(node, r, j)) {
                    if (mat == BM) { --r; --j; } else if (mat == BIR) --r; else --j;
                    mat = (uint32_t)stack[cur].d[cur_defl++].to_mat;
                    continue;
                }
                if (mat == BM) {
                    if (r == 0) { mat = BIC; --j; r = -1; lead = true; break; }
                    const int32_t ms = sub(node, r, j);
                    const int src = pick(node, r - 1, j - 1, val(node, BM, r, j), ms, ms, ms, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --r; --j;
                } else if (mat == BIR) {
                    if (r == 0) { lead = true; r = -1; break; }
                    const int src = pick(node, r - 1, j, val(node, BIR, r, j), -go, -go, -ge, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --r;
                } else {
                    const int src = pick(node, r, j - 1, val(node, BIC, r, j), -go, -ge, -go, node, r, j);
                    if (src < 0) return VGK_EINVAL;
                    mat = (uint32_t)src; --j;
                }
            }
            if (lead) { mat = BIC; while (j > 0) { emit(node, VGK_OP_D, 1); --j; } }
            if (at_deflection(node, r, 0)) {                                                   // (:1158-1222)
                emit(node, op_of(mat), 1);
                const Defl d = stack[cur].d[cur_defl++];
                if (!find_deflect_seed(node, d.to_node, dpath)) return VGK_EINVAL;
                for (int32_t e : dpath) emit(e, op_of(mat), 0);
                if (r == 0 && mat == BM) lead = true;
                if (mat == BM) --r;
                mat = (uint32_t)d.to_mat; node = d.to_node; j = nd(node).len - 1;
                continue;
            }
            const BSeed* seeds = T.seeds.data() + hp.seeds.off + n.seed_off;
            const uint32_t* pool = T.pool.data() + hp.pool.off;
            int found = -1; uint32_t fmat = BM; bool flead = lead;
            if (lead) {
                emit(node, VGK_OP_D, 1);
                for (uint32_t si = 0; si < n.n_seeds; ++si) {
                    const BNode& sd = nd((int32_t)seeds[si].node);
                    const int32_t diff = (int32_t)((int64_t)ge * (sd.cum + sd.len - n.cum));
                    if (diff == 0 && found < 0) found = (int)si;
                    else propose(S - diff, node, r, 0, (int32_t)seeds[si].node, BIC);
                }
                if (found < 0) {
                    if (!n.as_source) return VGK_EINVAL;
                    for (uint32_t q = 0; q < n.src_path_len; ++q) emit((int32_t)pool[n.src_path_off + q], VGK_OP_D, 0);
                    return VGK_OK;
                }
            } else {
                emit(node, op_of(mat), 1);
                const int32_t cur_val = val(node, mat == BM ? BM : BIC, r, 0);
                const int32_t ms = mat == BM ? sub(node, r, 0) : 0;
                for (uint32_t si = 0; si < n.n_seeds; ++si) {
                    const int32_t seed = (int32_t)seeds[si].node;
                    const BNode& sd = nd(seed);
                    const int64_t snt = sd.top + sd.len, snb = sd.bot + sd.len, sj = sd.len - 1;
                    if (r > snb - (mat == BIC ? 1 : 0) || r < snt) continue;
                    if (mat == BM && r == 0) {
                        const int32_t diff = cur_val - (-go - (sd.cum + sd.len - 1) * ge + ms);
                        if (diff == 0 && found < 0) { found = (int)si; fmat = BIC; flead = true; }
                        else propose(S - diff, node, r, 0, seed, BIC);
                        continue;
                    }
                    const int64_t sr = mat == BM ? r - 1 : r;
                    const int32_t dm = mat == BM ? ms : -go, dc = mat == BM ? ms : -ge, dr = mat == BM ? ms : -go;
                    { const int32_t src = val(seed, BM, sr, sj), diff = cur_val - (src + dm);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BM; } else if (live(src)) propose(S - diff, node, r, 0, seed, BM); }
                    { const int32_t src = val(seed, BIC, sr, sj); if (live(src)) { const int32_t diff = cur_val - (src + dc);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BIC; } else propose(S - diff, node, r, 0, seed, BIC); } }
                    { const int32_t src = val(seed, BIR, sr, sj); if (live(src)) { const int32_t diff = cur_val - (src + dr);
                      if (diff == 0 && found < 0) { found = (int)si; fmat = BIR; } else propose(S - diff, node, r, 0, seed, BIR); } }
                }
                if (found < 0) {
                    if (!n.as_source) return VGK_EINVAL;
                    int64_t ins;
                    if (mat == BM) { if (cur_val != (r > 0 ? -go - (int32_t)(r - 1) * ge : 0) + ms) return VGK_EINVAL; ins = r; }
                    else           { if (cur_val != -go - (int32_t)r * ge - go) return VGK_EINVAL; ins = r + 1; }
                    for (uint32_t q = 0; q < n.src_path_len; ++q) emit((int32_t)pool[n.src_path_off + q], VGK_OP_D, 0);
                    const int32_t end_node = n.src_path_len ? (int32_t)pool[n.src_path_off + n.src_path_len - 1] : node;
                    for (int64_t q = 0; q < ins; ++q) emit(end_node, VGK_OP_I, 1);
                    return VGK_OK;
                }
            }
            const BSeed& sr = seeds[found];
            for (uint32_t q = 0; q < sr.path_len; ++q) emit((int32_t)pool[sr.path_off + q], op_of(mat), 0);
            if (!lead) { if (mat == BM) --r; mat = fmat; lead = flead; }
            node = (int32_t)sr.node; j = nd(node).len - 1;
        }
    }
    // all alignments of the problem, best first (BandedGlobalAligner::traceback :2329-2423)
    int run(uint32_t max_alns, std::vector<vgk_result>& results, std::vector<vgk_op>& ops) {
        go = ctx->sc.gap_open; ge = ctx->sc.gap_extend; L = p.read_len; max = max_alns;
        for (uint32_t c = 0; c < hp.starts.len; ++c) {
            const int32_t u = (int32_t)T.starts[hp.starts.off + c];
            const BNode& n = nd(u);
            const int64_t k = (L - 1) - (n.len - 1) - n.top;
            if (k < 0 || k > n.bot - n.top) continue;
            const int32_t cand[3] = { val(u, BM, L - 1, n.len - 1), val(u, BIR, L - 1, n.len - 1), val(u, BIC, L - 1, n.len - 1) };
            const uint32_t cmat[3] = { BM, BIR, BIC };
            for (int q = 0; q < 3; ++q) if (live(cand[q])) insert({}, cand[q], Defl{u, L - 1, n.len - 1, u, (int32_t)cmat[q]}, T.start_prefix[hp.starts.off + c]);
        }
        const int32_t empty_score = -go - (int32_t)(L - 1) * ge;
        size_t next_empty = 0, n_empty = hp.empty_walks.size();
        if (stack.empty() && !n_empty) return VGK_ENOBAND;
        while (cur < stack.size() || next_empty < n_empty) {
            vgk_result out{}; out.ops_begin = (uint32_t)ops.size();
            const bool take_empty = cur >= stack.size() ? true : (empty_score >= stack[cur].score && next_empty < n_empty);
            if (take_empty) {                                                                  // next_empty_alignment (:2616-2668)
                const Span w = hp.empty_walks[next_empty++];
                out.score = empty_score;
                for (uint32_t e = w.len; e-- > 0;) {
                    vgk_op o{}; o.node = T.prefix[w.off + e];
                    if (e + 1 == w.len) { o.op = VGK_OP_I; o.len = (uint16_t)L; } else { o.op = VGK_OP_M; o.len = 0; }
                    ops.push_back(o);
                }
                --max;
                if ((int64_t)stack.size() > max) { if (cur + 1 == stack.size()) { stack.pop_back(); next_empty = n_empty; } else stack.pop_back(); }
            } else {
                cur_defl = 1; runs.clear();
                const int rc = trace();
                if (rc != VGK_OK) return rc;
                out.score = stack[cur].score;
                for (size_t q = runs.size(); q-- > 0;) { vgk_op o = runs[q]; if (o.len == 0) o.op = VGK_OP_M; ops.push_back(o); }
                const Span pre = stack[cur].prefix;
                for (uint32_t e = pre.len; e-- > 0;) { vgk_op o{}; o.node = T.prefix[pre.off + e]; o.op = VGK_OP_M; o.len = 0; ops.push_back(o); }
                ++cur;
                if (cur >= stack.size()) next_empty = n_empty;
            }
            out.n_ops = (uint32_t)(ops.size() - out.ops_begin); out.status = VGK_OK;
            results.push_back(out);
            if (results.size() >= max_alns) break;
        }
        return results.empty() ? VGK_ENOBAND : VGK_OK;
    }
};

}  // namespace

extern "C" {

// max_alt_alns == 0: the primary alignment, traced on the device; otherwise the k best, enumerated on the host over the
// device-filled score matrices (results[i * max_alt_alns + k], n_alignments[i])
static int banded_align_impl(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n, uint32_t max_alt_alns,
                             vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    const bool multi = max_alt_alns > 0;
    if (multi && !n_alignments) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (multi) ctx->multi_host_walks = 0;
    ctx->banded_ms[0] = ctx->banded_ms[1] = 0; ctx->banded_cells = 0; ctx->banded_bytes = 0; ctx->banded_last_valid = false;
    uint64_t budget = ctx->be->memory_bytes() / 2;
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    if (!budget) budget = 1ull << 30;
    Backend* be = ctx->be.get();
    const bool qa = ctx->has_qa;
    if (!ctx->banded_host) ctx->banded_host = std::make_shared<HostArenas>();
    HostArenas& H = *static_cast<HostArenas*>(ctx->banded_host.get());

    const bool timing = std::getenv("VGAMD_BANDED_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto lap = [&](const char* what) { if (!timing) return; auto t = now(); std::fprintf(stderr, "[vgk_banded_align] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
    // pass 1: geometry and tables of every problem
    std::vector<Prep> hps(n);
    if (H.store.empty()) { H.scratch.resize(MAX_THREADS); H.store.resize(MAX_THREADS); }
    std::vector<Scratch>& scratch = H.scratch; std::vector<Store>& store = H.store;
    for (Store& T : store) T.clear();
    parallel_for(n, [&](uint32_t i, unsigned t) { hps[i].thread = t; prepare(ctx, problems[i], hps[i], scratch[t], store[t]); });
    lap("prepare");

    size_t used = 0; int rc_all = VGK_OK;
    std::vector<uint32_t> owner;
    uint32_t i = 0;
    while (i < n) {
        // pass 2: place a sub-batch that fits the budget
        owner.clear();
        uint64_t n_nodes = 0, n_seeds = 0, n_pool = 0, n_starts = 0, n_read = 0, n_graph = 0, tb_bytes = 0, last_elems = 0, ops_total = 0;
        uint32_t j = i;
        for (; j < n; ++j) {
            const Prep& hp = hps[j];
            if (!hp.on_device) continue;
            const vgk_banded_problem& p = problems[j];
            if (!owner.empty() && ((tb_bytes + hp.tb_bytes) * (multi ? 13 : 1) + (last_elems + hp.last_elems) * 4 + (ops_total + hp.ops_cap) * 2 * sizeof(vgk_op) > budget ||
                                   n_nodes + p.graph.n_nodes > 0xfffffff0ull || n_read + p.read_len > 0xfffffff0ull || n_graph + hp.bases > 0xfffffff0ull)) break;
            n_nodes += p.graph.n_nodes; n_seeds += hp.seeds.len; n_pool += hp.pool.len; n_starts += hp.starts.len;
            n_read += p.read_len; n_graph += hp.bases; tb_bytes += hp.tb_bytes; last_elems += hp.last_elems; ops_total += hp.ops_cap;
            owner.push_back(j);
        }
        const uint32_t m = (uint32_t)owner.size();
        BProb* probs = H.probs.get(be, m);
        if (!probs) return VGK_ENOMEM;
        {
            uint64_t a_nodes = 0, a_seeds = 0, a_pool = 0, a_starts = 0, a_read = 0, a_graph = 0, a_tb = 0, a_last = 0, a_ops = 0;
            for (uint32_t a = 0; a < m; ++a) {
                Prep& hp = hps[owner[a]]; const vgk_banded_problem& p = problems[owner[a]];
                BProb pb{};
                pb.L = p.read_len; pb.n_nodes = p.graph.n_nodes; pb.Hpad = hp.Hpad; pb.graph_len = (uint32_t)hp.bases;
                pb.node_base = (uint32_t)a_nodes; pb.seed_base = (uint32_t)a_seeds; pb.pool_base = (uint32_t)a_pool; pb.start_base = (uint32_t)a_starts;
                pb.n_starts = hp.starts.len; pb.read_off = (uint32_t)a_read; pb.graph_off = (uint32_t)a_graph;
                pb.tb_base = a_tb; pb.last_base = a_last; pb.ops_off = a_ops; pb.ops_cap = hp.ops_cap;
                a_nodes += pb.n_nodes; a_seeds += hp.seeds.len; a_pool += hp.pool.len; a_starts += hp.starts.len;
                a_read += pb.L; a_graph += hp.bases; a_tb += hp.tb_bytes; a_last += hp.last_elems; a_ops += hp.ops_cap;
                hp.arena = a; probs[a] = pb;
            }
        }
        lap("place");
        BResult* dres = H.dres.get(be, m + 1);
        if (!dres) return VGK_ENOMEM;
        const vgk_op* dops = nullptr;
        if (m) {
            // pass 3: copy into the arenas
            BNode* nodes = H.nodes.get(be, n_nodes); BSeed* seeds = H.seeds.get(be, n_seeds); uint32_t* pool = H.pool.get(be, n_pool); BStart* starts = H.starts.get(be, n_starts);
            uint8_t* reads = H.reads.get(be, n_read); uint8_t* quals = qa ? H.quals.get(be, n_read) : nullptr; uint8_t* graph = H.graph.get(be, n_graph);
            if (!nodes || !seeds || !pool || !starts || !reads || (qa && !quals) || !graph) return VGK_ENOMEM;
            parallel_for(m, [&](uint32_t a, unsigned) {
                const Prep& hp = hps[owner[a]]; const BProb& pb = probs[a]; const vgk_banded_problem& p = problems[owner[a]];
                const Store& T = store[hp.thread];
                std::copy(T.nodes.begin() + hp.nodes.off, T.nodes.begin() + hp.nodes.off + hp.nodes.len, nodes + pb.node_base);
                std::copy(T.seeds.begin() + hp.seeds.off, T.seeds.begin() + hp.seeds.off + hp.seeds.len, seeds + pb.seed_base);
                std::copy(T.pool.begin() + hp.pool.off, T.pool.begin() + hp.pool.off + hp.pool.len, pool + pb.pool_base);
                for (uint32_t q = 0; q < hp.starts.len; ++q) starts[pb.start_base + q].node = T.starts[hp.starts.off + q];
                code_bases<true>(reads + pb.read_off, p.read, pb.L);
                if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
                code_bases<true>(graph + pb.graph_off, p.graph.seq, pb.graph_len);
            });
            lap("arenas");
            // launches: one per rows-per-lane class; inside a class the problems with the most cells first (counting sort on log2(cells))
            uint32_t* order = H.order.get(be, m);
            if (!order) return VGK_ENOMEM;
            std::vector<BandedLaunch> launches;
            {
                auto key = [&](uint32_t a) { return hps[owner[a]].order_key; };
                std::vector<uint32_t> count(B_ROW_CLASSES * 64 + 1, 0);
                for (uint32_t a = 0; a < m; ++a) ++count[key(a) + 1];
                for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
                std::vector<uint32_t> at(count.begin(), count.end() - 1);
                for (uint32_t a = 0; a < m; ++a) order[at[key(a)]++] = a;
                for (uint32_t r = 0; r < B_ROW_CLASSES; ++r) {
                    const uint32_t lo = count[r * 64], hi = count[(r + 1) * 64];
                    if (lo == hi) continue;
                    // LDS staging area: score table | read codes | qualities | graph codes of the largest problem of the launch
                    uint64_t lds = 0;
                    for (uint32_t b = lo; b < hi; ++b) { const BProb& pb = probs[order[b]]; lds = std::max<uint64_t>(lds, (qa ? 6400u : 32u) + (uint64_t)pb.L * (qa ? 2 : 1) + pb.graph_len + 96); }
                    launches.push_back({1u << r, lo, hi - lo, lds <= 40 * 1024 ? (uint32_t)lds : 0u});
                }
            }
            lap("sort");
            BandedParams P{};
            int rc;
            // plain contexts: the table, and behind it its rows as 64-bit words for the kernel's byte permute (banded_device.hpp BMAT_ROWS_AT)
            int8_t* mat_rows = ctx->banded_mat_rows;                            // (lives as long as the asynchronous upload needs it)
            std::memset(mat_rows, 0, BMAT_BYTES);
            if (!qa) {
                std::memcpy(mat_rows, ctx->sc.matrix, 25);
                for (int g = 0; g < 5; ++g) std::memcpy(mat_rows + BMAT_ROWS_AT + 8 * g, ctx->sc.matrix + 5 * g, 5);
            }
            const int8_t* mat = qa ? ctx->qmat.data() : mat_rows;
            if ((rc = stage(ctx, S_PROBS, (const BProb*)probs, m, P.probs)) || (rc = stage(ctx, S_ORDER, (const uint32_t*)order, m, P.order)) ||
                (rc = stage(ctx, S_NODES, (const BNode*)nodes, n_nodes, P.nodes)) || (rc = stage(ctx, S_SEEDS, (const BSeed*)seeds, n_seeds, P.seeds)) ||
                (rc = stage(ctx, S_POOL, (const uint32_t*)pool, n_pool, P.pool)) || (rc = stage(ctx, S_STARTS, (const BStart*)starts, n_starts, P.starts)) ||
                (rc = stage(ctx, S_READS, (const uint8_t*)reads, n_read, P.reads)) || (qa && (rc = stage(ctx, S_QUALS, (const uint8_t*)quals, n_read, P.quals))) ||
                (rc = stage(ctx, S_GRAPH, (const uint8_t*)graph, n_graph, P.graph)) || (rc = stage(ctx, S_MAT, mat, qa ? 6400 : BMAT_BYTES, P.mat))) return rc;
            lap("h2d");
            P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend; P.n = m;
            P.tb = (uint8_t*)ensure(ctx, S_TB, std::max<uint64_t>(tb_bytes, 256));
            P.last = (int32_t*)ensure(ctx, S_LAST, std::max<uint64_t>(last_elems, 64) * sizeof(int32_t));
            P.ops = (vgk_op*)ensure(ctx, S_OPS, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
            P.dense = (vgk_op*)ensure(ctx, S_DENSE, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
            uint8_t* rblock = (uint8_t*)ensure(ctx, S_RESULTS, (size_t)m * sizeof(BResult) + 64);
            if (multi) P.scores = (int32_t*)ensure(ctx, S_SCORES, std::max<uint64_t>(tb_bytes, 256) * 3 * sizeof(int32_t));
            if (!P.tb || !P.last || !P.ops || !P.dense || !rblock || (multi && !P.scores)) return VGK_ENOMEM;
            P.dense_count = (unsigned long long*)rblock; P.results = (BResult*)(rblock + 64);
            if ((rc = be->zero(rblock, 64))) return rc;
            lap("scratch");
            if ((rc = be->run_banded(P, launches.data(), (uint32_t)launches.size()))) return rc;
            lap("kernels");
            ctx->banded_last = P; ctx->banded_last_launches = launches; ctx->banded_last_valid = (i == 0 && j == n);     // the whole call in one sub-batch
            if (multi) {
                // k-best.  Round 3: the alternates are enumerated by a kernel over the score matrices where they lie (banded_multi_device.hpp,
                // one lane per problem); only what it declines — problems with chains of empty nodes from source to sink, tracebacks with more
                // deflections than a slot holds — is walked by a host thread as before, and only then do the matrices come back.
                ctx->banded_ms[0] += be->last_ms(3);
                std::vector<std::vector<vgk_result>> pres(m); std::vector<std::vector<vgk_op>> pops(m); std::vector<int> pstat(m, VGK_ETOOBIG);
                // (the walk's op windows: two op lists per alternate; a sub-batch whose windows would not fit beside the matrices stays with the host threads)
                uint64_t window_ops = 0;
                for (uint32_t a = 0; a < m; ++a) window_ops += 2ull * max_alt_alns * probs[a].ops_cap;
                const bool on_device = max_alt_alns + 1 <= 64 && !std::getenv("VGAMD_MULTI_HOST_WALK") && window_ops < (1ull << 32) && window_ops * sizeof(vgk_op) <= budget / 2;
                if (on_device) {
                    BandedMultiParams Q{};
                    Q.P = P; Q.max_alt = max_alt_alns;
                    std::vector<uint32_t> sp_off(n_starts + 1, 0), sp_len(n_starts + 1, 0), prefix; std::vector<uint8_t> host_only(m, 0); std::vector<uint64_t> ops_off(m + 1, 0);
                    for (uint32_t a = 0; a < m; ++a) {
                        const Prep& hp = hps[owner[a]]; const Store& T = store[hp.thread];
                        host_only[a] = hp.empty_walks.empty() ? 0 : 1;
                        for (uint32_t c = 0; c < hp.starts.len; ++c) {
                            const Span pre = T.start_prefix[hp.starts.off + c];
                            sp_off[probs[a].start_base + c] = (uint32_t)prefix.size(); sp_len[probs[a].start_base + c] = pre.len;
                            prefix.insert(prefix.end(), T.prefix.begin() + (long)pre.off, T.prefix.begin() + (long)pre.off + pre.len);
                        }
                        ops_off[a + 1] = ops_off[a] + 2ull * max_alt_alns * probs[a].ops_cap;
                    }
                    const uint64_t n_res = (uint64_t)m * max_alt_alns, slots = max_alt_alns + 1;
                    auto up = [&](int slot, const void* src, size_t bytes) -> void* {
                        void* d = ctx->ensure_scratch(slot, std::max<size_t>(bytes, 16)); if (!d) return nullptr;
                        if (src && bytes && be->upload(d, src, bytes)) return nullptr;
                        return d;
                    };
                    Q.pool = (BmTrace*)up(72, nullptr, sizeof(BmTrace) * slots * m); Q.order = (uint32_t*)up(73, nullptr, sizeof(uint32_t) * slots * m);
                    Q.sp_off = (const uint32_t*)up(74, sp_off.data(), sizeof(uint32_t) * (n_starts + 1)); Q.sp_len = (const uint32_t*)up(75, sp_len.data(), sizeof(uint32_t) * (n_starts + 1));
                    Q.prefix = (const uint32_t*)up(76, prefix.data(), sizeof(uint32_t) * prefix.size()); Q.host_only = (const uint8_t*)up(77, host_only.data(), m);
                    Q.results = (vgk_result*)up(78, nullptr, sizeof(vgk_result) * n_res); Q.n_alignments = (uint32_t*)up(79, nullptr, sizeof(uint32_t) * m);
                    Q.ops = (vgk_op*)up(80, nullptr, sizeof(vgk_op) * ops_off[m]); Q.ops_off = (const uint64_t*)up(81, ops_off.data(), sizeof(uint64_t) * m);
                    Q.status = (int32_t*)up(82, nullptr, sizeof(int32_t) * m);
                    if (!Q.pool || !Q.order || !Q.sp_off || !Q.sp_len || !Q.prefix || !Q.host_only || !Q.results || !Q.n_alignments || !Q.ops || !Q.ops_off || !Q.status) return VGK_ENOMEM;
                    if ((rc = be->zero(Q.results, sizeof(vgk_result) * n_res))) return rc;
                    if ((rc = be->run_banded_multi(Q))) return rc;
                    std::vector<int32_t> dstat(m); std::vector<uint32_t> dcnt(m);
                    if ((rc = be->download(dstat.data(), Q.status, sizeof(int32_t) * m))) return rc;
                    if ((rc = be->download(dcnt.data(), Q.n_alignments, sizeof(uint32_t) * m))) return rc;
                    std::vector<vgk_result> dres(n_res); std::vector<vgk_op> dops2;
                    const uint32_t blocks = (uint32_t)((n_res + Backend::OPS_SCAN_BLOCK - 1) / Backend::OPS_SCAN_BLOCK);
                    uint32_t* offs = (uint32_t*)up(83, nullptr, sizeof(uint32_t) * n_res);
                    uint32_t* sums = (uint32_t*)up(84, nullptr, sizeof(uint32_t) * (blocks + 8));
                    if (!offs || !sums) return VGK_ENOMEM;
                    uint64_t total = 0;
                    rc = be->ops_offsets(Q.results, (uint32_t)n_res, offs, sums, &total);
                    if (rc == VGK_OK) {
                        vgk_result* pres_d = (vgk_result*)up(85, nullptr, sizeof(vgk_result) * n_res);
                        vgk_op* pops_d = (vgk_op*)up(86, nullptr, sizeof(vgk_op) * std::max<uint64_t>(total, 1));
                        if (!pres_d || !pops_d) return VGK_ENOMEM;
                        if ((rc = be->ops_gather(Q.results, Q.ops, (uint32_t)n_res, offs, sums, pres_d, pops_d))) return rc;
                        if ((rc = be->sync_fetch())) return rc;
                        dops2.resize(total);
                        if ((rc = be->download(dres.data(), pres_d, sizeof(vgk_result) * n_res))) return rc;
                        if (total && (rc = be->download(dops2.data(), pops_d, sizeof(vgk_op) * total))) return rc;
                    } else if (rc == VGK_EUNSUPPORTED) {
                        dops2.resize(ops_off[m]);
                        if ((rc = be->download(dres.data(), Q.results, sizeof(vgk_result) * n_res))) return rc;
                        if (ops_off[m] && (rc = be->download(dops2.data(), Q.ops, sizeof(vgk_op) * ops_off[m]))) return rc;
                    } else return rc;
                    for (uint32_t a = 0; a < m; ++a) {
                        if (dstat[a] == VGK_ETOOBIG) continue;                       // a host thread walks this one
                        pstat[a] = dstat[a];
                        if (dstat[a] != VGK_OK) continue;
                        pres[a].assign(dres.begin() + (size_t)a * max_alt_alns, dres.begin() + (size_t)a * max_alt_alns + dcnt[a]);
                        for (vgk_result& r : pres[a]) {
                            const uint32_t at = (uint32_t)pops[a].size();
                            pops[a].insert(pops[a].end(), dops2.begin() + r.ops_begin, dops2.begin() + r.ops_begin + r.n_ops);
                            r.ops_begin = at;
                        }
                    }
                }
                std::vector<uint32_t> todo;
                for (uint32_t a = 0; a < m; ++a) if (pstat[a] == VGK_ETOOBIG) todo.push_back(a);
                ctx->multi_host_walks += todo.size();
                if (!todo.empty()) {
                    int32_t* hs = H.scores.get(be, tb_bytes * 3 + 1);
                    if (!hs) return VGK_ENOMEM;
                    if ((rc = be->download(hs, P.scores, (size_t)tb_bytes * 3 * sizeof(int32_t)))) return rc;
                    parallel_for((uint32_t)todo.size(), [&](uint32_t k, unsigned) {
                        const uint32_t a = todo[k];
                        const Prep& hp = hps[owner[a]];
                        MultiTracer mt{ctx, problems[owner[a]], hp, store[hp.thread], hs + 3 * probs[a].tb_base};
                        pres[a].clear(); pops[a].clear();
                        pstat[a] = mt.run(max_alt_alns, pres[a], pops[a]);
                    });
                }
                lap("d2h");
                for (uint32_t a = 0; a < m; ++a) {
                    const uint32_t q = owner[a]; vgk_result* r = results + (size_t)q * max_alt_alns;
                    ctx->banded_cells += hps[q].cells;
                    n_alignments[q] = 0;
                    if (pstat[a] != VGK_OK) { std::memset(r, 0, sizeof *r); r->status = pstat[a]; continue; }
                    if (!ops || used + pops[a].size() > ops_cap) { std::memset(r, 0, sizeof *r); r->status = VGK_EOPS; rc_all = VGK_EOPS; continue; }
                    for (size_t k = 0; k < pres[a].size(); ++k) { r[k] = pres[a][k]; r[k].ops_begin += (uint32_t)used; }
                    std::copy(pops[a].begin(), pops[a].end(), ops + used); used += pops[a].size();
                    n_alignments[q] = (uint32_t)pres[a].size();
                }
                lap("results");
            } else {
                unsigned long long dense_n = 0;
                if ((rc = be->download(&dense_n, P.dense_count, sizeof dense_n))) return rc;
                if ((rc = be->download(dres, P.results, (size_t)m * sizeof(BResult)))) return rc;
                vgk_op* hd = H.dops.get(be, dense_n + 1);
                if (!hd) return VGK_ENOMEM;
                if (dense_n && (rc = be->download(hd, P.dense, (size_t)dense_n * sizeof(vgk_op)))) return rc;
                dops = hd;
                ctx->banded_ms[0] += be->last_ms(3); ctx->banded_ms[1] += be->last_ms(4);
                lap("d2h");
            }
        }
        if (multi) {       // problems that never reached the device carry their status; the rest was written above
            for (uint32_t q = i; q < j; ++q) if (!hps[q].on_device) { vgk_result* r = results + (size_t)q * max_alt_alns; std::memset(r, 0, sizeof *r); r->status = hps[q].status; n_alignments[q] = 0; }
            i = j;
            continue;
        }
        // results in the caller's order; the empty-walk rule and the empty sink prefixes are host bookkeeping (:2611-2668, :196-203):
        // sizes first, then a prefix sum, then every problem writes its own slice
        parallel_for(j - i, [&](uint32_t k, unsigned) {
            const uint32_t q = i + k; Prep& hp = hps[q]; vgk_result& r = results[q];
            std::memset(&r, 0, sizeof r);
            hp.need = 0;
            if (!hp.on_device) { r.status = hp.status; return; }
            const BResult& dr = dres[hp.arena]; const vgk_banded_problem& p = problems[q];
            const int32_t empty_score = -ctx->sc.gap_open - (int32_t)(p.read_len - 1) * ctx->sc.gap_extend;
            const bool have = dr.status != VGK_ENOBAND;
            hp.use_empty_walk = hp.have_empty_walk && (!have || empty_score >= dr.score);
            if (hp.use_empty_walk) { hp.need = hp.empty_walk.len; r.score = empty_score; r.status = VGK_OK; }
            else if (dr.status != VGK_OK) r.status = dr.status;
            else { hp.need = dr.n_ops + store[hp.thread].start_prefix[hp.starts.off + dr.start].len; r.score = dr.score; r.status = VGK_OK; }
        });
        for (uint32_t q = i; q < j; ++q) {
            Prep& hp = hps[q]; vgk_result& r = results[q];
            r.ops_begin = (uint32_t)used;
            if (hp.on_device) { ctx->banded_cells += hp.cells; ctx->banded_bytes += hp.in_bytes + 2ull * dres[hp.arena].n_ops; }
            if (r.status != VGK_OK) continue;
            if (!ops || used + hp.need > ops_cap) { r.status = VGK_EOPS; rc_all = VGK_EOPS; hp.need = 0; continue; }
            r.n_ops = hp.need; used += hp.need;
        }
        parallel_for(j - i, [&](uint32_t k, unsigned) {
            const uint32_t q = i + k; const Prep& hp = hps[q]; const vgk_result& r = results[q];
            if (r.status != VGK_OK || !r.n_ops) return;
            const Store& T = store[hp.thread];
            vgk_op* out = ops + r.ops_begin;
            if (hp.use_empty_walk) {
                for (uint32_t e = hp.empty_walk.len; e-- > 0;) {
                    vgk_op o{}; o.node = T.prefix[hp.empty_walk.off + e];
                    if (e + 1 == hp.empty_walk.len) { o.op = VGK_OP_I; o.len = (uint16_t)problems[q].read_len; } else { o.op = VGK_OP_M; o.len = 0; }
                    *out++ = o;
                }
            } else {
                const BResult& dr = dres[hp.arena];
                const vgk_op* src = dops + dr.ops_begin;
                for (uint32_t e = 0; e < dr.n_ops; ++e) { vgk_op o = src[e]; if (o.len == 0) o.op = VGK_OP_M; *out++ = o; }
                const Span pre = T.start_prefix[hp.starts.off + dr.start];
                for (uint32_t e = pre.len; e-- > 0;) { vgk_op o{}; o.node = T.prefix[pre.off + e]; o.op = VGK_OP_M; o.len = 0; *out++ = o; }
            }
        });
        lap("results");
        i = j;
    }
    if (ops_written) *ops_written = used;
    return rc_all;
}

// ---- a large call of primary alignments as a pipeline of sub-batches, two in flight ---------------------------------------------------
// banded_align_impl prepares everything, then runs sub-batch after sub-batch with the host waiting for each one's kernels: 14 ms of host work
// beside 7 ms of kernels per 100 000 problems.  Here the call is cut into quarters; a quarter's geometry (prepare) and arenas are made while
// the kernels of the quarter before run, its results are handed out while the next one's run; the ops and results come back on the fetch
// stream behind an event.  Same passes, same tables, same kernels — only the order of the host's work changes.  (The batch does not stay
// resident for vgk_banded_rerun: a caller that wants that sets VGAMD_BANDED_ONE_BATCH=1.)
struct BSub { uint32_t i = 0, j = 0, m = 0; std::vector<uint32_t> owner; BandedParams P{}; std::vector<BandedLaunch> launches; uint64_t sizes[9] = {0}; };

static int banded_align_pipelined(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                                  vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->banded_ms[0] = ctx->banded_ms[1] = 0; ctx->banded_cells = 0; ctx->banded_bytes = 0; ctx->banded_last_valid = false;
    uint64_t budget = ctx->be->memory_bytes() / 4;
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    if (!budget) budget = 1ull << 30;
    Backend* be = ctx->be.get();
    const bool qa = ctx->has_qa;
    if (!ctx->banded_host) ctx->banded_host = std::make_shared<HostArenas>();
    HostArenas& H = *static_cast<HostArenas*>(ctx->banded_host.get());
    if (!H.be) { H.be = be; for (void*& e : H.ev) e = be->event_create(); }
    const bool timing = std::getenv("VGAMD_BANDED_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[vgk_banded_align] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
    // (the records of 100 000 problems are 18 MB: constructed by the threads that prepare them, not by this one before anything else starts)
    struct PrepArray { Prep* p; uint32_t made = 0; ~PrepArray() { for (uint32_t k = 0; k < made; ++k) p[k].~Prep(); std::free(p); } } hps_store{(Prep*)std::malloc(sizeof(Prep) * (size_t)std::max<uint32_t>(n, 1))};
    if (!hps_store.p) return VGK_ENOMEM;
    Prep* const hps = hps_store.p;
    if (H.store.empty()) { H.scratch.resize(MAX_THREADS); H.store.resize(MAX_THREADS); }
    std::vector<Scratch>& scratch = H.scratch; std::vector<Store>& store = H.store;
    for (Store& T : store) T.clear();
    const uint32_t quarter = (n + 3u) / 4u;
    uint32_t prepared = 0;
    std::vector<uint64_t> qoff[2]; uint32_t qlo[2] = {0, 0};
    size_t used = 0; int rc_all = VGK_OK;
    // plain contexts: the table, and behind it its rows as 64-bit words for the kernel's byte permute (banded_device.hpp BMAT_ROWS_AT)
    int8_t* mat_rows = ctx->banded_mat_rows;
    std::memset(mat_rows, 0, BMAT_BYTES);
    if (!qa) { std::memcpy(mat_rows, ctx->sc.matrix, 25); for (int g = 0; g < 5; ++g) std::memcpy(mat_rows + BMAT_ROWS_AT + 8 * g, ctx->sc.matrix + 5 * g, 5); }
    const int8_t* mat = qa ? ctx->qmat.data() : mat_rows;

    // ---- first half of a sub-batch: geometry of its quarter (if not there yet), placement, arenas, upload, launch
    auto build = [&](uint32_t from, BSub& S, int set) -> int {
        PinnedSet& A = H.set[set];
        const uint32_t limit = std::min<uint32_t>(n, (from / quarter + 1u) * quarter);
        if (prepared < limit) {
            // the node records — 64 bytes a node, the bulk of the tables — are written straight into the quarter's page-locked table, a
            // problem's at the sum of the node counts before it (a declined problem's slots stay unused); sub-batches upload slices of it
            const uint32_t lo = prepared, par = (lo / quarter) & 1u;
            std::vector<uint64_t>& off = qoff[par]; off.assign((size_t)(limit - lo) + 1, 0);
            for (uint32_t k = 0; k < limit - lo; ++k) off[k + 1] = off[k] + problems[lo + k].graph.n_nodes;
            BNode* table = H.qnodes[par].get(be, off[limit - lo] + 1);
            if (!table) return VGK_ENOMEM;
            qlo[par] = lo;
            parallel_for(limit - lo, [&](uint32_t k, unsigned t) { Prep* hp = new (&hps[lo + k]) Prep(); hp->thread = t; prepare(ctx, problems[lo + k], *hp, scratch[t], store[t], table + off[k]); });
            prepared = limit; hps_store.made = limit;
            lap("prepare");
        }
        const uint32_t par = (from / quarter) & 1u; const std::vector<uint64_t>& noff = qoff[par]; const uint32_t nlo = qlo[par];
        auto node_at = [&](uint32_t q) { return noff[q - nlo] - noff[from - nlo]; };      // a problem's records inside the sub-batch's slice of the table
        S.i = from; S.owner.clear(); S.launches.clear();
        uint64_t n_nodes = 0, n_seeds = 0, n_pool = 0, n_starts = 0, n_read = 0, n_graph = 0, tb_bytes = 0, last_elems = 0, ops_total = 0;
        uint32_t j = from;
        // The usual case — the whole quarter fits the device budget — is placed by sums over chunks of problems on the host threads (one
        // thread walking 25 000 records and their problems was 0.65 ms a quarter); otherwise the running sum below cuts the quarter.
        struct Sums { uint64_t v[10]; };                          // problems on the device, then the nine arena sizes
        const uint32_t n_chunks = chunk_count(limit - from);
        std::vector<Sums> chunk(n_chunks + 1, Sums{});
        auto sizes_of = [&](uint32_t q, uint64_t (&v)[10]) {
            const Prep& hp = hps[q]; const vgk_banded_problem& p = problems[q];
            v[0] = 1; v[1] = p.graph.n_nodes; v[2] = hp.seeds.len; v[3] = hp.pool.len; v[4] = hp.starts.len; v[5] = p.read_len; v[6] = hp.bases; v[7] = hp.tb_bytes; v[8] = hp.last_elems; v[9] = hp.ops_cap;
        };
        parallel_chunks(limit - from, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Sums t{};
            for (uint32_t q = from + lo; q < from + hi; ++q) if (hps[q].on_device) { uint64_t v[10]; sizes_of(q, v); for (int x = 0; x < 10; ++x) t.v[x] += v[x]; }
            chunk[c + 1] = t;
        });
        for (uint32_t c = 0; c < n_chunks; ++c) for (int x = 0; x < 10; ++x) chunk[c + 1].v[x] += chunk[c].v[x];
        const Sums& all = chunk[n_chunks];
        const bool whole = all.v[7] + all.v[8] * 4 + all.v[9] * 2 * sizeof(vgk_op) <= budget && all.v[1] <= 0xfffffff0ull && all.v[5] <= 0xfffffff0ull && all.v[6] <= 0xfffffff0ull;
        if (whole) {
            const uint32_t m_all = (uint32_t)all.v[0];
            S.owner.resize(m_all);
            BProb* probs_w = H.set[set].probs.get(be, std::max<uint32_t>(m_all, 1));
            if (!probs_w) return VGK_ENOMEM;
            parallel_chunks(limit - from, [&](uint32_t lo, uint32_t hi, uint32_t c) {
                Sums at = chunk[c];
                for (uint32_t q = from + lo; q < from + hi; ++q) {
                    Prep& hp = hps[q];
                    if (!hp.on_device) continue;
                    const vgk_banded_problem& p = problems[q];
                    uint64_t v[10]; sizes_of(q, v);
                    BProb pb{};
                    pb.L = p.read_len; pb.n_nodes = p.graph.n_nodes; pb.Hpad = hp.Hpad; pb.graph_len = (uint32_t)hp.bases;
                    pb.node_base = (uint32_t)node_at(q); pb.seed_base = (uint32_t)at.v[2]; pb.pool_base = (uint32_t)at.v[3]; pb.start_base = (uint32_t)at.v[4];
                    pb.n_starts = hp.starts.len; pb.read_off = (uint32_t)at.v[5]; pb.graph_off = (uint32_t)at.v[6];
                    pb.tb_base = at.v[7]; pb.last_base = at.v[8]; pb.ops_off = at.v[9]; pb.ops_cap = hp.ops_cap;
                    const uint32_t a = (uint32_t)at.v[0];
                    hp.arena = a; probs_w[a] = pb; S.owner[a] = q;
                    for (int x = 0; x < 10; ++x) at.v[x] += v[x];
                }
            });
            n_nodes = noff[limit - nlo] - noff[from - nlo]; n_seeds = all.v[2]; n_pool = all.v[3]; n_starts = all.v[4]; n_read = all.v[5]; n_graph = all.v[6]; tb_bytes = all.v[7]; last_elems = all.v[8]; ops_total = all.v[9];
            j = limit;
        }
        else for (; j < limit; ++j) {
            const Prep& hp = hps[j];
            if (!hp.on_device) continue;
            const vgk_banded_problem& p = problems[j];
            if (!S.owner.empty() && ((tb_bytes + hp.tb_bytes) + (last_elems + hp.last_elems) * 4 + (ops_total + hp.ops_cap) * 2 * sizeof(vgk_op) > budget ||
                                     n_nodes + p.graph.n_nodes > 0xfffffff0ull || n_read + p.read_len > 0xfffffff0ull || n_graph + hp.bases > 0xfffffff0ull)) break;
            n_nodes += p.graph.n_nodes; n_seeds += hp.seeds.len; n_pool += hp.pool.len; n_starts += hp.starts.len;
            n_read += p.read_len; n_graph += hp.bases; tb_bytes += hp.tb_bytes; last_elems += hp.last_elems; ops_total += hp.ops_cap;
            S.owner.push_back(j);
        }
        if (!whole) n_nodes = noff[j - nlo] - noff[from - nlo];      // (the slice of the quarter's table, declined problems' slots included)
        S.j = j;
        const std::vector<uint32_t>& owner = S.owner;
        const uint32_t m = S.m = (uint32_t)owner.size();
        if (!m) return VGK_OK;
        BProb* probs = A.probs.get(be, m);
        if (!probs) return VGK_ENOMEM;
        if (!whole) { uint64_t a_nodes = 0, a_seeds = 0, a_pool = 0, a_starts = 0, a_read = 0, a_graph = 0, a_tb = 0, a_last = 0, a_ops = 0;
          for (uint32_t a = 0; a < m; ++a) {
            Prep& hp = hps[owner[a]]; const vgk_banded_problem& p = problems[owner[a]];
            BProb pb{};
            pb.L = p.read_len; pb.n_nodes = p.graph.n_nodes; pb.Hpad = hp.Hpad; pb.graph_len = (uint32_t)hp.bases;
            pb.node_base = (uint32_t)node_at(owner[a]); pb.seed_base = (uint32_t)a_seeds; pb.pool_base = (uint32_t)a_pool; pb.start_base = (uint32_t)a_starts;
            pb.n_starts = hp.starts.len; pb.read_off = (uint32_t)a_read; pb.graph_off = (uint32_t)a_graph;
            pb.tb_base = a_tb; pb.last_base = a_last; pb.ops_off = a_ops; pb.ops_cap = hp.ops_cap;
            a_nodes += pb.n_nodes; a_seeds += hp.seeds.len; a_pool += hp.pool.len; a_starts += hp.starts.len;
            a_read += pb.L; a_graph += hp.bases; a_tb += hp.tb_bytes; a_last += hp.last_elems; a_ops += hp.ops_cap;
            hp.arena = a; probs[a] = pb;
          } }
        BSeed* seeds = A.seeds.get(be, n_seeds); uint32_t* pool = A.pool.get(be, n_pool); BStart* starts = A.starts.get(be, n_starts);
        uint8_t* reads = A.reads.get(be, n_read); uint8_t* quals = qa ? A.quals.get(be, n_read) : nullptr; uint8_t* graph = A.graph.get(be, n_graph);
        uint32_t* order = A.order.get(be, m);
        if (!seeds || !pool || !starts || !reads || (qa && !quals) || !graph || !order) return VGK_ENOMEM;
        parallel_for(m, [&](uint32_t a, unsigned) {
            const Prep& hp = hps[owner[a]]; const BProb& pb = probs[a]; const vgk_banded_problem& p = problems[owner[a]];
            const Store& T = store[hp.thread];
            std::copy(T.seeds.begin() + hp.seeds.off, T.seeds.begin() + hp.seeds.off + hp.seeds.len, seeds + pb.seed_base);
            std::copy(T.pool.begin() + hp.pool.off, T.pool.begin() + hp.pool.off + hp.pool.len, pool + pb.pool_base);
            for (uint32_t q = 0; q < hp.starts.len; ++q) starts[pb.start_base + q].node = T.starts[hp.starts.off + q];
            code_bases<true>(reads + pb.read_off, p.read, pb.L);
            if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
            code_bases<true>(graph + pb.graph_off, p.graph.seq, pb.graph_len);
        });
        // launches: one per rows-per-lane class; inside a class the problems with the most cells first (counting sort on log2(cells))
        { auto key = [&](uint32_t a) { return hps[owner[a]].order_key; };
          std::vector<uint32_t> count(B_ROW_CLASSES * 64 + 1, 0);
          for (uint32_t a = 0; a < m; ++a) ++count[key(a) + 1];
          for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
          std::vector<uint32_t> at(count.begin(), count.end() - 1);
          for (uint32_t a = 0; a < m; ++a) order[at[key(a)]++] = a;
          for (uint32_t r = 0; r < B_ROW_CLASSES; ++r) {
            const uint32_t lo = count[r * 64], hi = count[(r + 1) * 64];
            if (lo == hi) continue;
            uint64_t lds = 0;      // LDS staging area: score table | read codes | qualities | graph codes of the largest problem of the launch
            for (uint32_t b = lo; b < hi; ++b) { const BProb& pb = probs[order[b]]; lds = std::max<uint64_t>(lds, (qa ? 6400u : 32u) + (uint64_t)pb.L * (qa ? 2 : 1) + pb.graph_len + 96); }
            S.launches.push_back({1u << r, lo, hi - lo, lds <= 40 * 1024 ? (uint32_t)lds : 0u});
          } }
        lap("arenas");
        S.sizes[0] = n_nodes; S.sizes[1] = n_seeds; S.sizes[2] = n_pool; S.sizes[3] = n_starts; S.sizes[4] = n_read; S.sizes[5] = n_graph; S.sizes[6] = tb_bytes; S.sizes[7] = last_elems; S.sizes[8] = ops_total;
        return VGK_OK;
    };
    // ---- ... upload and launch (after the sub-batch before has been fetched: its downloads would queue behind these uploads — a BNode is 64
    // bytes, a quarter's tables 30 MB)
    auto launch = [&](BSub& S, int set) -> int {
        const uint32_t m = S.m;
        if (!m) return VGK_OK;
        PinnedSet& A = H.set[set]; const int base = set ? S_SET1 : 0;
        const uint64_t n_nodes = S.sizes[0], n_seeds = S.sizes[1], n_pool = S.sizes[2], n_starts = S.sizes[3], n_read = S.sizes[4], n_graph = S.sizes[5], tb_bytes = S.sizes[6], last_elems = S.sizes[7], ops_total = S.sizes[8];
        const uint32_t par = (S.i / quarter) & 1u;
        const BProb* probs = A.probs.p; const uint32_t* order = A.order.p; const BNode* nodes = H.qnodes[par].p + qoff[par][S.i - qlo[par]]; const BSeed* seeds = A.seeds.p; const uint32_t* pool = A.pool.p; const BStart* starts = A.starts.p;
        const uint8_t* reads = A.reads.p; const uint8_t* quals = A.quals.p; const uint8_t* graph = A.graph.p;
        BandedParams& P = S.P; P = BandedParams{};
        int rc;
        if ((rc = stage(ctx, base + S_PROBS, (const BProb*)probs, m, P.probs)) || (rc = stage(ctx, base + S_ORDER, (const uint32_t*)order, m, P.order)) ||
            (rc = stage(ctx, base + S_NODES, (const BNode*)nodes, n_nodes, P.nodes)) || (rc = stage(ctx, base + S_SEEDS, (const BSeed*)seeds, n_seeds, P.seeds)) ||
            (rc = stage(ctx, base + S_POOL, (const uint32_t*)pool, n_pool, P.pool)) || (rc = stage(ctx, base + S_STARTS, (const BStart*)starts, n_starts, P.starts)) ||
            (rc = stage(ctx, base + S_READS, (const uint8_t*)reads, n_read, P.reads)) || (qa && (rc = stage(ctx, base + S_QUALS, (const uint8_t*)quals, n_read, P.quals))) ||
            (rc = stage(ctx, base + S_GRAPH, (const uint8_t*)graph, n_graph, P.graph)) || (rc = stage(ctx, base + S_MAT, mat, qa ? 6400 : BMAT_BYTES, P.mat))) return rc;
        P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend; P.n = m;
        P.tb = (uint8_t*)ensure(ctx, base + S_TB, std::max<uint64_t>(tb_bytes, 256));
        P.last = (int32_t*)ensure(ctx, base + S_LAST, std::max<uint64_t>(last_elems, 64) * sizeof(int32_t));
        P.ops = (vgk_op*)ensure(ctx, base + S_OPS, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
        P.dense = (vgk_op*)ensure(ctx, base + S_DENSE, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
        uint8_t* rblock = (uint8_t*)ensure(ctx, base + S_RESULTS, (size_t)m * sizeof(BResult) + 64);
        if (!P.tb || !P.last || !P.ops || !P.dense || !rblock) return VGK_ENOMEM;
        P.dense_count = (unsigned long long*)rblock; P.results = (BResult*)(rblock + 64);
        if ((rc = be->zero(rblock, 64))) return rc;
        if ((rc = be->run_banded_async(P, S.launches.data(), (uint32_t)S.launches.size(), set))) return rc;
        if ((rc = be->event_record(H.ev[set]))) return rc;
        lap("launch");
        return VGK_OK;
    };

    // ---- second half: the packed ops and results back behind the event, then the caller's order (the empty-walk rule and the empty sink
    // prefixes are host bookkeeping, :2611-2668, :196-203): sizes first, then a running sum, then every problem writes its own slice
    auto finish = [&](BSub& S, int set) -> int {
        PinnedSet& A = H.set[set];
        const uint32_t m = S.m, i = S.i, j = S.j; const BandedParams& P = S.P;
        BResult* dres = A.dres.get(be, m + 1); const vgk_op* dops = nullptr;
        if (!dres) return VGK_ENOMEM;
        if (m) {
            int rc;
            if (H.ev[set]) { if ((rc = be->fetch_after(H.ev[set]))) return rc; }
            else if ((rc = be->sync())) return rc;
            unsigned long long* dense_n = A.count.get(be, 8);
            if (!dense_n) return VGK_ENOMEM;
            if ((rc = be->download_fetch_async(dense_n, P.dense_count, sizeof(unsigned long long)))) return rc;
            if ((rc = be->download_fetch_async(dres, P.results, (size_t)m * sizeof(BResult)))) return rc;
            if ((rc = be->sync_fetch())) return rc;
            vgk_op* hd = A.dops.get(be, dense_n[0] + 1);
            if (!hd) return VGK_ENOMEM;
            if (dense_n[0] && (rc = be->download_fetch(hd, P.dense, (size_t)dense_n[0] * sizeof(vgk_op)))) return rc;
            dops = hd;
            ctx->banded_ms[0] += be->banded_ms(set, 0); ctx->banded_ms[1] += be->banded_ms(set, 1);
            lap("fetch");
        }
        parallel_for(j - i, [&](uint32_t k, unsigned) {
            const uint32_t q = i + k; Prep& hp = hps[q]; vgk_result& r = results[q];
            std::memset(&r, 0, sizeof r);
            hp.need = 0;
            if (!hp.on_device) { r.status = hp.status; return; }
            const BResult& dr = dres[hp.arena]; const vgk_banded_problem& p = problems[q];
            const int32_t empty_score = -ctx->sc.gap_open - (int32_t)(p.read_len - 1) * ctx->sc.gap_extend;
            const bool have = dr.status != VGK_ENOBAND;
            hp.use_empty_walk = hp.have_empty_walk && (!have || empty_score >= dr.score);
            if (hp.use_empty_walk) { hp.need = hp.empty_walk.len; r.score = empty_score; r.status = VGK_OK; }
            else if (dr.status != VGK_OK) r.status = dr.status;
            else { hp.need = dr.n_ops + store[hp.thread].start_prefix[hp.starts.off + dr.start].len; r.score = dr.score; r.status = VGK_OK; }
        });
        for (uint32_t q = i; q < j; ++q) {
            Prep& hp = hps[q]; vgk_result& r = results[q];
            r.ops_begin = (uint32_t)used;
            if (hp.on_device) { ctx->banded_cells += hp.cells; ctx->banded_bytes += hp.in_bytes + 2ull * dres[hp.arena].n_ops; }
            if (r.status != VGK_OK) continue;
            if (!ops || used + hp.need > ops_cap) { r.status = VGK_EOPS; rc_all = VGK_EOPS; hp.need = 0; continue; }
            r.n_ops = hp.need; used += hp.need;
        }
        parallel_for(j - i, [&](uint32_t k, unsigned) {
            const uint32_t q = i + k; const Prep& hp = hps[q]; const vgk_result& r = results[q];
            if (r.status != VGK_OK || !r.n_ops) return;
            const Store& T = store[hp.thread];
            vgk_op* out = ops + r.ops_begin;
            if (hp.use_empty_walk) {
                for (uint32_t e = hp.empty_walk.len; e-- > 0;) {
                    vgk_op o{}; o.node = T.prefix[hp.empty_walk.off + e];
                    if (e + 1 == hp.empty_walk.len) { o.op = VGK_OP_I; o.len = (uint16_t)problems[q].read_len; } else { o.op = VGK_OP_M; o.len = 0; }
                    *out++ = o;
                }
            } else {
                const BResult& dr = dres[hp.arena];
                const vgk_op* src = dops + dr.ops_begin;
                for (uint32_t e = 0; e < dr.n_ops; ++e) { vgk_op o = src[e]; if (o.len == 0) o.op = VGK_OP_M; *out++ = o; }
                const Span pre = T.start_prefix[hp.starts.off + dr.start];
                for (uint32_t e = pre.len; e-- > 0;) { vgk_op o{}; o.node = T.prefix[pre.off + e]; o.op = VGK_OP_M; o.len = 0; *out++ = o; }
            }
        });
        lap("results");
        return VGK_OK;
    };

    BSub subs[2]; int set = 0; bool pending = false; int rc = VGK_OK;
    for (uint32_t i = 0; i < n && rc == VGK_OK;) {
        BSub& S = subs[set];
        rc = build(i, S, set);                                     // (host only: the kernels of the sub-batch before run meanwhile)
        i = S.j;
        if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
        if (rc == VGK_OK) rc = launch(S, set);
        pending = rc == VGK_OK;
        set ^= 1;
    }
    if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
    if (rc != VGK_OK) { be->sync(); be->sync_fetch(); return rc; }
    if (ops_written) *ops_written = used;
    return rc_all;
}

// ---- A large call of graphs WITHOUT empty nodes: the geometry on the device (banded_geom_device.hpp).  The host gathers each sub-batch's raw
// arrays — node lengths, predecessor lists, coded reads and bases: 0.8 kB a problem against the 2.1 kB of tables prepare() makes — checks them
// on the way, and a lane per problem makes the band, the node records, the flattened predecessors and the candidate end nodes where the fill
// kernels read them; what comes back before the fills are launched is 40 bytes a problem (status, rows per lane, arena sizes), from which the
// host places the arenas and orders the launches exactly as the host-geometry path does.  Same kernels, same tables, same results.  Four
// sub-batches, two in flight, as in banded_align_pipelined; the geometry kernel runs on the side stream under the fills of the sub-batch before.
// Returns BANDED_NOT_HERE — nothing written — when a graph of the call has an empty node or a sub-batch does not fit the device budget.
struct GPrep { int32_t status; uint32_t N, E, bases, slot, arena; uint64_t in_bytes; bool on_device; };      // slot: among the sub-batch's gathered problems (tables, BGeomOut); arena: among those the geometry accepted (BProb, BResult); in_bytes: its share of the call's algorithmic bytes but for cells and ops

static int banded_align_device_geometry(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                                        vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    uint64_t budget = ctx->be->memory_bytes() / 4;
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    if (!budget) budget = 1ull << 30;
    Backend* be = ctx->be.get();
    const bool qa = ctx->has_qa;
    if (!ctx->banded_host) ctx->banded_host = std::make_shared<HostArenas>();
    HostArenas& H = *static_cast<HostArenas*>(ctx->banded_host.get());
    if (!H.be) { H.be = be; for (void*& e : H.ev) e = be->event_create(); }
    const bool timing = std::getenv("VGAMD_BANDED_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[vgk_banded_align/device geometry] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
    std::vector<GPrep> gp(n);
    std::atomic<int> not_here{0};                                 // a problem whose answer the host path's order of tests decides (below)
    // empty nodes anywhere: not this path (found before anything is written)
    { std::atomic<int> empty{0};
      parallel_chunks(n, [&](uint32_t lo, uint32_t hi, uint32_t) {
          for (uint32_t q = lo; q < hi && !empty.load(std::memory_order_relaxed); ++q) {
              const vgk_graph& g = problems[q].graph;
              if (!g.node_len) continue;
              for (uint32_t v = 0; v < g.n_nodes; ++v) if (g.node_len[v] == 0) { empty.store(1, std::memory_order_relaxed); break; }
          }
      });
      if (empty.load()) return BANDED_NOT_HERE; }
    lap("scan");
    ctx->banded_ms[0] = ctx->banded_ms[1] = 0; ctx->banded_cells = 0; ctx->banded_bytes = 0; ctx->banded_last_valid = false;
    int8_t* mat_rows = ctx->banded_mat_rows;
    std::memset(mat_rows, 0, BMAT_BYTES);
    if (!qa) { std::memcpy(mat_rows, ctx->sc.matrix, 25); for (int g = 0; g < 5; ++g) std::memcpy(mat_rows + BMAT_ROWS_AT + 8 * g, ctx->sc.matrix + 5 * g, 5); }
    const int8_t* mat = qa ? ctx->qmat.data() : mat_rows;
    uint32_t n_subs = 4;                                          // (VGAMD_BANDED_SUBS: 2 .. 16 sub-batches of a call)
    if (const char* e = std::getenv("VGAMD_BANDED_SUBS")) n_subs = (uint32_t)std::min(16, std::max(2, std::atoi(e)));
    const uint32_t quarter = (n + n_subs - 1u) / n_subs;
    size_t used = 0; int rc_all = VGK_OK;

    struct Sums { uint64_t v[6]; };                              // problems, nodes, edges, read bases, graph bases, (unused)
    // ---- first half of a sub-batch: sizes, gather + checks, geometry on the device, placement
    auto build = [&](uint32_t from, BSub& S, int set) -> int {
        PinnedSet& A = H.set[set]; HostArenas::GeomSet& G = H.gset[set]; const int gbase = set ? G_SET1 : G_SET0, base = set ? S_SET1 : 0;
        const uint32_t limit = std::min<uint32_t>(n, (from / quarter + 1u) * quarter), cnt = limit - from;
        S.i = from; S.j = limit; S.owner.clear(); S.launches.clear(); S.m = 0;
        const uint32_t n_chunks = chunk_count(cnt);
        std::vector<Sums> chunk(n_chunks + 1, Sums{});
        // what can be said before the arrays are walked (prepare()'s first tests, in its order) and the sizes
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Sums t{};
            for (uint32_t q = from + lo; q < from + hi; ++q) {
                const vgk_banded_problem& p = problems[q]; const vgk_graph& g = p.graph; GPrep& z = gp[q];
                z.status = VGK_OK; z.N = g.n_nodes; z.E = 0; z.bases = 0; z.slot = 0; z.arena = 0; z.in_bytes = 0; z.on_device = false;
                const uint32_t N = g.n_nodes;
                if (!N || !p.read_len || !p.read || !g.node_len || !g.pred_off || !g.seq || (qa && !p.qual)) { z.status = VGK_EINVAL; continue; }
                if (g.pred_off[N] > g.pred_off[0] && !g.pred_idx) { z.status = VGK_EINVAL; continue; }
                uint64_t total = 0; int st = VGK_OK;
                for (uint32_t v = 0; v < N && st == VGK_OK; ++v) {
                    if (g.pred_off[v + 1] < g.pred_off[v]) { st = VGK_EINVAL; break; }
                    for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) if (g.pred_idx[e] >= v) { st = VGK_EINVAL; break; }
                    if (st == VGK_OK && g.node_len[v] > 65535u) st = VGK_ETOOBIG;
                    total += g.node_len[v];
                }
                if (st == VGK_OK && p.read_len > 65535u) st = VGK_ETOOBIG;
                // (a padding of 2^26 diagonals and more, a graph of more than 2^24 bases: prepare() declines them, but after its band pass
                // — VGK_ETOOBIG or VGK_ENOBAND, whichever test comes first there; the lane's 32-bit diagonals need not see such a problem)
                // (a negative padding too: the host path's answer)
                if (st == VGK_OK && ((uint32_t)p.band_padding >= (1u << 26) || total > (1u << 24))) {
                    not_here.store(1, std::memory_order_relaxed); st = VGK_ETOOBIG;
                }
                if (st != VGK_OK) { z.status = st; continue; }
                z.E = g.pred_off[N] - g.pred_off[0]; z.bases = (uint32_t)total; z.on_device = true;
                z.in_bytes = p.read_len + total + 8ull * N + 4ull * g.pred_off[N] + 16;
                t.v[0] += 1; t.v[1] += N; t.v[2] += z.E; t.v[3] += p.read_len; t.v[4] += total;
            }
            chunk[c + 1] = t;
        });
        if (not_here.load()) return BANDED_NOT_HERE;
        for (uint32_t c = 0; c < n_chunks; ++c) for (int x = 0; x < 6; ++x) chunk[c + 1].v[x] += chunk[c].v[x];
        const Sums all = chunk[n_chunks];
        const uint32_t m = (uint32_t)all.v[0];
        if (!m) return VGK_OK;
        if (all.v[1] + m > 0xfffffff0ull || all.v[2] > 0xfffffff0ull || all.v[3] > 0xfffffff0ull || all.v[4] > 0xfffffff0ull) return BANDED_NOT_HERE;
        S.owner.resize(m);
        BGeomProb* gprobs = G.gprobs.get(be, m); uint32_t* node_len = G.node_len.get(be, all.v[1] + 1); uint32_t* pred_off = G.pred_off.get(be, all.v[1] + m + 1);
        uint32_t* pred_idx = G.pred_idx.get(be, all.v[2] + 1); BGeomOut* gout = G.gout.get(be, m);
        uint8_t* reads = A.reads.get(be, all.v[3] + 1); uint8_t* quals = qa ? A.quals.get(be, all.v[3] + 1) : nullptr; uint8_t* graph = A.graph.get(be, all.v[4] + 1);
        BProb* probs = A.probs.get(be, m); uint32_t* order = A.order.get(be, m);
        if (!gprobs || !node_len || !pred_off || !pred_idx || !gout || !reads || (qa && !quals) || !graph || !probs || !order) return VGK_ENOMEM;
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Sums at = chunk[c];
            for (uint32_t q = from + lo; q < from + hi; ++q) {
                GPrep& z = gp[q];
                if (!z.on_device) continue;
                const vgk_banded_problem& p = problems[q]; const vgk_graph& g = p.graph;
                const uint32_t a = (uint32_t)at.v[0], nb = (uint32_t)at.v[1], eb = (uint32_t)at.v[2];
                BGeomProb gb{}; gb.node_base = nb; gb.edge_base = eb; gb.n_nodes = z.N; gb.L = p.read_len; gb.band_padding = p.band_padding;
                gb.permissive = (p.flags & VGK_BANDED_PERMISSIVE) ? 1u : 0u; gb.max_cells = p.max_cells;
                gprobs[a] = gb;
                std::memcpy(node_len + nb, g.node_len, (size_t)z.N * 4);
                const uint32_t e0 = g.pred_off[0]; uint32_t* po = pred_off + nb + a;
                for (uint32_t v = 0; v <= z.N; ++v) po[v] = g.pred_off[v] - e0;
                if (z.E) std::memcpy(pred_idx + eb, g.pred_idx + e0, (size_t)z.E * 4);
                BProb pb{};
                pb.L = p.read_len; pb.n_nodes = z.N; pb.graph_len = z.bases; pb.node_base = nb; pb.seed_base = eb; pb.pool_base = 0; pb.start_base = nb;
                pb.read_off = (uint32_t)at.v[3]; pb.graph_off = (uint32_t)at.v[4]; pb.ops_cap = (uint32_t)(p.read_len + (uint64_t)z.bases + 2ull * z.N + 8);
                probs[a] = pb;                                      // (rows per lane, candidate ends and the arena offsets follow the geometry)
                z.slot = a; S.owner[a] = q;
                at.v[0] += 1; at.v[1] += z.N; at.v[2] += z.E; at.v[3] += p.read_len; at.v[4] += z.bases;
            }
        });
        // the geometry: raw arrays up on the side stream, a lane per problem behind them, the 40 bytes per problem back
        BGeomParams Q{}; Q.n = m;
        void* d_gprobs = ensure(ctx, gbase + G_PROBS, (uint64_t)m * sizeof(BGeomProb)); void* d_len = ensure(ctx, gbase + G_NODELEN, (all.v[1] + 1) * 4);
        void* d_poff = ensure(ctx, gbase + G_PREDOFF, (all.v[1] + m + 1) * 4); void* d_pidx = ensure(ctx, gbase + G_PREDIDX, (all.v[2] + 1) * 4);
        void* d_tmp = ensure(ctx, gbase + G_TMP, (all.v[1] + 1) * 12); void* d_out = ensure(ctx, gbase + G_OUT, (uint64_t)m * sizeof(BGeomOut));
        void* d_nodes = ensure(ctx, base + S_NODES, (all.v[1] + 1) * sizeof(BNode)); void* d_seeds = ensure(ctx, base + S_SEEDS, (all.v[2] + 1) * sizeof(BSeed));
        void* d_starts = ensure(ctx, base + S_STARTS, (all.v[1] + 1) * sizeof(BStart)); void* d_pool = ensure(ctx, base + S_POOL, 16);
        if (!d_gprobs || !d_len || !d_poff || !d_pidx || !d_tmp || !d_out || !d_nodes || !d_seeds || !d_starts || !d_pool) return VGK_ENOMEM;
        // (the coded reads and bases travel on the side stream as well, under the fills of the sub-batch before ...
        void* d_reads = ensure(ctx, base + S_READS, all.v[3] + 1); void* d_quals = qa ? ensure(ctx, base + S_QUALS, all.v[3] + 1) : nullptr; void* d_graph = ensure(ctx, base + S_GRAPH, all.v[4] + 1);
        if (!d_reads || (qa && !d_quals) || !d_graph) return VGK_ENOMEM;
        int rc;
        if ((rc = be->upload_side(d_gprobs, gprobs, (size_t)m * sizeof(BGeomProb))) || (rc = be->upload_side(d_len, node_len, (size_t)all.v[1] * 4)) ||
            (rc = be->upload_side(d_poff, pred_off, (size_t)(all.v[1] + m) * 4)) || (all.v[2] && (rc = be->upload_side(d_pidx, pred_idx, (size_t)all.v[2] * 4)))) return rc;
        Q.probs = (const BGeomProb*)d_gprobs; Q.node_len = (const uint32_t*)d_len; Q.pred_off = (const uint32_t*)d_poff; Q.pred_idx = (const uint32_t*)d_pidx;
        Q.tmp = (int32_t*)d_tmp; Q.nodes = (BNode*)d_nodes; Q.seeds = (BSeed*)d_seeds; Q.starts = (BStart*)d_starts; Q.out = (BGeomOut*)d_out;
        if ((rc = be->run_banded_geometry(Q))) return rc;
        lap("gather");
        // while that runs: the reads and the graphs' bases, coded
        parallel_for(m, [&](uint32_t a, unsigned) {
            const BProb& pb = probs[a]; const vgk_banded_problem& p = problems[S.owner[a]];
            code_bases<true>(reads + pb.read_off, p.read, pb.L);
            if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
            code_bases<true>(graph + pb.graph_off, p.graph.seq, pb.graph_len);
        });
        lap("bases");
        if ((rc = be->download_side(gout, d_out, (size_t)m * sizeof(BGeomOut)))) return rc;
        // (... behind the geometry's answer, so that the host does not wait for them; the fills do: main_after_side in launch)
        if ((rc = be->upload_side(d_reads, reads, (size_t)all.v[3])) || (qa && (rc = be->upload_side(d_quals, quals, (size_t)all.v[3]))) || (rc = be->upload_side(d_graph, graph, (size_t)all.v[4]))) return rc;
        lap("geometry");
        // placement: the problems the geometry accepted keep their slots in the tables; traceback bytes, last columns and op slots are laid out
        // behind each other in their order (serial: three running sums over 40-byte records)
        // (two passes over chunks of slots on the host threads: the accepted problems' sums, then every chunk places its own)
        struct Place { uint64_t kept, tb, last, ops; };
        const uint32_t p_chunks = chunk_count(m);
        std::vector<Place> pl(p_chunks + 1, Place{0, 0, 0, 0});
        parallel_chunks(m, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Place t{0, 0, 0, 0};
            for (uint32_t a = lo; a < hi; ++a) { const BGeomOut& o = gout[a]; if (o.status != VGK_OK) continue; t.kept += 1; t.tb += o.tb_bytes; t.last += o.last_elems; t.ops += probs[a].ops_cap; }
            pl[c + 1] = t;
        });
        for (uint32_t c = 0; c < p_chunks; ++c) { pl[c + 1].kept += pl[c].kept; pl[c + 1].tb += pl[c].tb; pl[c + 1].last += pl[c].last; pl[c + 1].ops += pl[c].ops; }
        const uint64_t tb_bytes = pl[p_chunks].tb, last_elems = pl[p_chunks].last, ops_total = pl[p_chunks].ops; const uint32_t kept = (uint32_t)pl[p_chunks].kept;
        std::vector<uint32_t> keys(std::max<uint32_t>(kept, 1));
        BProb* placed = A.probs2.get(be, std::max<uint32_t>(kept, 1));      // (the accepted problems move up: the kernels count problems, not slots)
        if (!placed) return VGK_ENOMEM;
        parallel_chunks(m, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Place at = pl[c];
            for (uint32_t a = lo; a < hi; ++a) {
                const BGeomOut& o = gout[a]; GPrep& z = gp[S.owner[a]];
                if (o.status != VGK_OK) { z.status = o.status; z.on_device = false; continue; }
                BProb pb = probs[a];
                pb.Hpad = 64u * o.R; pb.n_starts = o.n_starts; pb.tb_base = at.tb; pb.last_base = at.last; pb.ops_off = at.ops;
                const uint32_t k = (uint32_t)at.kept;
                placed[k] = pb; z.arena = k; keys[k] = o.order_key; order[k] = k;
                at.kept += 1; at.tb += o.tb_bytes; at.last += o.last_elems; at.ops += pb.ops_cap;
            }
        });
        probs = placed;
        if (tb_bytes + last_elems * 4 + ops_total * 2 * sizeof(vgk_op) > budget) return BANDED_NOT_HERE;
        // launches: one per rows-per-lane class; inside a class the problems with the most cells first (counting sort on log2(cells))
        { std::vector<uint32_t> count(B_ROW_CLASSES * 64 + 1, 0), sorted(kept);
          for (uint32_t k = 0; k < kept; ++k) ++count[keys[k] + 1];
          for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
          std::vector<uint32_t> at(count.begin(), count.end() - 1);
          for (uint32_t k = 0; k < kept; ++k) sorted[at[keys[k]]++] = order[k];
          std::copy(sorted.begin(), sorted.end(), order);
          for (uint32_t r = 0; r < B_ROW_CLASSES; ++r) {
            const uint32_t lo = count[r * 64], hi = count[(r + 1) * 64];
            if (lo == hi) continue;
            uint64_t lds = 0;
            for (uint32_t b = lo; b < hi; ++b) { const BProb& pb = probs[order[b]]; lds = std::max<uint64_t>(lds, (qa ? 6400u : 32u) + (uint64_t)pb.L * (qa ? 2 : 1) + pb.graph_len + 96); }
            S.launches.push_back({1u << r, lo, hi - lo, lds <= 40 * 1024 ? (uint32_t)lds : 0u});
          } }
        S.m = m;
        S.sizes[0] = all.v[1]; S.sizes[1] = all.v[2]; S.sizes[2] = 0; S.sizes[3] = all.v[1]; S.sizes[4] = all.v[3]; S.sizes[5] = all.v[4]; S.sizes[6] = tb_bytes; S.sizes[7] = last_elems; S.sizes[8] = ops_total;
        BandedParams& P = S.P; P = BandedParams{};
        P.nodes = (const BNode*)d_nodes; P.seeds = (const BSeed*)d_seeds; P.pool = (const uint32_t*)d_pool; P.starts = (const BStart*)d_starts;
        P.reads = (const uint8_t*)d_reads; P.quals = (const uint8_t*)d_quals; P.graph = (const uint8_t*)d_graph;
        P.n = kept;
        lap("placement");
        return VGK_OK;
    };
    // ---- ... the rest up on the main stream, the fills and the tracebacks behind it
    auto launch = [&](BSub& S, int set) -> int {
        if (!S.m) return VGK_OK;
        PinnedSet& A = H.set[set]; const int base = set ? S_SET1 : 0;
        BandedParams& P = S.P;
        const uint64_t tb_bytes = S.sizes[6], last_elems = S.sizes[7], ops_total = S.sizes[8];
        int rc;
        if ((rc = stage(ctx, base + S_PROBS, (const BProb*)A.probs2.p, std::max<uint32_t>(P.n, 1), P.probs)) || (rc = stage(ctx, base + S_ORDER, (const uint32_t*)A.order.p, std::max<uint32_t>(P.n, 1), P.order)) ||
            (rc = stage(ctx, base + S_MAT, mat, qa ? 6400 : BMAT_BYTES, P.mat))) return rc;
        P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend;
        P.tb = (uint8_t*)ensure(ctx, base + S_TB, std::max<uint64_t>(tb_bytes, 256));
        P.last = (int32_t*)ensure(ctx, base + S_LAST, std::max<uint64_t>(last_elems, 64) * sizeof(int32_t));
        P.ops = (vgk_op*)ensure(ctx, base + S_OPS, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
        P.dense = (vgk_op*)ensure(ctx, base + S_DENSE, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
        uint8_t* rblock = (uint8_t*)ensure(ctx, base + S_RESULTS, (size_t)S.m * sizeof(BResult) + 64);
        if (!P.tb || !P.last || !P.ops || !P.dense || !rblock) return VGK_ENOMEM;
        P.dense_count = (unsigned long long*)rblock; P.results = (BResult*)(rblock + 64);
        if ((rc = be->zero(rblock, 64)) || (rc = be->main_after_side())) return rc;
        if (P.n && (rc = be->run_banded_async(P, S.launches.data(), (uint32_t)S.launches.size(), set))) return rc;
        if ((rc = be->event_record(H.ev[set]))) return rc;
        lap("launch");
        return VGK_OK;
    };
    // ---- second half: results and packed ops back behind the event, then the caller's order
    auto finish = [&](BSub& S, int set) -> int {
        PinnedSet& A = H.set[set]; HostArenas::GeomSet& G = H.gset[set];
        const uint32_t m = S.m, i = S.i, j = S.j; const BandedParams& P = S.P;
        BResult* dres = A.dres.get(be, m + 1); const vgk_op* dops = nullptr; const BGeomOut* gout = G.gout.p;
        if (!dres) return VGK_ENOMEM;
        if (m && P.n) {
            int rc;
            if (H.ev[set]) { if ((rc = be->fetch_after(H.ev[set]))) return rc; }
            else if ((rc = be->sync())) return rc;
            unsigned long long* dense_n = A.count.get(be, 8);
            if (!dense_n) return VGK_ENOMEM;
            if ((rc = be->download_fetch_async(dense_n, P.dense_count, sizeof(unsigned long long)))) return rc;
            if ((rc = be->download_fetch_async(dres, P.results, (size_t)m * sizeof(BResult)))) return rc;
            if ((rc = be->sync_fetch())) return rc;
            vgk_op* hd = A.dops.get(be, dense_n[0] + 1);
            if (!hd) return VGK_ENOMEM;
            if (dense_n[0] && (rc = be->download_fetch(hd, P.dense, (size_t)dense_n[0] * sizeof(vgk_op)))) return rc;
            dops = hd;
            ctx->banded_ms[0] += be->banded_ms(set, 0); ctx->banded_ms[1] += be->banded_ms(set, 1);
            lap("fetch");
        }
        // statuses, scores and op counts per problem with the chunks' sums beside them; when everything fits the caller's op buffer (the
        // usual case) a problem's slice follows from the sums and the copy-out is one parallel pass, otherwise the running sum decides problem
        // by problem which ones still fit, as the host-geometry path does
        const uint32_t cnt = j - i, n_chunks = chunk_count(cnt);
        std::vector<uint32_t> need(cnt, 0);
        struct Tot { uint64_t ops, cells, bytes; };
        std::vector<Tot> tot(n_chunks + 1, Tot{0, 0, 0});
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            Tot t{0, 0, 0};
            for (uint32_t k = lo; k < hi; ++k) {
                const uint32_t q = i + k; const GPrep& z = gp[q]; vgk_result& r = results[q];
                std::memset(&r, 0, sizeof r);
                if (!z.on_device) { r.status = z.status; continue; }
                const BResult& dr = dres[z.arena];
                t.cells += gout[z.slot].cells; t.bytes += z.in_bytes + gout[z.slot].cells + 2ull * dr.n_ops;
                if (dr.status != VGK_OK) { r.status = dr.status; continue; }
                need[k] = dr.n_ops; r.score = dr.score; r.status = VGK_OK; t.ops += dr.n_ops;
            }
            tot[c + 1] = t;
        });
        for (uint32_t c = 0; c < n_chunks; ++c) { tot[c + 1].ops += tot[c].ops; tot[c + 1].cells += tot[c].cells; tot[c + 1].bytes += tot[c].bytes; }
        ctx->banded_cells += tot[n_chunks].cells; ctx->banded_bytes += tot[n_chunks].bytes;
        const bool all_fit = ops && used + tot[n_chunks].ops <= ops_cap;
        if (!all_fit) for (uint32_t q = i; q < j; ++q) {
            vgk_result& r = results[q];
            r.ops_begin = (uint32_t)used;
            if (r.status != VGK_OK) continue;
            if (!ops || used + need[q - i] > ops_cap) { r.status = VGK_EOPS; rc_all = VGK_EOPS; need[q - i] = 0; continue; }
            r.n_ops = need[q - i]; used += need[q - i];
        }
        parallel_chunks(cnt, [&](uint32_t lo, uint32_t hi, uint32_t c) {
            uint64_t at = used + tot[c].ops;
            for (uint32_t k = lo; k < hi; ++k) {
                const uint32_t q = i + k; vgk_result& r = results[q];
                if (all_fit) { r.ops_begin = (uint32_t)at; if (r.status == VGK_OK) { r.n_ops = need[k]; at += need[k]; } }
                if (r.status != VGK_OK || !r.n_ops) continue;
                const BResult& dr = dres[gp[q].arena];
                const vgk_op* src = dops + dr.ops_begin; vgk_op* out = ops + r.ops_begin;
                for (uint32_t e = 0; e < dr.n_ops; ++e) { vgk_op o = src[e]; if (o.len == 0) o.op = VGK_OP_M; *out++ = o; }
            }
        });
        if (all_fit) used += tot[n_chunks].ops;
        lap("results");
        return VGK_OK;
    };

    BSub subs[2]; int set = 0; bool pending = false; int rc = VGK_OK;
    for (uint32_t i = 0; i < n && rc == VGK_OK;) {
        BSub& S = subs[set];
        rc = build(i, S, set);
        i = S.j;
        // (launched BEFORE the results of the sub-batch before are fetched: what this path sends up behind the geometry is a few hundred
        // kilobytes of descriptors, nothing a download would queue behind — the device never waits for the host's copy-out)
        if (rc == VGK_OK) rc = launch(S, set);
        if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
        pending = rc == VGK_OK;
        set ^= 1;
    }
    if (pending && rc == VGK_OK) rc = finish(subs[set ^ 1], set ^ 1);
    if (rc != VGK_OK) { be->sync_side(); be->sync(); be->sync_fetch(); return rc; }
    if (ops_written) *ops_written = used;
    return rc_all;
}

int vgk_banded_align(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                     vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) try {
    uint32_t pipeline_from = 32768u;                              // (VGAMD_BANDED_PIPELINE_MIN: tests cut small calls in four as well)
    if (const char* e = std::getenv("VGAMD_BANDED_PIPELINE_MIN")) pipeline_from = (uint32_t)std::max(4, std::atoi(e));
    if (ctx && problems && results && n >= pipeline_from && !std::getenv("VGAMD_BANDED_ONE_BATCH")) {
        // graphs without empty nodes: their geometry on the device (VGAMD_BANDED_HOST_GEOMETRY=1: on the host threads, as for every other call)
        if (!std::getenv("VGAMD_BANDED_HOST_GEOMETRY")) {
            const int rc = banded_align_device_geometry(ctx, problems, n, results, ops, ops_cap, ops_written);
            if (rc != BANDED_NOT_HERE) return rc;
        }
        return banded_align_pipelined(ctx, problems, n, results, ops, ops_cap, ops_written);
    }
    return banded_align_impl(ctx, problems, n, 0, results, nullptr, ops, ops_cap, ops_written);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_banded_align_multi(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n, uint32_t max_alt_alns,
                           vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written) try {
    if (!max_alt_alns) return VGK_EINVAL;
    return banded_align_impl(ctx, problems, n, max_alt_alns, results, n_alignments, ops, ops_cap, ops_written);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_banded_rerun(vgk_ctx* ctx) try {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->banded_last_valid) return VGK_EINVAL;
    Backend* be = ctx->be.get();
    int rc;
    if ((rc = be->zero(ctx->banded_last.dense_count, 64))) return rc;
    if ((rc = be->run_banded(ctx->banded_last, ctx->banded_last_launches.data(), (uint32_t)ctx->banded_last_launches.size()))) return rc;
    ctx->banded_ms[0] = be->last_ms(3); ctx->banded_ms[1] = be->last_ms(4);
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_banded_last(vgk_ctx* ctx, int which) {
    if (!ctx) return 0.0;
    switch (which) { case 0: return ctx->banded_ms[0]; case 1: return ctx->banded_ms[1];
                     case 2: return (double)ctx->banded_cells; case 3: return (double)ctx->banded_bytes; default: return 0.0; }
}

}  // extern "C"
