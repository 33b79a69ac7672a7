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
    bool on_device = false;
    uint32_t R = 1, Hpad = 64, thread = 0;
    uint64_t cells = 0, bases = 0, tb_bytes = 0, last_elems = 0;
    uint32_t ops_cap = 0;
    Span nodes, seeds, pool, starts;   // starts: candidate end nodes; start_prefix[k] = the empty sink-side nodes of candidate k, sink first (:2455-2480)
    bool have_empty_walk = false;      // a source-to-sink chain of empty nodes (:2464-2472)
    Span empty_walk;                   // in Store::prefix, sink first
    uint32_t arena = 0;                // index inside its sub-batch
    uint32_t need = 0;                 // ops this problem hands back
    bool use_empty_walk = false;
};
struct Store {                         // per-thread flat tables of pass 1
    std::vector<BNode> nodes; std::vector<BSeed> seeds; std::vector<uint32_t> pool, starts, prefix;
    std::vector<Span> start_prefix;    // parallel to starts
};

struct Scratch {                       // per-thread reusable buffers of pass 1
    std::vector<int64_t> len, shortest, longest, top, bot, cum;
    std::vector<uint8_t> masked;
    std::vector<uint32_t> succ_off, succ, fill;
    struct Item { uint32_t node, path_off, path_len; };
    std::vector<Item> stack;
    std::vector<int64_t> st; std::vector<uint32_t> path;
};

// Band geometry of one problem (find_banded_paths :2174-2268, path_lengths_to_sinks :2122-2170, shortest_seq_paths :2271-2293)
// and the tables the kernels need.
void prepare(const vgk_ctx* ctx, const vgk_banded_problem& p, Prep& hp, Scratch& S, Store& T) {
    const vgk_graph& g = p.graph;
    const uint32_t N = g.n_nodes; const int64_t L = p.read_len;
    if (!N || !L || !p.read || !g.node_len || !g.pred_off || !g.seq || (ctx->has_qa && !p.qual)) { hp.status = VGK_EINVAL; return; }
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) if (g.pred_idx[e] >= v) { hp.status = VGK_EINVAL; return; }
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
    if (R > 16 || L > (1 << 24) || total_bases > (1u << 24)) { hp.status = VGK_ETOOBIG; return; }      // engine limit: bands up to 1024 diagonals
    hp.R = R; hp.Hpad = 64 * R;
    const uint32_t Hpad = hp.Hpad;

    // node records with the flattened predecessor lists
    uint64_t tb_off = 0, last_off = 0; uint32_t seq_off = 0;
    int64_t prev_filled = -1;              // the node the wave will have in its registers when it reaches v
    const size_t keep_nodes = T.nodes.size(), keep_seeds = T.seeds.size(), keep_pool = T.pool.size();
    auto fail = [&](int code) { T.nodes.resize(keep_nodes); T.seeds.resize(keep_seeds); T.pool.resize(keep_pool); hp.status = code; };
    T.nodes.resize(keep_nodes + N);
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
            if (!nd.chain) for (uint32_t q = 0; q < n_seeds; ++q) T.nodes[keep_nodes + T.seeds[T.seeds.size() - 1 - q].node].keep_last = 1;
            const uint32_t granule = std::max<uint32_t>(R, 4), H = (uint32_t)(bot[v] - top[v] + 1);
            nd.stride = (H + granule - 1) / granule * granule;
            nd.tb_off = (uint32_t)tb_off;
            tb_off += (uint64_t)len[v] * nd.stride;
            if (tb_off > 0xfffffff0ull) { fail(VGK_ETOOBIG); return; }
        }
        T.nodes[keep_nodes + v] = nd;
    }
    hp.nodes = {keep_nodes, N}; hp.seeds = {keep_seeds, (uint32_t)(T.seeds.size() - keep_seeds)}; hp.pool = {keep_pool, (uint32_t)(T.pool.size() - keep_pool)};
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
                    if (!hp.have_empty_walk) { hp.have_empty_walk = true; hp.empty_walk = {T.prefix.size(), (uint32_t)S.path.size()}; T.prefix.insert(T.prefix.end(), S.path.begin(), S.path.end()); }
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
    for (uint32_t q = 0; q < hp.starts.len; ++q) T.nodes[keep_nodes + T.starts[hp.starts.off + q]].keep_last = 1;
    // last / first columns only where a traceback or a successor will read them
    for (uint32_t v = 0; v < N; ++v) {
        BNode& nd = T.nodes[keep_nodes + v];
        if (nd.masked || nd.len == 0 || (nd.chain && !nd.keep_last)) continue;
        nd.last_off = (uint32_t)last_off; last_off += 5ull * nd.stride;
    }
    hp.tb_bytes = (tb_off + 255) & ~255ull; hp.last_elems = last_off;
    hp.ops_cap = (uint32_t)(L + total_bases + 2ull * N + 8);
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
template <class T> struct RawBuf {
    T* p = nullptr; size_t cap = 0;
    T* get(size_t n) { if (n > cap) { std::free(p); cap = n + n / 4 + 64; p = (T*)std::malloc(cap * sizeof(T)); } return p; }
    ~RawBuf() { std::free(p); }
};
struct HostArenas {
    RawBuf<BProb> probs; RawBuf<BNode> nodes; RawBuf<BSeed> seeds; RawBuf<uint32_t> pool, order; RawBuf<BStart> starts;
    RawBuf<uint8_t> reads, quals, graph; RawBuf<BResult> dres; RawBuf<vgk_op> dops;
};
enum { S_PROBS, S_ORDER, S_NODES, S_SEEDS, S_POOL, S_STARTS, S_READS, S_QUALS, S_GRAPH, S_MAT, S_TB, S_LAST, S_OPS, S_DENSE, S_RESULTS, S_COUNT };
static_assert(S_COUNT <= sizeof(vgk_ctx::scratch) / sizeof(vgk_ctx::DevBuf), "scratch slots");

}  // namespace

extern "C" {

int vgk_banded_align(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                     vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
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
    std::vector<Scratch> scratch(MAX_THREADS); std::vector<Store> store(MAX_THREADS);
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
            if (!owner.empty() && (tb_bytes + hp.tb_bytes + (last_elems + hp.last_elems) * 4 + (ops_total + hp.ops_cap) * 2 * sizeof(vgk_op) > budget ||
                                   n_nodes + p.graph.n_nodes > 0xfffffff0ull || n_read + p.read_len > 0xfffffff0ull || n_graph + hp.bases > 0xfffffff0ull)) break;
            n_nodes += p.graph.n_nodes; n_seeds += hp.seeds.len; n_pool += hp.pool.len; n_starts += hp.starts.len;
            n_read += p.read_len; n_graph += hp.bases; tb_bytes += hp.tb_bytes; last_elems += hp.last_elems; ops_total += hp.ops_cap;
            owner.push_back(j);
        }
        const uint32_t m = (uint32_t)owner.size();
        BProb* probs = H.probs.get(m);
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
        BResult* dres = H.dres.get(m + 1);
        const vgk_op* dops = nullptr;
        if (m) {
            // pass 3: copy into the arenas
            BNode* nodes = H.nodes.get(n_nodes); BSeed* seeds = H.seeds.get(n_seeds); uint32_t* pool = H.pool.get(n_pool); BStart* starts = H.starts.get(n_starts);
            uint8_t* reads = H.reads.get(n_read); uint8_t* quals = qa ? H.quals.get(n_read) : nullptr; uint8_t* graph = H.graph.get(n_graph);
            parallel_for(m, [&](uint32_t a, unsigned) {
                const Prep& hp = hps[owner[a]]; const BProb& pb = probs[a]; const vgk_banded_problem& p = problems[owner[a]];
                const Store& T = store[hp.thread];
                std::copy(T.nodes.begin() + hp.nodes.off, T.nodes.begin() + hp.nodes.off + hp.nodes.len, nodes + pb.node_base);
                std::copy(T.seeds.begin() + hp.seeds.off, T.seeds.begin() + hp.seeds.off + hp.seeds.len, seeds + pb.seed_base);
                std::copy(T.pool.begin() + hp.pool.off, T.pool.begin() + hp.pool.off + hp.pool.len, pool + pb.pool_base);
                for (uint32_t q = 0; q < hp.starts.len; ++q) starts[pb.start_base + q].node = T.starts[hp.starts.off + q];
                uint8_t* rd = reads + pb.read_off;
                for (uint32_t q = 0; q < pb.L; ++q) rd[q] = nt_code(p.read[q]);
                if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
                uint8_t* gr = graph + pb.graph_off;
                for (uint32_t q = 0; q < pb.graph_len; ++q) gr[q] = nt_code(p.graph.seq[q]);
            });
            lap("arenas");
            // launches: one per rows-per-lane class; inside a class the problems with the most cells first (counting sort on log2(cells))
            uint32_t* order = H.order.get(m);
            std::vector<BandedLaunch> launches;
            {
                auto key = [&](uint32_t a) { const Prep& hp = hps[owner[a]]; uint32_t r = 0; while ((1u << r) < hp.R) ++r;
                                             uint32_t lg = 0; while ((hp.cells >> lg) > 1 && lg < 63) ++lg; return r * 64 + (63 - lg); };
                std::vector<uint32_t> count(5 * 64 + 1, 0);
                for (uint32_t a = 0; a < m; ++a) ++count[key(a) + 1];
                for (size_t k = 1; k < count.size(); ++k) count[k] += count[k - 1];
                std::vector<uint32_t> at(count.begin(), count.end() - 1);
                for (uint32_t a = 0; a < m; ++a) order[at[key(a)]++] = a;
                for (uint32_t r = 0; r < 5; ++r) {
                    const uint32_t lo = count[r * 64], hi = count[(r + 1) * 64];
                    if (lo == hi) continue;
                    // LDS staging area: score table | read codes | qualities | graph codes of the largest problem of the launch
                    uint64_t lds = 0;
                    for (uint32_t b = lo; b < hi; ++b) { const BProb& pb = probs[order[b]]; lds = std::max<uint64_t>(lds, (qa ? 6400u : 32u) + (uint64_t)pb.L * (qa ? 2 : 1) + pb.graph_len + 16); }
                    launches.push_back({1u << r, lo, hi - lo, lds <= 40 * 1024 ? (uint32_t)lds : 0u});
                }
            }
            lap("sort");
            BandedParams P{};
            int rc;
            const int8_t* mat = qa ? ctx->qmat.data() : ctx->sc.matrix;
            if ((rc = stage(ctx, S_PROBS, (const BProb*)probs, m, P.probs)) || (rc = stage(ctx, S_ORDER, (const uint32_t*)order, m, P.order)) ||
                (rc = stage(ctx, S_NODES, (const BNode*)nodes, n_nodes, P.nodes)) || (rc = stage(ctx, S_SEEDS, (const BSeed*)seeds, n_seeds, P.seeds)) ||
                (rc = stage(ctx, S_POOL, (const uint32_t*)pool, n_pool, P.pool)) || (rc = stage(ctx, S_STARTS, (const BStart*)starts, n_starts, P.starts)) ||
                (rc = stage(ctx, S_READS, (const uint8_t*)reads, n_read, P.reads)) || (qa && (rc = stage(ctx, S_QUALS, (const uint8_t*)quals, n_read, P.quals))) ||
                (rc = stage(ctx, S_GRAPH, (const uint8_t*)graph, n_graph, P.graph)) || (rc = stage(ctx, S_MAT, mat, qa ? 6400 : 25, P.mat))) return rc;
            lap("h2d");
            P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend; P.n = m;
            P.tb = (uint8_t*)ensure(ctx, S_TB, std::max<uint64_t>(tb_bytes, 256));
            P.last = (int32_t*)ensure(ctx, S_LAST, std::max<uint64_t>(last_elems, 64) * sizeof(int32_t));
            P.ops = (vgk_op*)ensure(ctx, S_OPS, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
            P.dense = (vgk_op*)ensure(ctx, S_DENSE, std::max<uint64_t>(ops_total, 1) * sizeof(vgk_op));
            uint8_t* rblock = (uint8_t*)ensure(ctx, S_RESULTS, (size_t)m * sizeof(BResult) + 64);
            if (!P.tb || !P.last || !P.ops || !P.dense || !rblock) return VGK_ENOMEM;
            P.dense_count = (unsigned long long*)rblock; P.results = (BResult*)(rblock + 64);
            if ((rc = be->zero(rblock, 64))) return rc;
            lap("scratch");
            if ((rc = be->run_banded(P, launches.data(), (uint32_t)launches.size()))) return rc;
            lap("kernels");
            ctx->banded_last = P; ctx->banded_last_launches = launches; ctx->banded_last_valid = (i == 0 && j == n);     // the whole call in one sub-batch
            unsigned long long dense_n = 0;
            if ((rc = be->download(&dense_n, P.dense_count, sizeof dense_n))) return rc;
            if ((rc = be->download(dres, P.results, (size_t)m * sizeof(BResult)))) return rc;
            vgk_op* hd = H.dops.get(dense_n + 1);
            if (dense_n && (rc = be->download(hd, P.dense, (size_t)dense_n * sizeof(vgk_op)))) return rc;
            dops = hd;
            ctx->banded_ms[0] += be->last_ms(3); ctx->banded_ms[1] += be->last_ms(4);
            lap("d2h");
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
            if (hp.on_device) {
                const vgk_banded_problem& p = problems[q];
                ctx->banded_cells += hp.cells;
                ctx->banded_bytes += p.read_len + hp.bases + 8ull * p.graph.n_nodes + 4ull * p.graph.pred_off[p.graph.n_nodes] + hp.cells + 16 + 2ull * dres[hp.arena].n_ops;
            }
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

int vgk_banded_rerun(vgk_ctx* ctx) {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->banded_last_valid) return VGK_EINVAL;
    Backend* be = ctx->be.get();
    int rc;
    if ((rc = be->zero(ctx->banded_last.dense_count, 64))) return rc;
    if ((rc = be->run_banded(ctx->banded_last, ctx->banded_last_launches.data(), (uint32_t)ctx->banded_last_launches.size()))) return rc;
    ctx->banded_ms[0] = be->last_ms(3); ctx->banded_ms[1] = be->last_ms(4);
    return VGK_OK;
}

double vgk_banded_last(vgk_ctx* ctx, int which) {
    if (!ctx) return 0.0;
    switch (which) { case 0: return ctx->banded_ms[0]; case 1: return ctx->banded_ms[1];
                     case 2: return (double)ctx->banded_cells; case 3: return (double)ctx->banded_bytes; default: return 0.0; }
}

}  // extern "C"
