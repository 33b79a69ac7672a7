// banded_api.cpp — vgk_banded_align: the host half of banded global alignment.
//
// What BandedGlobalAligner's constructor does per call on the CPU (reference:
// src/banded_global_aligner.cpp:1961-2110 — band ends, masking, cell budget, shortest lead
// sequences, one BAMatrix per node with its seed pointers) becomes flat tables for the wavefront
// kernels in banded_device.hpp; the predecessor lists are flattened here, once, in the LIFO order the
// reference's fill and traceback pop them, with the empty nodes each one is reached through.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include "ctx.hpp"

using namespace vgk;

namespace {

inline uint8_t nt_code(char ch) {      // gssw_create_nt_table: case-insensitive ACGT, everything else N
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct HostProblem {                   // what the host keeps of a problem until the results come back
    int status = VGK_OK;
    bool on_device = false;
    std::vector<std::vector<uint32_t>> start_prefix;   // per start candidate: empty sink-side nodes, sink first (:2455-2480)
    bool have_empty_walk = false;                       // a source-to-sink chain of empty nodes (:2464-2472)
    std::vector<uint32_t> empty_walk;                   // sink first
    uint32_t R = 1;
    uint64_t cells = 0;
};

struct Arena {
    std::vector<BProb> probs; std::vector<BNode> nodes; std::vector<BSeed> seeds; std::vector<uint32_t> pool;
    std::vector<BStart> starts; std::vector<uint8_t> reads, quals, graph;
    uint64_t tb_bytes = 0, last_elems = 0, ops_total = 0;
    std::vector<uint32_t> owner;       // arena problem -> index into the caller's array
};

// Band geometry of one problem (find_banded_paths :2174-2268, path_lengths_to_sinks :2122-2170, shortest_seq_paths :2271-2293)
// and the tables the kernels need.  Returns the per-problem status; appends to the arena only when the problem runs.
int prepare(const vgk_ctx* ctx, const vgk_banded_problem& p, HostProblem& hp, Arena& A) {
    const vgk_graph& g = p.graph;
    const uint32_t N = g.n_nodes; const int64_t L = p.read_len;
    if (!N || !L || !p.read || !g.node_len || !g.pred_off || (!g.seq && N)) return VGK_EINVAL;
    if (ctx->has_qa && !p.qual) return VGK_EINVAL;
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) if (g.pred_idx[e] >= v) return VGK_EINVAL;
    std::vector<std::vector<uint32_t>> succ(N);
    for (uint32_t v = 0; v < N; ++v) for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) succ[g.pred_idx[e]].push_back(v);
    auto is_source = [&](uint32_t v) { return g.pred_off[v] == g.pred_off[v + 1]; };
    const int64_t inf = std::numeric_limits<int64_t>::max();
    std::vector<int64_t> len(N), shortest(N), longest(N, 0), top(N, inf), bot(N, std::numeric_limits<int64_t>::min()), cum(N);
    std::vector<uint8_t> masked(N, 0);
    uint64_t total_bases = 0;
    for (uint32_t v = 0; v < N; ++v) { len[v] = g.node_len[v]; total_bases += g.node_len[v]; shortest[v] = succ[v].empty() ? 0 : inf; }
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
        for (uint32_t w : succ[v]) { top[w] = std::min(top[w], et); bot[w] = std::max(bot[w], eb); }
        cells += (uint64_t)(bot[v] - top[v] + 1) * (uint64_t)len[v];
        if (len[v]) max_h = std::max(max_h, bot[v] - top[v] + 1);
    }
    hp.cells = cells;
    if (p.max_cells && cells > p.max_cells) return VGK_ETOOBIG;
    for (uint32_t v = 0; v < N; ++v) cum[v] = is_source(v) ? 0 : inf;
    for (uint32_t v = 0; v < N; ++v) for (uint32_t w : succ[v]) cum[w] = std::min(cum[w], cum[v] + len[v]);
    if (!permissive) {
        bool any = false;
        for (uint32_t v = 0; v < N; ++v) if (succ[v].empty() && !masked[v]) any = true;
        if (!any) return VGK_ENOBAND;
    }
    uint32_t R = 1; while ((int64_t)R * 64 < max_h) R *= 2;
    if (R > 16 || L > (1 << 24) || total_bases > (1u << 24)) return VGK_ETOOBIG;       // engine limit: bands up to 1024 diagonals
    hp.R = R;
    const uint32_t Hpad = 64 * R;

    BProb pb{};
    pb.L = (uint32_t)L; pb.n_nodes = N; pb.Hpad = Hpad;
    pb.node_base = (uint32_t)A.nodes.size(); pb.seed_base = (uint32_t)A.seeds.size(); pb.pool_base = (uint32_t)A.pool.size();
    pb.start_base = (uint32_t)A.starts.size(); pb.read_off = (uint32_t)A.reads.size(); pb.graph_off = (uint32_t)A.graph.size();
    pb.tb_base = A.tb_bytes; pb.last_base = A.last_elems;
    if (A.nodes.size() + N > 0xfffffff0u || A.reads.size() + (uint64_t)L > 0xfffffff0u || A.graph.size() + total_bases > 0xfffffff0u) return VGK_ETOOBIG;

    // flattened predecessor lists
    const size_t keep_nodes = A.nodes.size(), keep_seeds = A.seeds.size();
    auto fail = [&](int code) { A.nodes.resize(keep_nodes); A.seeds.resize(keep_seeds); return code; };
    uint64_t tb_off = 0, last_off = 0; uint32_t seq_off = 0;
    std::vector<uint32_t> pool_local;       // relative to pool_base; slot 0.. hold paths
    struct Item { uint32_t node, path_off, path_len; };
    std::vector<Item> stack;
    for (uint32_t v = 0; v < N; ++v) {
        BNode nd{};
        nd.top = masked[v] ? 0 : (int32_t)top[v]; nd.bot = masked[v] ? -1 : (int32_t)bot[v];
        nd.len = (int32_t)len[v]; nd.cum = masked[v] || cum[v] == inf ? 0 : (int32_t)cum[v];
        nd.seq_off = seq_off; seq_off += (uint32_t)len[v];
        nd.masked = masked[v];
        nd.seed_off = (uint32_t)(A.seeds.size() - pb.seed_base);
        if (!masked[v] && len[v]) {
            nd.as_source = is_source(v);
            stack.clear();
            for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) stack.push_back({g.pred_idx[e], 0, 0});
            uint32_t n_seeds = 0;
            while (!stack.empty()) {
                const Item it = stack.back(); stack.pop_back();
                if (masked[it.node]) continue;
                if (len[it.node] == 0) {
                    const uint32_t noff = (uint32_t)pool_local.size();
                    for (uint32_t q = 0; q < it.path_len; ++q) pool_local.push_back(pool_local[it.path_off + q]);
                    pool_local.push_back(it.node);
                    if (is_source(it.node)) { nd.as_source = 1; nd.src_path_off = noff; nd.src_path_len = it.path_len + 1; }
                    for (uint32_t e = g.pred_off[it.node]; e < g.pred_off[it.node + 1]; ++e) stack.push_back({g.pred_idx[e], noff, it.path_len + 1});
                    continue;
                }
                A.seeds.push_back({it.node, it.path_off, it.path_len}); ++n_seeds;
            }
            if (n_seeds > 0xffff) return fail(VGK_ETOOBIG);
            nd.n_seeds = (uint16_t)n_seeds;
            nd.tb_off = (uint32_t)tb_off; nd.last_off = (uint32_t)last_off;
            tb_off += (uint64_t)len[v] * Hpad; last_off += 3ull * Hpad;
            if (tb_off > 0xfffffff0ull) return fail(VGK_ETOOBIG);
        }
        A.nodes.push_back(nd);
    }
    A.pool.insert(A.pool.end(), pool_local.begin(), pool_local.end());
    // where a traceback may start (:2442-2556): every sink in topological order (PARITY-UNPINNED: the reference iterates an
    // unordered_set of matrix pointers), looking through empty sinks to their predecessors depth-first, last predecessor first
    {
        std::vector<int64_t> st; std::vector<uint32_t> path;
        for (uint32_t v = 0; v < N; ++v) {
            if (!succ[v].empty() || masked[v]) continue;
            st.assign(1, v); path.clear();
            while (!st.empty()) {
                const int64_t u = st.back(); st.pop_back();
                if (u < 0) { path.pop_back(); continue; }
                if (masked[u]) continue;
                if (len[u] == 0) {
                    path.push_back((uint32_t)u); st.push_back(-1);
                    if (is_source((uint32_t)u)) { if (!hp.have_empty_walk) { hp.have_empty_walk = true; hp.empty_walk = path; } continue; }
                    for (uint32_t e = g.pred_off[u]; e < g.pred_off[u + 1]; ++e) st.push_back(g.pred_idx[e]);
                    continue;
                }
                A.starts.push_back({(uint32_t)u}); hp.start_prefix.push_back(path);
            }
        }
    }
    pb.n_starts = (uint32_t)(A.starts.size() - pb.start_base);
    for (int64_t i = 0; i < L; ++i) A.reads.push_back(nt_code(p.read[i]));
    if (ctx->has_qa) A.quals.insert(A.quals.end(), p.qual, p.qual + L);
    for (uint64_t i = 0; i < total_bases; ++i) A.graph.push_back(nt_code(g.seq[i]));
    pb.ops_off = A.ops_total; pb.ops_cap = (uint32_t)(L + total_bases + 2ull * N + 8);
    A.ops_total += pb.ops_cap;
    A.tb_bytes += (tb_off + 255) & ~255ull; A.last_elems += last_off;
    A.probs.push_back(pb);
    hp.on_device = true;
    return VGK_OK;
}

template <class T> int to_dev(Backend* be, std::vector<void*>& held, const std::vector<T>& v, const T*& out) {
    void* d = be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!d) return VGK_ENOMEM;
    held.push_back(d);
    if (!v.empty()) { int rc = be->upload(d, v.data(), v.size() * sizeof(T)); if (rc) return rc; }
    out = (const T*)d;
    return VGK_OK;
}

// run one arena on the device and scatter the results
int run_arena(vgk_ctx* ctx, Arena& A, std::vector<HostProblem>& hps, const vgk_banded_problem* problems,
              std::vector<BResult>& results, std::vector<vgk_op>& ops) {
    Backend* be = ctx->be.get();
    std::vector<void*> held;
    auto cleanup = [&](int rc) { for (void* d : held) be->release(d); return rc; };
    const uint32_t n = (uint32_t)A.probs.size();
    if (!n) return VGK_OK;
    // launches: one per rows-per-lane class, long problems first inside a class
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t ra = hps[A.owner[a]].R, rb = hps[A.owner[b]].R;
        if (ra != rb) return ra < rb;
        return hps[A.owner[a]].cells > hps[A.owner[b]].cells;
    });
    std::vector<BandedLaunch> launches;
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i; const uint32_t R = hps[A.owner[order[i]]].R;
        while (j < n && hps[A.owner[order[j]]].R == R) ++j;
        launches.push_back({R, i, j - i});
        i = j;
    }
    BandedParams P{};
    int rc;
    if ((rc = to_dev(be, held, A.probs, P.probs))) return cleanup(rc);
    if ((rc = to_dev(be, held, order, P.order))) return cleanup(rc);
    if ((rc = to_dev(be, held, A.nodes, P.nodes))) return cleanup(rc);
    if ((rc = to_dev(be, held, A.seeds, P.seeds))) return cleanup(rc);
    if ((rc = to_dev(be, held, A.pool, P.pool))) return cleanup(rc);
    if ((rc = to_dev(be, held, A.starts, P.starts))) return cleanup(rc);
    if ((rc = to_dev(be, held, A.reads, P.reads))) return cleanup(rc);
    if (ctx->has_qa) { if ((rc = to_dev(be, held, A.quals, P.quals))) return cleanup(rc); }
    if ((rc = to_dev(be, held, A.graph, P.graph))) return cleanup(rc);
    std::vector<int8_t> mat = ctx->has_qa ? ctx->qmat : std::vector<int8_t>(ctx->sc.matrix, ctx->sc.matrix + 25);
    if ((rc = to_dev(be, held, mat, P.mat))) return cleanup(rc);
    P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend; P.n = n;
    auto dev_alloc = [&](uint64_t bytes) -> void* { void* d = be->alloc(bytes); if (d) held.push_back(d); return d; };
    P.tb = (uint8_t*)dev_alloc(std::max<uint64_t>(A.tb_bytes, 256));
    P.last = (int32_t*)dev_alloc(std::max<uint64_t>(A.last_elems, 64) * sizeof(int32_t));
    P.ops = (vgk_op*)dev_alloc(std::max<uint64_t>(A.ops_total, 1) * sizeof(vgk_op));
    P.results = (BResult*)dev_alloc((size_t)n * sizeof(BResult));
    if (!P.tb || !P.last || !P.ops || !P.results) return cleanup(VGK_ENOMEM);
    if ((rc = be->run_banded(P, launches.data(), (uint32_t)launches.size()))) return cleanup(rc);
    results.resize(n); ops.resize(std::max<uint64_t>(A.ops_total, 1));
    if ((rc = be->download(results.data(), P.results, (size_t)n * sizeof(BResult)))) return cleanup(rc);
    if ((rc = be->download(ops.data(), P.ops, (size_t)A.ops_total * sizeof(vgk_op)))) return cleanup(rc);
    ctx->banded_ms[0] += be->last_ms(3); ctx->banded_ms[1] += be->last_ms(4);
    (void)problems;
    return cleanup(VGK_OK);
}

}  // namespace

extern "C" {

int vgk_banded_align(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                     vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->banded_ms[0] = ctx->banded_ms[1] = 0; ctx->banded_cells = 0; ctx->banded_bytes = 0;
    uint64_t budget = ctx->be->memory_bytes() / 2;
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    if (!budget) budget = 1ull << 30;
    size_t used = 0; int rc_all = VGK_OK;
    std::vector<HostProblem> hps(n);
    uint32_t i = 0;
    while (i < n) {
        // cut a sub-batch that fits the budget
        Arena A; uint32_t j = i;
        for (; j < n; ++j) {
            const uint64_t before = A.tb_bytes + A.last_elems * 4 + A.ops_total * sizeof(vgk_op);
            if (j > i && before > budget) break;
            hps[j].status = prepare(ctx, problems[j], hps[j], A);
            if (hps[j].on_device) A.owner.push_back(j);
        }
        std::vector<BResult> dres; std::vector<vgk_op> dops;
        int rc = run_arena(ctx, A, hps, problems, dres, dops);
        if (rc) return rc;
        // results in the caller's order; the empty-walk rule and the empty sink prefixes are host bookkeeping (:2611-2668, :196-203)
        uint32_t a = 0;
        for (uint32_t q = i; q < j; ++q) {
            vgk_result& r = results[q];
            std::memset(&r, 0, sizeof r);
            r.ops_begin = (uint32_t)used;
            HostProblem& hp = hps[q];
            if (!hp.on_device) { r.status = hp.status; continue; }
            const BProb& pb = A.probs[a]; const BResult& dr = dres[a]; ++a;
            ctx->banded_cells += hp.cells;
            const vgk_banded_problem& p = problems[q];
            uint64_t bases = 0; for (uint32_t v = 0; v < p.graph.n_nodes; ++v) bases += p.graph.node_len[v];
            ctx->banded_bytes += p.read_len + bases + 8ull * p.graph.n_nodes + 4ull * p.graph.pred_off[p.graph.n_nodes] + hp.cells + 16 + 2ull * dr.n_ops;
            const int32_t empty_score = -ctx->sc.gap_open - (int32_t)(p.read_len - 1) * ctx->sc.gap_extend;
            const bool have = dr.status != VGK_ENOBAND;
            std::vector<vgk_op> out;
            if (hp.have_empty_walk && (!have || empty_score >= dr.score)) {
                r.score = empty_score; r.status = VGK_OK;
                for (size_t k = hp.empty_walk.size(); k-- > 0;) {
                    vgk_op o{}; o.node = hp.empty_walk[k];
                    if (k + 1 == hp.empty_walk.size()) { o.op = VGK_OP_I; o.len = (uint16_t)p.read_len; } else { o.op = VGK_OP_M; o.len = 0; }
                    out.push_back(o);
                }
            } else if (dr.status != VGK_OK) {
                r.status = dr.status;
            } else {
                r.score = dr.score; r.status = VGK_OK;
                const vgk_op* src = dops.data() + pb.ops_off + dr.ops_begin;
                for (uint32_t k = 0; k < dr.n_ops; ++k) { vgk_op o = src[k]; if (o.len == 0) o.op = VGK_OP_M; out.push_back(o); }
                const std::vector<uint32_t>& prefix = hp.start_prefix[dr.start];
                for (size_t k = prefix.size(); k-- > 0;) { vgk_op o{}; o.node = prefix[k]; o.op = VGK_OP_M; o.len = 0; out.push_back(o); }
            }
            if (r.status == VGK_OK) {
                if (used + out.size() > ops_cap || !ops) { r.status = VGK_EOPS; rc_all = VGK_EOPS; }
                else { std::copy(out.begin(), out.end(), ops + used); r.n_ops = (uint32_t)out.size(); used += out.size(); }
            }
        }
        i = j;
    }
    if (ops_written) *ops_written = used;
    return rc_all;
}

double vgk_banded_last(vgk_ctx* ctx, int which) {
    if (!ctx) return 0.0;
    switch (which) { case 0: return ctx->banded_ms[0]; case 1: return ctx->banded_ms[1];
                     case 2: return (double)ctx->banded_cells; case 3: return (double)ctx->banded_bytes; default: return 0.0; }
}

}  // extern "C"
