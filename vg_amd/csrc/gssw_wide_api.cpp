// gssw_wide_api.cpp — vgk_gssw_align's route for problems outside the packed kernels' range (gssw_wide_device.hpp): reads of more
// than 1024 rows and scorings whose reachable scores do not fit the packed kernels' 11 bits.  Same modes, same results; the
// reference's own 4.4 kbp tail (src/unittest/minimizer_mapper.cpp:682-709) takes this route.
//
// Packing follows vgk_gssw_pack (vgk_api.cpp) — the same column-info stream, node table and predecessor CSR, what
// GSSWAligner::create_gssw_graph builds per call (src/aligner.cpp:30-85) as flat arenas — serially: these problems are rare and
// large, the host's share is a few microseconds per thousand cells.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "backend.hpp"
#include "batch.hpp"
#include "ctx.hpp"
#include "gssw_wide.hpp"

using namespace vgk;

namespace {

struct Packed {
    std::vector<WideProb> probs;
    std::vector<uint8_t> colinfo; std::vector<uint32_t> prof; std::vector<NodeRec> nodes; std::vector<uint32_t> preds;
    uint64_t scratch = 0, tb = 0, carry = 0, ops = 0;
    void clear() { probs.clear(); colinfo.clear(); prof.clear(); nodes.clear(); preds.clear(); scratch = tb = carry = ops = 0; }
};

}  // namespace

int vgk::wide_problem_status(const vgk_ctx* ctx, const vgk_gssw_problem& p) {
    if (p.read_len == 0 || !p.read || p.graph.n_nodes == 0) return VGK_EINVAL;
    const uint32_t mode = p.flags & 15u;
    if (mode != VGK_GSSW_LOCAL && mode != VGK_GSSW_PINNED && mode != VGK_XDROP_PINNED) return VGK_EINVAL;
    const vgk_graph& g = p.graph;
    if ((ctx->has_qa && !p.qual) || (mode == VGK_GSSW_PINNED && !p.pinning) || !g.node_len || !g.pred_off || !g.seq) return VGK_EINVAL;
    if (p.read_len >= 65535u) return VGK_ETOOLONG;            // vgk_op.len is 16 bits: a whole-read insertion must fit
    uint64_t R = 0;
    for (uint32_t v = 0; v < g.n_nodes; ++v) {
        const uint32_t pb = g.pred_off[v], pe = g.pred_off[v + 1];
        if (pe < pb || g.node_len[v] == 0 || (pe > pb && !g.pred_idx)) return VGK_EINVAL;
        if (g.node_len[v] > 65535u) return VGK_ETOOBIG;
        for (uint32_t k = pb; k < pe; ++k) if (g.pred_idx[k] >= v) return VGK_EINVAL;
        R += g.node_len[v];
    }
    if (R >= (1u << 20)) return VGK_ETOOBIG;
    return VGK_OK;
}

// Appends problem p to the arenas.  The caller has checked it with wide_problem_status.
static void pack_one(const vgk_ctx* ctx, const vgk_gssw_problem& p, Packed& A) {
    const vgk_graph& g = p.graph;
    const uint32_t mode = p.flags & 15u; const bool xdrop = mode == VGK_XDROP_PINNED;
    WideProb d{};
    d.flags = p.flags; d.L = p.read_len + (xdrop ? 1u : 0u); d.n_nodes = g.n_nodes;
    d.max_gap = xdrop ? ((std::max<uint32_t>(p.max_gap_length, 1u) + 7u) & ~7u) : 0u;
    // which full-length bonuses this problem grants, and their values (src/aligner.cpp:401-402, 942-952, 1164-1167)
    const int first_b = ctx->has_qa ? ctx->qbon[p.qual[0]] : ctx->sc.full_length_bonus;
    const int last_b = ctx->has_qa ? ctx->qbon[p.qual[p.read_len - 1]] : ctx->sc.full_length_bonus;
    d.bonus_start = xdrop ? 0 : first_b;
    d.bonus_end = (mode == VGK_GSSW_PINNED) ? 0 : last_b;
    // rows per lane: 8 while one strip of 256 lanes holds the read, else 16
    d.K = d.L <= WIDE_LANES * 8u ? 8u : 16u;
    d.n_strips = (d.L + WIDE_LANES * d.K - 1) / (WIDE_LANES * d.K);
    d.Lpad = d.n_strips * WIDE_LANES * d.K;
    // per-row profile words: byte b = score against reference base b + bias, both bonuses folded in; X-drop row 0 consumes nothing
    d.prof_off = (uint32_t)A.prof.size();
    if (xdrop) A.prof.push_back(0u);
    for (uint32_t r = 0; r < p.read_len; ++r) {
        const int code = nt_read(p.read[r]);
        uint32_t w = 0;
        for (int b4 = 0; b4 < 4; ++b4) {
            const int s = ctx->has_qa ? ctx->qmat[25 * p.qual[r] + 5 * b4 + code] : ctx->sc.matrix[5 * b4 + code];
            w |= (uint32_t)(s + (int)ctx->bias) << (8 * b4);
        }
        const uint32_t row = r + (xdrop ? 1u : 0u);
        w += 0x01010101u * row_bonus((uint32_t)d.bonus_start, (uint32_t)d.bonus_end, row, d.L);
        A.prof.push_back(w);
    }
    // nodes whose last column is saved (a successor seeds from it / the pinned end) and nodes seeded from scratch
    std::vector<uint8_t> store(g.n_nodes, 0), slow(g.n_nodes, 0);
    for (uint32_t v = 0; v < g.n_nodes; ++v) {
        const uint32_t pb = g.pred_off[v], pe = g.pred_off[v + 1];
        const bool chain = (pe - pb == 1) && g.pred_idx[pb] + 1 == v;
        slow[v] = ((v > 0 || xdrop) && !chain) ? 1 : 0;
        if (slow[v]) for (uint32_t k = pb; k < pe; ++k) store[g.pred_idx[k]] = 1;
        if (mode == VGK_GSSW_PINNED && p.pinning[v]) store[v] = 1;
    }
    d.col_off = (uint32_t)A.colinfo.size(); d.node_off = (uint32_t)A.nodes.size();
    uint32_t col = 0, slots = 0, seq_pos = 0;
    for (uint32_t v = 0; v < g.n_nodes; ++v) {
        NodeRec nr;
        nr.col_start = col; nr.col_end = col + g.node_len[v];
        nr.pred_begin = (uint32_t)A.preds.size(); nr.n_pred = g.pred_off[v + 1] - g.pred_off[v];
        for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) A.preds.push_back(g.pred_idx[k]);
        nr.slot = store[v] ? (int32_t)slots++ : -1;
        nr.pinning = (mode == VGK_GSSW_PINNED && p.pinning[v]) ? 1u : 0u;
        A.nodes.push_back(nr);
        for (uint32_t k = 0; k < g.node_len[v]; ++k, ++seq_pos) {
            uint8_t ci = (uint8_t)nt_ref(g.seq[seq_pos]);
            if (k == 0) { ci |= CI_NODE_START; if (slow[v]) ci |= CI_SEED_SLOW; }
            if (k + 1 == g.node_len[v] && store[v]) ci |= CI_STORE_END;
            A.colinfo.push_back(ci);
        }
        col = nr.col_end;
    }
    d.R = col; d.n_slots = slots;
    d.scratch_off = A.scratch; A.scratch += (uint64_t)slots * d.Lpad;
    d.carry_off = A.carry; A.carry += d.n_strips > 1 ? d.R : 0;
    d.strip_dwords = (uint64_t)(d.R + WIDE_LANES - 1) * WIDE_LANES * (d.K / 8);
    d.tb_off = A.tb;
    if (p.flags & VGK_GSSW_TRACEBACK) A.tb += d.strip_dwords * d.n_strips;
    d.ops_off = (uint32_t)A.ops; d.ops_cap = (p.flags & VGK_GSSW_TRACEBACK) ? p.read_len + d.R + 2 : 0;
    A.ops += d.ops_cap;
    A.probs.push_back(d);
}

int vgk::wide_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, const uint32_t* idx, uint32_t m,
                    vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_at) {
    if (!m) return VGK_OK;
    Backend* be = ctx->be.get();
    ctx->wide_ms[0] = ctx->wide_ms[1] = 0; ctx->wide_cells = ctx->wide_tb_cells = 0; ctx->wide_launches = 0;
    uint64_t budget = be->memory_bytes() ? be->memory_bytes() / 4 : (2ull << 30);
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    std::lock_guard<std::mutex> lock(ctx->mu);
    Packed A;
    // an upper bound of what a problem takes in HBM (every node saved; codes for every cell), to cut the call into sub-batches
    auto estimate = [&](const vgk_gssw_problem& p) -> uint64_t {
        uint64_t R = 0; for (uint32_t v = 0; v < p.graph.n_nodes; ++v) R += p.graph.node_len[v];
        const uint64_t L = p.read_len + 1ull, K = L <= WIDE_LANES * 8u ? 8 : 16, strips = (L + WIDE_LANES * K - 1) / (WIDE_LANES * K), Lpad = strips * WIDE_LANES * K;
        return sizeof(WPair) * (p.graph.n_nodes * Lpad + R) + 4 * (R + WIDE_LANES) * WIDE_LANES * (K / 8) * strips + 16 * (L + R) + 64ull * p.graph.n_nodes + 1024;
    };
    uint32_t begin = 0;
    while (begin < m) {
        A.clear();
        uint32_t end = begin; uint64_t bytes = 0;
        std::vector<uint32_t> owner;                       // runnable problems of this sub-batch (positions in idx)
        for (; end < m; ++end) {
            const vgk_gssw_problem& p = problems[idx[end]];
            const int st = wide_problem_status(ctx, p);
            if (st != VGK_OK) { std::memset(&results[idx[end]], 0, sizeof(vgk_result)); results[idx[end]].status = st; continue; }
            const uint64_t pb = estimate(p);
            if (!owner.empty() && bytes + pb > budget) break;
            bytes += pb; owner.push_back(end);
        }
        for (uint32_t k : owner) pack_one(ctx, problems[idx[k]], A);
        const uint32_t n = (uint32_t)owner.size();
        if (n) {
            if (A.colinfo.size() >= (1ull << 32) || A.prof.size() >= (1ull << 32) || A.ops >= (1ull << 32)) return VGK_ETOOBIG;
            WideParams P{};
            auto dev = [&](int slot, const void* src, uint64_t bytes) -> void* {
                void* d = ctx->ensure_scratch(slot, std::max<uint64_t>(bytes, 16)); if (!d) return nullptr;
                if (src && bytes && be->upload(d, src, bytes)) return nullptr;
                return d;
            };
            std::vector<uint32_t> order(n);
            uint32_t n8 = 0;
            for (uint32_t k = 0; k < n; ++k) if (A.probs[k].K == 8) order[n8++] = k;
            { uint32_t at = n8; for (uint32_t k = 0; k < n; ++k) if (A.probs[k].K != 8) order[at++] = k; }
            A.colinfo.resize(A.colinfo.size() + 8, (uint8_t)CI_INVALID);
            P.probs = (const WideProb*)dev(88, A.probs.data(), sizeof(WideProb) * n); P.n = n;
            P.order = (const uint32_t*)dev(89, order.data(), 4ull * n);
            P.colinfo = (const uint8_t*)dev(90, A.colinfo.data(), A.colinfo.size());
            P.prof = (const uint32_t*)dev(91, A.prof.data(), 4ull * A.prof.size());
            P.nodes = (const NodeRec*)dev(92, A.nodes.data(), sizeof(NodeRec) * A.nodes.size());
            P.preds = (const uint32_t*)dev(93, A.preds.data(), 4ull * A.preds.size());
            P.scratch = (WPair*)dev(94, nullptr, sizeof(WPair) * (A.scratch + 1));
            P.carry = (WPair*)dev(95, nullptr, sizeof(WPair) * (A.carry + 1));
            P.tb = (uint32_t*)dev(96, nullptr, 4ull * (A.tb + 1));
            P.best = (unsigned long long*)dev(97, nullptr, 8ull * (n + 1));
            P.results = (vgk_result*)dev(98, nullptr, sizeof(vgk_result) * (n + 1ull));
            P.ops = (vgk_op*)dev(99, nullptr, sizeof(vgk_op) * (A.ops + 1));
            if (!P.probs || !P.order || !P.colinfo || !P.prof || !P.nodes || !P.preds || !P.scratch || !P.carry || !P.tb || !P.best || !P.results || !P.ops) return VGK_ENOMEM;
            P.bias = (int32_t)ctx->bias; P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend;
            int rc = be->zero(P.best, 8ull * (n + 1));
            if (!rc) rc = be->run_gssw_wide(P, n8, n - n8);
            if (rc) return rc;
            { uint64_t cells = 0, tb = 0; for (uint32_t k = 0; k < n; ++k) { const WideProb& d = A.probs[k]; cells += (uint64_t)d.L * d.R; if (d.flags & VGK_GSSW_TRACEBACK) tb += (uint64_t)d.L * d.R; }
              ctx->wide_cells += cells; ctx->wide_tb_cells += tb; ctx->wide_launches += 1; }
            std::vector<vgk_result> res(n); std::vector<vgk_op> all(A.ops + 1);
            if ((rc = be->download(res.data(), P.results, sizeof(vgk_result) * n))) return rc;
            if (A.ops && (rc = be->download(all.data(), P.ops, sizeof(vgk_op) * A.ops))) return rc;
            for (uint32_t k = 0; k < n; ++k) {
                vgk_result r = res[k];
                const uint32_t src = r.ops_begin;
                if (r.status == VGK_OK && r.n_ops) {
                    if (!ops || *ops_at + r.n_ops > ops_cap) { r.status = VGK_EOPS; r.n_ops = 0; }
                    else { std::memcpy(ops + *ops_at, all.data() + src, sizeof(vgk_op) * r.n_ops); }
                } else r.n_ops = 0;
                r.ops_begin = (uint32_t)*ops_at; *ops_at += r.n_ops;
                results[idx[owner[k]]] = r;
            }
            ctx->wide_ms[0] += be->last_ms(13); ctx->wide_ms[1] += be->last_ms(14);
        }
        begin = end;
    }
    return VGK_OK;
}

extern "C" double vgk_gssw_wide_last(vgk_ctx* ctx, int which) {
    if (!ctx) return 0.0;
    switch (which) { case 0: return ctx->wide_ms[0]; case 1: return ctx->wide_ms[1]; case 2: return (double)ctx->wide_cells; case 3: return (double)ctx->wide_tb_cells; case 4: return (double)ctx->wide_launches; default: return 0.0; }
}
