// gssw_multi_api.cpp — vgk_gssw_align_multi: k-best pinned alignments (Aligner::align_pinned_multi, reference
// src/aligner.cpp:423-435, :455-480).
//
// Device: the pinned fill with every cell's H / E / F kept (gssw_matrix_device.hpp), in sub-batches that fit the memory
// budget, then the alternates enumerated over them, one lane per problem (gssw_multi_device.hpp).  Host: a thread per problem
// the kernel declines (or all of them when max_alt_alns > 62) walks the same rules over that problem's downloaded matrices.
//
// The enumeration (gssw's own is not in the reference snapshot — DESIGN.md §13; the oracle states the same rules): a traceback
// is the walk of the single traceback's state machine over H / E / F.  The sources of a state come in a fixed order — H:
// the diagonal through each predecessor column (list order), then E, then F; E: per predecessor column gap-open, gap-extend;
// F: gap-open, gap-extend — each with a loss = the state's value minus the value through that source.  The default walk takes
// the first source without loss; an alternate takes named other sources at some states (its deflections) and scores the start
// value minus the losses.  While the walk of an alternate runs past its last deflection, every other source worth more than 0
// that it passes becomes a new alternate; alternates are walked best first (earlier proposals first among equals) until
// max_alt_alns are out.  A gap is never opened directly after a gap of the same kind was opened (it would print as one longer
// gap, i.e. as another alternate), diagonal sources through predecessor cells worth 0 count as one (the alignment starts at the
// current cell either way), and a walk ends only where the DP value is 0 or the read is used up.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {

inline uint8_t code_read(char ch) {     // gssw_create_nt_table: case-insensitive ACGT, else N
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline uint8_t code_ref(char ch) {      // after nonATGCNtoN (src/aligner.cpp:39): upper-case ACGT only
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

enum { AT_H = 0, AT_E = 1, AT_F = 2 };
struct Source { int32_t value; int st, r, c; };                  // st < 0: the walk ends after this step
struct Deflection { int st, r, c; uint32_t take; };
struct Alternate { int32_t score; uint32_t start; std::vector<Deflection> deflections; };

class Tracer {
public:
    Tracer(const vgk_ctx* ctx, const vgk_gssw_problem& p, const MProb& pb, const int32_t* cells)
        : p(p), L((int)pb.L), R((int)pb.R), go(ctx->sc.gap_open), ge(ctx->sc.gap_extend), H(cells), E(cells + (size_t)pb.R * pb.L), F(cells + 2 * (size_t)pb.R * pb.L) {
        const vgk_graph& g = p.graph;
        col0.resize(g.n_nodes + 1, 0);
        for (uint32_t v = 0; v < g.n_nodes; ++v) col0[v + 1] = col0[v] + (int)g.node_len[v];
        node_of.resize((size_t)R);
        for (uint32_t v = 0; v < g.n_nodes; ++v) for (int c = col0[v]; c < col0[v + 1]; ++c) node_of[(size_t)c] = (int)v;
        mat = ctx->has_qa ? ctx->qmat.data() : ctx->sc.matrix;
        start_bonus = pb.start_bonus; qa = ctx->has_qa;
    }

    void run(uint32_t max_alt_alns, std::vector<vgk_result>& results, std::vector<vgk_op>& ops) {
        std::vector<int> starts;
        for (uint32_t v = 0; v < p.graph.n_nodes; ++v) if (p.pinning[v]) starts.push_back(col0[v + 1] - 1);
        std::vector<Alternate> queue;                            // best first; stable among equals
        for (uint32_t s = 0; s < starts.size(); ++s) {
            const int32_t v = h(starts[s], L - 1);
            if (v > 0) offer(queue, Alternate{v, s, {}}, max_alt_alns);
        }
        std::vector<vgk_op> one;
        while (!queue.empty() && results.size() < max_alt_alns) {
            const Alternate alt = std::move(queue.front()); queue.erase(queue.begin());
            vgk_result res{};
            walk(alt, starts, queue, max_alt_alns - (uint32_t)results.size() - 1, res, one);
            res.ops_begin = (uint32_t)ops.size();
            ops.insert(ops.end(), one.begin(), one.end());
            results.push_back(res);
        }
    }

private:
    const vgk_gssw_problem& p; int L, R, go, ge; const int32_t *H, *E, *F;
    std::vector<int> col0, node_of; const int8_t* mat; int32_t start_bonus; bool qa;

    int32_t h(int c, int r) const { return H[(size_t)c * L + r]; }
    int32_t e(int c, int r) const { return E[(size_t)c * L + r]; }
    int32_t f(int c, int r) const { return F[(size_t)c * L + r]; }
    int32_t score(int r, int c) const {
        const int ref = code_ref(p.graph.seq[c]), rd = code_read(p.read[r]);
        return (qa ? mat[25 * p.qual[r] + 5 * ref + rd] : mat[5 * ref + rd]) + (r == 0 ? start_bonus : 0);
    }
    void pred_cols(int c, std::vector<int>& out) const {
        out.clear();
        const int v = node_of[(size_t)c];
        if (c != col0[v]) { out.push_back(c - 1); return; }
        for (uint32_t k = p.graph.pred_off[v]; k < p.graph.pred_off[v + 1]; ++k) out.push_back(col0[p.graph.pred_idx[k] + 1] - 1);
    }
    // the sources of a state in their fixed order; n_diag = how many of them are diagonal steps (H states)
    void sources(int st, int r, int c, bool no_e, bool no_f, std::vector<Source>& out, uint32_t& n_diag) const {
        std::vector<int> pc; pred_cols(c, pc);
        out.clear(); n_diag = 0;
        if (st == AT_H) {
            const int32_t s = score(r, c);
            if (r == 0 || pc.empty()) out.push_back({s, -1, r - 1, c});
            else {
                bool zero_seen = false;                          // predecessors whose cell is worth 0 all mean "the alignment starts here": one source
                for (int q : pc) {
                    const int32_t d = h(q, r - 1);
                    if (d == 0) { if (zero_seen) continue; zero_seen = true; }
                    out.push_back({d + s, AT_H, r - 1, q});
                }
            }
            n_diag = (uint32_t)out.size();
            out.push_back({no_e ? 0 : e(c, r), AT_E, r, c});
            out.push_back({no_f ? 0 : f(c, r), AT_F, r, c});
        } else if (st == AT_E) {
            for (int q : pc) { out.push_back({h(q, r) - go, AT_H, r, q}); out.push_back({e(q, r) - ge, AT_E, r, q}); }
        } else if (r > 0) {
            out.push_back({h(c, r - 1) - go, AT_H, r - 1, c});
            out.push_back({f(c, r - 1) - ge, AT_F, r - 1, c});
        }
    }
    bool explains(int r, int c, bool no_e, bool no_f) const {     // can H(r, c) be left without entering E (F)?
        const int32_t value = h(c, r);
        if (value == 0) return true;
        std::vector<Source> src; uint32_t nd; sources(AT_H, r, c, no_e, no_f, src, nd);
        for (const Source& s : src) if (s.value == value) return true;
        return false;
    }
    static void offer(std::vector<Alternate>& queue, Alternate a, uint32_t room) {
        if (!room) return;
        size_t at = queue.size();
        while (at > 0 && queue[at - 1].score < a.score) --at;
        if (at >= room) return;
        queue.insert(queue.begin() + (long)at, std::move(a));
        if (queue.size() > room) queue.resize(room);
    }
    static void push(std::vector<vgk_op>& o, uint32_t node, int op, uint32_t len) {
        if (!o.empty() && o.back().node == node && o.back().op == op) { o.back().len = (uint16_t)(o.back().len + len); return; }
        vgk_op x{}; x.node = node; x.op = (uint8_t)op; x.len = (uint16_t)len; o.push_back(x);
    }

    void walk(const Alternate& alt, const std::vector<int>& starts, std::vector<Alternate>& queue, uint32_t room, vgk_result& res, std::vector<vgk_op>& o) const {
        int st = AT_H, r = L - 1, c = starts[alt.start], first_c = c;
        bool no_e = false, no_f = false;
        size_t next = 0;
        int32_t lost = 0;
        const int32_t start_value = h(c, r);
        res.score = alt.score; res.status = VGK_OK; res.end_node = node_of[(size_t)c]; res.end_offset = c - col0[(size_t)node_of[(size_t)c]]; res.end_read = r;
        o.clear();
        std::vector<Source> src; uint32_t n_diag = 0;
        for (;;) {
            const int32_t value = st == AT_H ? h(c, r) : st == AT_E ? e(c, r) : f(c, r);
            if (st == AT_H && value == 0) break;
            sources(st, r, c, no_e, no_f, src, n_diag);
            size_t take = src.size();
            const bool here = next < alt.deflections.size() && alt.deflections[next].st == st && alt.deflections[next].r == r && alt.deflections[next].c == c;
            if (here) take = alt.deflections[next++].take;
            else {
                for (size_t k = 0; k < src.size(); ++k) if (src[k].value == value) { take = k; break; }
                if (take >= src.size() && (no_e || no_f)) {      // gap_open == gap_extend: like the single traceback, re-open the gap
                    no_e = no_f = false;
                    sources(st, r, c, false, false, src, n_diag);
                    for (size_t k = 0; k < src.size(); ++k) if (src[k].value == value) { take = k; break; }
                }
            }
            if (take >= src.size()) break;
            if (next == alt.deflections.size() && !here) {
                for (size_t k = 0; k < src.size(); ++k) {
                    if (k == take || src[k].value <= 0) continue;
                    if (st != AT_H && src[k].st == AT_H && !explains(src[k].r, src[k].c, st == AT_E, st == AT_F)) continue;
                    const int32_t score2 = start_value - lost - (value - src[k].value);
                    if (score2 <= 0) continue;
                    Alternate a{score2, alt.start, alt.deflections};
                    a.deflections.push_back({st, r, c, (uint32_t)k});
                    offer(queue, std::move(a), room);
                }
            }
            lost += value - src[take].value;
            const uint32_t node = (uint32_t)node_of[(size_t)c];
            if (st == AT_H && take < n_diag) {
                push(o, node, VGK_OP_M, 1); first_c = c; no_e = no_f = false;
                if (src[take].st < 0) { r -= 1; break; }
                r = src[take].r; c = src[take].c;
            } else if (st == AT_H) st = src[take].st;
            else if (st == AT_E) { push(o, node, VGK_OP_D, 1); first_c = c; no_e = src[take].st == AT_H; no_f = false; st = src[take].st; c = src[take].c; }
            else { push(o, node, VGK_OP_I, 1); no_f = src[take].st == AT_H; no_e = false; st = src[take].st; r = src[take].r; }
        }
        if (r >= 0) push(o, (uint32_t)node_of[(size_t)first_c], VGK_OP_S, (uint32_t)r + 1);
        std::reverse(o.begin(), o.end());
        res.n_ops = (uint32_t)o.size(); res.first_offset = first_c - col0[(size_t)node_of[(size_t)first_c]];
    }
};

struct MultiHost { RawBuf<uint8_t> reads, quals, graph; RawBuf<MProb> probs; RawBuf<MNode> nodes; RawBuf<uint32_t> preds; RawBuf<int32_t> cells; };

}  // namespace

extern "C" {

uint64_t vgk_gssw_multi_host_walks(const vgk_ctx* ctx) { return ctx ? ctx->multi_host_walks : 0; }

int vgk_gssw_align_multi(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n, uint32_t max_alt_alns,
                         vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written) try {
    if (!ctx || (!problems && n) || (!results && n) || (!n_alignments && n) || !max_alt_alns) return VGK_EINVAL;
    if (ops_written) *ops_written = 0;
    if (!n) return VGK_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    Backend* be = ctx->be.get();
    uint64_t budget = be->memory_bytes() / 2;
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    if (!budget) budget = 1ull << 30;
    if (!ctx->multi_host) ctx->multi_host = std::make_shared<MultiHost>();
    MultiHost& Hs = *static_cast<MultiHost*>(ctx->multi_host.get());
    const bool qa = ctx->has_qa;
    const bool on_device = max_alt_alns + 2 <= 64 && !std::getenv("VGAMD_MULTI_HOST_WALK");     // a lane's slot pool is a 64-bit mask
    ctx->multi_host_walks = 0;

    // validation, per problem: failures are answered in the problem's first result
    std::vector<int> status(n, VGK_OK); std::vector<uint32_t> cols(n, 0);
    parallel_for(n, [&](uint32_t i, unsigned) {
        const vgk_gssw_problem& p = problems[i]; const vgk_graph& g = p.graph;
        int st = VGK_OK; uint64_t R = 0;
        if ((p.flags & 15u) != VGK_GSSW_PINNED || !p.pinning || !p.read || !p.read_len || !g.n_nodes || !g.node_len || !g.seq || !g.pred_off || (qa && !p.qual)) st = VGK_EINVAL;
        else if (p.read_len > 1024) st = VGK_ETOOLONG;
        else {
            bool pinned_any = false;
            for (uint32_t v = 0; v < g.n_nodes && st == VGK_OK; ++v) {
                if (!g.node_len[v] || g.pred_off[v + 1] < g.pred_off[v]) st = VGK_EINVAL;
                for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1] && st == VGK_OK; ++k) if (g.pred_idx[k] >= v) st = VGK_EINVAL;
                R += g.node_len[v]; pinned_any |= p.pinning[v] != 0;
            }
            if (st == VGK_OK && !pinned_any) st = VGK_EINVAL;
            if (st == VGK_OK && R * p.read_len > (1ull << 28)) st = VGK_ETOOBIG;
        }
        status[i] = st; cols[i] = (uint32_t)R;
    });

    size_t used = 0; int rc_all = VGK_OK;
    for (uint32_t i = 0; i < n;) {
        // a sub-batch whose matrices fit the budget
        uint64_t n_cells = 0, n_read = 0, n_graph = 0, n_nodes = 0, n_preds = 0, n_window = 0;
        uint32_t j = i; std::vector<uint32_t> owner;
        for (; j < n; ++j) {
            if (status[j] != VGK_OK) continue;
            const vgk_gssw_problem& p = problems[j];
            const uint64_t c3 = 3ull * cols[j] * p.read_len;
            const uint64_t w2 = on_device ? 2ull * max_alt_alns * ((uint64_t)cols[j] + p.read_len + 2) : 0;      // the walk's op window, in int32 units
            if (!owner.empty() && ((n_cells + c3) * sizeof(int32_t) + n_window * 4 + w2 * 4 > budget || (n_window + w2) / 2 >= (1ull << 31))) break;
            n_window += w2;
            n_cells += c3; n_read += p.read_len; n_graph += cols[j]; n_nodes += p.graph.n_nodes; n_preds += p.graph.pred_off[p.graph.n_nodes] - p.graph.pred_off[0];
            owner.push_back(j);
        }
        const uint32_t m = (uint32_t)owner.size();
        std::vector<std::vector<vgk_result>> pres(m); std::vector<std::vector<vgk_op>> pops(m);
        MProb* probs = Hs.probs.get(m + 1);
        if (m) {
            uint8_t* reads = Hs.reads.get(n_read + 1); uint8_t* quals = qa ? Hs.quals.get(n_read + 1) : nullptr; uint8_t* graph = Hs.graph.get(n_graph + 1);
            MNode* nodes = Hs.nodes.get(n_nodes + 1); uint32_t* preds = Hs.preds.get(n_preds + 1);
            { uint64_t a_cells = 0, a_read = 0, a_graph = 0, a_nodes = 0, a_preds = 0;
              for (uint32_t a = 0; a < m; ++a) {
                  const vgk_gssw_problem& p = problems[owner[a]];
                  MProb pb{}; pb.L = p.read_len; pb.n_nodes = p.graph.n_nodes; pb.R = cols[owner[a]];
                  pb.read_off = (uint32_t)a_read; pb.graph_off = (uint32_t)a_graph; pb.node_off = (uint32_t)a_nodes; pb.mat_off = a_cells;
                  pb.start_bonus = qa ? ctx->qbon[p.qual[0]] : ctx->sc.full_length_bonus; pb.status = VGK_OK;
                  probs[a] = pb;
                  // the predecessor lists go to one arena; pred_begin is an offset into it
                  uint32_t col = 0;
                  for (uint32_t v = 0; v < p.graph.n_nodes; ++v) {
                      MNode nd{}; nd.col_start = col; nd.col_end = col + p.graph.node_len[v]; nd.pred_begin = (uint32_t)a_preds; nd.n_pred = p.graph.pred_off[v + 1] - p.graph.pred_off[v];
                      for (uint32_t k = p.graph.pred_off[v]; k < p.graph.pred_off[v + 1]; ++k) preds[a_preds++] = p.graph.pred_idx[k];
                      nodes[a_nodes + v] = nd; col = nd.col_end;
                  }
                  a_cells += 3ull * pb.R * pb.L; a_read += pb.L; a_graph += pb.R; a_nodes += pb.n_nodes;
              } }
            parallel_for(m, [&](uint32_t a, unsigned) {
                const vgk_gssw_problem& p = problems[owner[a]]; const MProb& pb = probs[a];
                for (uint32_t r = 0; r < pb.L; ++r) reads[pb.read_off + r] = code_read(p.read[r]);
                if (qa) std::memcpy(quals + pb.read_off, p.qual, pb.L);
                for (uint32_t c = 0; c < pb.R; ++c) graph[pb.graph_off + c] = code_ref(p.graph.seq[c]);
            });
            GsswMatrixParams P{};
            P.n = m; P.go = ctx->sc.gap_open; P.ge = ctx->sc.gap_extend;
            auto dev = [&](int slot, const void* src, size_t bytes) -> void* {
                void* d = ctx->ensure_scratch(slot, std::max<size_t>(bytes, 16)); if (!d) return nullptr;
                if (src && bytes && be->upload(d, src, bytes)) return nullptr;
                return d;
            };
            P.probs = (MProb*)dev(40, probs, sizeof(MProb) * m);
            P.reads = (const uint8_t*)dev(41, reads, n_read); P.quals = qa ? (const uint8_t*)dev(42, quals, n_read) : nullptr;
            P.graph = (const uint8_t*)dev(43, graph, n_graph); P.nodes = (const MNode*)dev(44, nodes, sizeof(MNode) * n_nodes);
            P.preds = (const uint32_t*)dev(45, preds, sizeof(uint32_t) * n_preds);
            P.mat = (const int8_t*)dev(46, qa ? ctx->qmat.data() : ctx->sc.matrix, qa ? 6400 : 25);
            P.cells = (int32_t*)dev(47, nullptr, sizeof(int32_t) * n_cells);
            if (!P.probs || !P.reads || (qa && !P.quals) || !P.graph || !P.nodes || !P.preds || !P.mat || !P.cells) return VGK_ENOMEM;
            int rc;
            if ((rc = be->run_gssw_matrix(P))) return rc;
            if ((rc = be->download(probs, P.probs, sizeof(MProb) * m))) return rc;
            // The alternates.  On the device (gssw_multi_device.hpp: one lane per problem walks its matrices where they lie; only the
            // alignments come back) when the queue fits a lane's slot pool; a problem the kernel declines (VGK_ETOOBIG: a node with
            // more predecessors, or an alternate with more deflections, than a slot holds) is walked by a host thread over its own
            // matrices, which then are the only ones copied back.
            std::vector<uint8_t> on_host(m, 1);
            if (on_device) {
                std::vector<uint8_t> pin(n_nodes + 1); std::vector<uint64_t> ops_off(m + 1, 0);
                { uint64_t a_nodes = 0;
                  for (uint32_t a = 0; a < m; ++a) {
                      const vgk_gssw_problem& p = problems[owner[a]];
                      for (uint32_t v = 0; v < p.graph.n_nodes; ++v) pin[a_nodes + v] = p.pinning[v] ? 1 : 0;
                      a_nodes += p.graph.n_nodes;
                      ops_off[a + 1] = ops_off[a] + (uint64_t)max_alt_alns * ((uint64_t)probs[a].L + probs[a].R + 2);
                  } }
                const uint64_t n_res = (uint64_t)m * max_alt_alns, slots = max_alt_alns + 2;
                GsswMultiParams Q{};
                Q.M = P; Q.max_alt = max_alt_alns;
                Q.pinning = (const uint8_t*)dev(72, pin.data(), n_nodes);
                Q.pool = (MtAlt*)dev(73, nullptr, sizeof(MtAlt) * slots * m);
                Q.order = (uint32_t*)dev(74, nullptr, sizeof(uint32_t) * slots * m);
                Q.results = (vgk_result*)dev(75, nullptr, sizeof(vgk_result) * n_res);
                Q.n_alignments = (uint32_t*)dev(76, nullptr, sizeof(uint32_t) * m);
                Q.status = (int32_t*)dev(77, nullptr, sizeof(int32_t) * m);
                Q.ops = (vgk_op*)dev(78, nullptr, sizeof(vgk_op) * ops_off[m]);
                Q.ops_off = (const uint64_t*)dev(79, ops_off.data(), sizeof(uint64_t) * m);
                if (!Q.pinning || !Q.pool || !Q.order || !Q.results || !Q.n_alignments || !Q.status || !Q.ops || !Q.ops_off) return VGK_ENOMEM;
                if ((rc = be->zero(Q.results, sizeof(vgk_result) * n_res))) return rc;
                if ((rc = be->run_gssw_multi(Q))) return rc;
                std::vector<int32_t> dstat(m); std::vector<uint32_t> dcnt(m);
                if ((rc = be->download(dstat.data(), Q.status, sizeof(int32_t) * m))) return rc;
                if ((rc = be->download(dcnt.data(), Q.n_alignments, sizeof(uint32_t) * m))) return rc;
                // the ops packed behind each other on the device (the windows are mostly air), then results + ops back
                std::vector<vgk_result> dres(n_res); std::vector<vgk_op> dops;
                const uint32_t blocks = (uint32_t)((n_res + Backend::OPS_SCAN_BLOCK - 1) / Backend::OPS_SCAN_BLOCK);
                uint32_t* offs = (uint32_t*)dev(80, nullptr, sizeof(uint32_t) * n_res);
                uint32_t* sums = (uint32_t*)dev(81, nullptr, sizeof(uint32_t) * (blocks + 8));
                uint64_t total = 0;
                if (!offs || !sums) return VGK_ENOMEM;
                rc = be->ops_offsets(Q.results, (uint32_t)n_res, offs, sums, &total);
                if (rc == VGK_OK) {
                    vgk_result* pres_d = (vgk_result*)dev(82, nullptr, sizeof(vgk_result) * n_res);
                    vgk_op* pops_d = (vgk_op*)dev(83, nullptr, sizeof(vgk_op) * std::max<uint64_t>(total, 1));
                    if (!pres_d || !pops_d) return VGK_ENOMEM;
                    if ((rc = be->ops_gather(Q.results, Q.ops, (uint32_t)n_res, offs, sums, pres_d, pops_d))) return rc;
                    if ((rc = be->sync_fetch())) return rc;
                    dops.resize(total);
                    if ((rc = be->download(dres.data(), pres_d, sizeof(vgk_result) * n_res))) return rc;
                    if (total && (rc = be->download(dops.data(), pops_d, sizeof(vgk_op) * total))) return rc;
                } else if (rc == VGK_EUNSUPPORTED) {               // (the emulator: no packing kernels — everything comes back as it lies)
                    dops.resize(ops_off[m]);
                    if ((rc = be->download(dres.data(), Q.results, sizeof(vgk_result) * n_res))) return rc;
                    if (ops_off[m] && (rc = be->download(dops.data(), Q.ops, sizeof(vgk_op) * ops_off[m]))) return rc;
                } else return rc;
                for (uint32_t a = 0; a < m; ++a) {
                    if (probs[a].status != VGK_OK || dstat[a] != VGK_OK) continue;
                    on_host[a] = 0;
                    pres[a].assign(dres.begin() + (size_t)a * max_alt_alns, dres.begin() + (size_t)a * max_alt_alns + dcnt[a]);
                    for (vgk_result& r : pres[a]) {
                        const uint32_t at = (uint32_t)pops[a].size();
                        pops[a].insert(pops[a].end(), dops.begin() + r.ops_begin, dops.begin() + r.ops_begin + r.n_ops);
                        r.ops_begin = at;
                    }
                }
            }
            // ... and the ones left to host threads, each over its own matrices
            { std::vector<uint32_t> todo;
              for (uint32_t a = 0; a < m; ++a) if (on_host[a] && probs[a].status == VGK_OK) todo.push_back(a);
              ctx->multi_host_walks += todo.size();
              if (!todo.empty()) {
                  uint64_t need = 0; std::vector<uint64_t> at(todo.size());
                  for (size_t k = 0; k < todo.size(); ++k) { at[k] = need; need += 3ull * probs[todo[k]].R * probs[todo[k]].L; }
                  int32_t* cells = Hs.cells.get(need + 1);
                  for (size_t k = 0; k < todo.size(); ++k)
                      if ((rc = be->download(cells + at[k], P.cells + probs[todo[k]].mat_off, sizeof(int32_t) * 3ull * probs[todo[k]].R * probs[todo[k]].L))) return rc;
                  parallel_for((uint32_t)todo.size(), [&](uint32_t k, unsigned) {
                      const uint32_t a = todo[k];
                      Tracer t(ctx, problems[owner[a]], probs[a], cells + at[k]);
                      t.run(max_alt_alns, pres[a], pops[a]);
                  });
              } }
        }
        // results in the caller's order
        uint32_t a = 0;
        for (uint32_t q = i; q < j; ++q) {
            vgk_result* r = results + (size_t)q * max_alt_alns;
            std::memset(r, 0, sizeof(vgk_result) * max_alt_alns);
            n_alignments[q] = 0;
            if (status[q] != VGK_OK) { r->status = status[q]; continue; }
            const uint32_t mine = a++;
            if (probs[mine].status != VGK_OK) { r->status = probs[mine].status; continue; }
            if (used + pops[mine].size() > ops_cap || (!ops && !pops[mine].empty())) { r->status = VGK_EOPS; rc_all = VGK_EOPS; continue; }
            for (size_t k = 0; k < pres[mine].size(); ++k) { r[k] = pres[mine][k]; r[k].ops_begin += (uint32_t)used; }
            std::copy(pops[mine].begin(), pops[mine].end(), ops + used); used += pops[mine].size();
            n_alignments[q] = (uint32_t)pres[mine].size();
        }
        i = j;
    }
    if (ops_written) *ops_written = used;
    return rc_all;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

}  // extern "C"
