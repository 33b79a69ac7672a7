// xdrop_band_api.cpp — vgk_xdrop_band_align: pinned X-drop extension WITH dozeu's band (include/vgk.h states the rules;
// reference call sites src/dozeu_interface.cpp:226, :261-283, src/xdrop_aligner.cpp:95-109).
//
// Device: one wavefront per problem fills the H / E / F columns, one of dozeu's 8-cell vectors per lane, trimming the front after
// every column (gssw_matrix_device.hpp: xdrop_band_wave_lane).  Host: the matrices come back and one host thread per problem picks
// the end cell and walks the traceback — the same split as the k-best pinned path (gssw_multi_api.cpp); VGK_XDROP_PINNED through
// vgk_gssw_* (every cell kept, 4-bit codes, traceback on the device) stays the default and the fast path.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {

inline uint8_t code_read(char ch) {
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline uint8_t code_ref(char ch) {      // dozeu sees the raw node sequences; anything but upper-case ACGT scores as N
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

struct BandHost { RawBuf<uint8_t> reads, quals, graph; RawBuf<MProb> probs; RawBuf<MNode> nodes; RawBuf<uint32_t> preds; RawBuf<int32_t> cells; };

// end cell + traceback of one problem over its trimmed matrices: first node in order / first column / smallest row with the best
// score; preference diagonal > deletion > insertion, gap open before extend, first explaining predecessor; ends at the root
class BandTracer {
public:
    BandTracer(const vgk_ctx* ctx, const vgk_gssw_problem& p, const MProb& pb, const int32_t* cells)
        : p(p), L((int)pb.L), rows((int)pb.L + 1), R((int)pb.R), go(ctx->sc.gap_open), ge(ctx->sc.gap_extend), gap_cells(pb.gap_cells), bonus(pb.start_bonus),
          H(cells), E(cells + (size_t)pb.R * (pb.L + 1)), F(cells + 2 * (size_t)pb.R * (pb.L + 1)) {
        const vgk_graph& g = p.graph;
        col0.resize(g.n_nodes + 1, 0);
        for (uint32_t v = 0; v < g.n_nodes; ++v) col0[v + 1] = col0[v] + (int)g.node_len[v];
        node_of.resize((size_t)R);
        for (uint32_t v = 0; v < g.n_nodes; ++v) for (int c = col0[v]; c < col0[v + 1]; ++c) node_of[(size_t)c] = (int)v;
        qa = ctx->has_qa; mat = qa ? ctx->qmat.data() : ctx->sc.matrix;
    }
    int run(bool want_tb, vgk_result& res, std::vector<vgk_op>& o) const {
        std::memset(&res, 0, sizeof res);
        res.end_node = -1; res.end_offset = -1; res.end_read = -1;
        int32_t best = 0; int best_c = -1, best_i = 0;
        for (int c = 0; c < R; ++c) {
            int32_t colmax = MNEG; int at = -1;
            for (int i = 0; i < rows; ++i) if (h(c, i) > colmax) { colmax = h(c, i); at = i; }
            if (colmax > best) { best = colmax; best_c = c; best_i = at; }
        }
        if (best >= 32767) return VGK_EOVERFLOW;
        res.score = best;
        if (best <= 0 || best_c < 0) { res.score = 0; return VGK_OK; }      // the root wins: the caller writes the full-length insertion
        res.end_node = node_of[(size_t)best_c]; res.end_offset = best_c - col0[(size_t)node_of[(size_t)best_c]]; res.end_read = best_i - 1;
        if (!want_tb) return VGK_OK;
        o.clear();
        int c = best_c, i = best_i; int32_t cur = best;
        push(o, (uint32_t)node_of[(size_t)c], VGK_OP_S, (uint32_t)(L - i));
        enum { ST_H, ST_E, ST_F } st = ST_H;
        const vgk_graph& g = p.graph;
        for (bool done = false; !done;) {
            const int n = node_of[(size_t)c];
            const bool first = c == col0[(size_t)n];
            const uint32_t pb = g.pred_off[n], pe = g.pred_off[n + 1];
            if (st == ST_H) {
                bool moved = false;
                if (i > 0) {
                    int32_t d = first ? (pe == pb ? root_h(i - 1) : MNEG) : h(c - 1, i - 1);
                    if (first) for (uint32_t k = pb; k < pe; ++k) d = std::max(d, h(col0[g.pred_idx[k] + 1] - 1, i - 1));
                    if (live(d) && cur == d + score(i, c)) {
                        push(o, (uint32_t)n, VGK_OP_M, 1);
                        cur = d; i -= 1; moved = true;
                        if (!first) c -= 1;
                        else if (pe == pb) { push(o, (uint32_t)n, VGK_OP_I, (uint32_t)i); done = true; }      // back at the root: leading insertion
                        else {
                            int found = -1;
                            for (uint32_t k = pb; k < pe; ++k) { const int q = col0[g.pred_idx[k] + 1] - 1; if (h(q, i) == cur) { found = q; break; } }
                            if (found < 0) return VGK_EINVAL;
                            c = found;
                        }
                    }
                }
                if (!moved) {
                    if (cur == e(c, i)) st = ST_E;
                    else if (i > 0 && cur == f(c, i)) st = ST_F;
                    else return VGK_EINVAL;
                }
            } else if (st == ST_E) {
                push(o, (uint32_t)n, VGK_OP_D, 1);
                if (first && pe == pb) {                         // deletion opened straight from the root column
                    if (!live(root_h(i)) || root_h(i) - go != cur) return VGK_EINVAL;
                    push(o, (uint32_t)n, VGK_OP_I, (uint32_t)i); done = true;
                } else {
                    int q = c - 1;
                    if (first) {
                        q = -1;
                        for (uint32_t k = pb; k < pe; ++k) { const int x = col0[g.pred_idx[k] + 1] - 1; if (e_next(x, i) == cur) { q = x; break; } }
                        if (q < 0) return VGK_EINVAL;
                    }
                    if (live(h(q, i)) && h(q, i) - go == cur) { st = ST_H; cur += go; } else cur += ge;
                    c = q;
                }
            } else {
                push(o, (uint32_t)n, VGK_OP_I, 1);
                if (i == 0) return VGK_EINVAL;
                if (live(h(c, i - 1)) && h(c, i - 1) - go == cur) { st = ST_H; cur += go; } else cur += ge;
                i -= 1;
            }
        }
        std::reverse(o.begin(), o.end());
        res.n_ops = (uint32_t)o.size(); res.first_offset = 0;
        return VGK_OK;
    }
private:
    const vgk_gssw_problem& p; int L, rows, R, go, ge, gap_cells, bonus; const int32_t *H, *E, *F;
    std::vector<int> col0, node_of; const int8_t* mat; bool qa;
    static bool live(int32_t v) { return v > MNEG / 2; }
    int32_t h(int c, int i) const { return H[(size_t)c * rows + i]; }
    int32_t e(int c, int i) const { return E[(size_t)c * rows + i]; }
    int32_t f(int c, int i) const { return F[(size_t)c * rows + i]; }
    int32_t e_next(int c, int i) const {                          // E of the column after c, from c's H and E
        const int32_t a = live(h(c, i)) ? h(c, i) - go : MNEG, b = live(e(c, i)) ? e(c, i) - ge : MNEG;
        return std::max(a, b);
    }
    int32_t root_h(int i) const { return i == 0 ? 0 : (i <= gap_cells && i <= L ? -(go + (i - 1) * ge) : MNEG); }
    int32_t score(int i, int c) const {
        const int ref = code_ref(p.graph.seq[c]), rd = code_read(p.read[i - 1]);
        return (qa ? mat[25 * p.qual[i - 1] + 5 * ref + rd] : mat[5 * ref + rd]) + (i == L ? bonus : 0);
    }
    static void push(std::vector<vgk_op>& o, uint32_t node, int op, uint32_t len) {
        if (!len) return;
        if (!o.empty() && o.back().node == node && o.back().op == op) { o.back().len = (uint16_t)(o.back().len + len); return; }
        vgk_op x{}; x.node = node; x.op = (uint8_t)op; x.len = (uint16_t)len; o.push_back(x);
    }
};

}  // namespace

extern "C" {

int vgk_xdrop_band_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                         vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written, uint64_t stats[2]) {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    if (ops_written) *ops_written = 0;
    if (stats) stats[0] = stats[1] = 0;
    for (uint32_t i = 0; i < n; ++i) if ((problems[i].flags & 15u) != VGK_XDROP_PINNED) return VGK_EINVAL;
    if (ctx->sc.gap_open < ctx->sc.gap_extend) return VGK_EUNSUPPORTED;          // the column scan needs gap_open >= gap_extend
    std::lock_guard<std::mutex> lock(ctx->mu);
    Backend* be = ctx->be.get();
    const bool qa = ctx->has_qa;
    if (!ctx->xband_host) ctx->xband_host = std::make_shared<BandHost>();
    BandHost& Hs = *static_cast<BandHost*>(ctx->xband_host.get());
    uint64_t budget = be->memory_bytes() ? be->memory_bytes() / 4 : (1ull << 30);
    if (const char* e = std::getenv("VGAMD_MAX_BATCH_BYTES")) budget = std::strtoull(e, nullptr, 10);
    // per-problem checks; what fails is answered in its own status
    std::vector<int> status(n, VGK_OK); std::vector<uint64_t> cols(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_gssw_problem& p = problems[i]; const vgk_graph& g = p.graph;
        int st = VGK_OK;
        if (!p.read_len || !p.read || !g.n_nodes || !g.node_len || !g.pred_off || !g.seq || (qa && !p.qual)) st = VGK_EINVAL;
        else if (p.read_len > 511) st = VGK_ETOOLONG;             // 64 lanes x one 8-row vector, rows 0 .. L
        else {
            uint64_t R = 0;
            for (uint32_t v = 0; v < g.n_nodes && st == VGK_OK; ++v) {
                if (!g.node_len[v] || g.pred_off[v + 1] < g.pred_off[v] || (g.pred_off[v + 1] > g.pred_off[v] && !g.pred_idx)) st = VGK_EINVAL;
                else if (g.node_len[v] > 65535u) st = VGK_ETOOBIG;
                else for (uint32_t k = g.pred_off[v]; k < g.pred_off[v + 1]; ++k) if (g.pred_idx[k] >= v) st = VGK_EINVAL;
                R += g.node_len[v];
            }
            if (st == VGK_OK && R * (p.read_len + 1ull) > (1ull << 28)) st = VGK_ETOOBIG;
            cols[i] = R;
        }
        status[i] = st;
    }
    size_t used = 0; int rc_all = VGK_OK; uint64_t in_band_total = 0, rect_total = 0; double ms = 0;
    for (uint32_t i = 0; i < n;) {
        uint64_t n_cells = 0, n_read = 0, n_graph = 0, n_nodes = 0, n_preds = 0;
        uint32_t j = i; std::vector<uint32_t> owner;
        for (; j < n; ++j) {
            if (status[j] != VGK_OK) continue;
            const vgk_gssw_problem& p = problems[j];
            const uint64_t c3 = 3ull * cols[j] * (p.read_len + 1ull);
            if (!owner.empty() && (n_cells + c3) * sizeof(int32_t) > budget) break;
            n_cells += c3; n_read += p.read_len; n_graph += cols[j]; n_nodes += p.graph.n_nodes; n_preds += p.graph.pred_off[p.graph.n_nodes] - p.graph.pred_off[0];
            owner.push_back(j);
        }
        const uint32_t m = (uint32_t)owner.size();
        std::vector<vgk_result> pres(m); std::vector<std::vector<vgk_op>> pops(m); std::vector<int> prc(m, VGK_OK);
        if (m) {
            MProb* probs = Hs.probs.get(m + 1);
            uint8_t* reads = Hs.reads.get(n_read + 1); uint8_t* quals = qa ? Hs.quals.get(n_read + 1) : nullptr; uint8_t* graph = Hs.graph.get(n_graph + 1);
            MNode* nodes = Hs.nodes.get(n_nodes + 1); uint32_t* preds = Hs.preds.get(n_preds + 1);
            uint64_t a_cells = 0, a_read = 0, a_graph = 0, a_nodes = 0, a_preds = 0;
            for (uint32_t a = 0; a < m; ++a) {
                const vgk_gssw_problem& p = problems[owner[a]];
                MProb pb{}; pb.L = p.read_len; pb.n_nodes = p.graph.n_nodes; pb.R = (uint32_t)cols[owner[a]];
                pb.read_off = (uint32_t)a_read; pb.graph_off = (uint32_t)a_graph; pb.node_off = (uint32_t)a_nodes; pb.mat_off = a_cells;
                pb.start_bonus = qa ? ctx->qbon[p.qual[p.read_len - 1]] : ctx->sc.full_length_bonus; pb.status = VGK_OK;
                const int32_t max_gap = (int32_t)std::max<uint32_t>(p.max_gap_length, 1u);
                pb.gap_cells = (max_gap + 7) & ~7; pb.xt = ((int32_t)ctx->sc.gap_open - (int32_t)ctx->sc.gap_extend) + (int32_t)ctx->sc.gap_extend * max_gap;
                probs[a] = pb;
                uint32_t col = 0;
                for (uint32_t v = 0; v < p.graph.n_nodes; ++v) {
                    MNode nd; nd.col_start = col; nd.col_end = col + p.graph.node_len[v]; nd.pred_begin = (uint32_t)a_preds; nd.n_pred = p.graph.pred_off[v + 1] - p.graph.pred_off[v];
                    for (uint32_t k = p.graph.pred_off[v]; k < p.graph.pred_off[v + 1]; ++k) preds[a_preds++] = p.graph.pred_idx[k];
                    nodes[a_nodes + v] = nd; col = nd.col_end;
                }
                a_cells += 3ull * pb.R * (pb.L + 1ull); a_read += pb.L; a_graph += pb.R; a_nodes += pb.n_nodes;
                rect_total += (uint64_t)pb.R * (pb.L + 1ull);
            }
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
            // (the scratch slots of the k-best pinned path: the two calls never overlap under the context lock)
            P.probs = (MProb*)dev(40, probs, sizeof(MProb) * m);
            P.reads = (const uint8_t*)dev(41, reads, n_read); P.quals = qa ? (const uint8_t*)dev(42, quals, n_read) : nullptr;
            P.graph = (const uint8_t*)dev(43, graph, n_graph); P.nodes = (const MNode*)dev(44, nodes, sizeof(MNode) * n_nodes);
            P.preds = (const uint32_t*)dev(45, preds, sizeof(uint32_t) * n_preds);
            P.mat = (const int8_t*)dev(46, qa ? ctx->qmat.data() : ctx->sc.matrix, qa ? 6400 : 25);
            P.cells = (int32_t*)dev(47, nullptr, sizeof(int32_t) * n_cells);
            P.node_fmax = (int32_t*)ctx->ensure_scratch(48, sizeof(int32_t) * (n_nodes + 1));
            P.stats = (unsigned long long*)ctx->ensure_scratch(49, 64);
            if (!P.probs || !P.reads || (qa && !P.quals) || !P.graph || !P.nodes || !P.preds || !P.mat || !P.cells || !P.node_fmax || !P.stats) return VGK_ENOMEM;
            int rc;
            if ((rc = be->zero(P.stats, 64))) return rc;
            if ((rc = be->run_xdrop_band(P))) return rc;
            ms += be->last_ms(7);
            unsigned long long in_band = 0;
            int32_t* cells = Hs.cells.get(n_cells + 1);
            if ((rc = be->download(&in_band, P.stats, sizeof in_band))) return rc;
            if ((rc = be->download(cells, P.cells, sizeof(int32_t) * n_cells))) return rc;
            in_band_total += in_band;
            parallel_for(m, [&](uint32_t a, unsigned) {
                BandTracer t(ctx, problems[owner[a]], probs[a], cells + probs[a].mat_off);
                prc[a] = t.run((problems[owner[a]].flags & VGK_GSSW_TRACEBACK) != 0, pres[a], pops[a]);
            });
        }
        uint32_t a = 0;
        for (uint32_t q = i; q < j; ++q) {
            vgk_result& r = results[q];
            if (status[q] != VGK_OK) { std::memset(&r, 0, sizeof r); r.status = status[q]; r.ops_begin = (uint32_t)used; continue; }
            const uint32_t mine = a++;
            r = pres[mine]; r.status = prc[mine]; r.ops_begin = (uint32_t)used;
            if (r.status != VGK_OK) { r.n_ops = 0; continue; }
            if (used + pops[mine].size() > ops_cap || (!ops && !pops[mine].empty())) { r.status = VGK_EOPS; r.n_ops = 0; rc_all = VGK_EOPS; continue; }
            std::copy(pops[mine].begin(), pops[mine].end(), ops + used); used += pops[mine].size();
        }
        i = j;
    }
    ctx->xband_ms = ms;
    if (ops_written) *ops_written = used;
    if (stats) { stats[0] = in_band_total; stats[1] = rect_total; }
    return rc_all;
}

double vgk_xdrop_band_last_ms(vgk_ctx* ctx) { return ctx ? ctx->xband_ms : 0.0; }

}  // extern "C"
