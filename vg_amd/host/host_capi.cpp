// host_capi.cpp — a small C surface over the C++ host shim so that other
// languages (the pytest suite via ctypes, or a future binding) can drive the
// mirrored Aligner interface exactly like src/unittest/*.cpp drives vg's.
#include <array>
#include <set>
#include <algorithm>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include "aligner.hpp"
#include "tail_stage.hpp"
#include "gbwt_extender.hpp"
#include "rescue_fixups.hpp"
#include "mapq_cap.hpp"
#include "seed_policy.hpp"
#include "extension_scoring.hpp"
#include "aligner_client.hpp"
#include "rescue_stage.hpp"
#include "rescue_resident.hpp"
#include "rescue_requests.hpp"
#include <sstream>

using namespace vgamd;

extern "C" {

struct vgh_graph { HashGraph g; };
struct vgh_aligner { std::unique_ptr<Aligner> a; };

static thread_local std::string g_last_error;
const char* vgh_last_error(void) { return g_last_error.c_str(); }

vgh_graph* vgh_graph_create(void) { return new vgh_graph(); }
void vgh_graph_destroy(vgh_graph* g) { delete g; }
int vgh_graph_add_node(vgh_graph* g, int64_t id, const char* seq) {
    try { g->g.create_handle(seq, id); return 0; } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// the same in bulk (graphs of millions of nodes): node k has id ids[k] and the bases seq[seq_off[k], seq_off[k + 1]); edge k joins from[k] -> to[k], forward strands
int vgh_graph_add_nodes(vgh_graph* g, uint64_t n, const int64_t* ids, const char* seq, const uint64_t* seq_off) {
    try { for (uint64_t k = 0; k < n; ++k) g->g.create_handle(std::string(seq + seq_off[k], seq + seq_off[k + 1]), ids[k]); return 0; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
int vgh_graph_add_edges(vgh_graph* g, uint64_t n, const int64_t* from, const int64_t* to) {
    try { for (uint64_t k = 0; k < n; ++k) g->g.create_edge(g->g.get_handle(from[k]), g->g.get_handle(to[k])); return 0; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
int vgh_graph_add_edge(vgh_graph* g, int64_t from, int64_t to) {
    try { g->g.create_edge(g->g.get_handle(from), g->g.get_handle(to)); return 0; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// engine_lib: NULL/"" = the HIP product library (fails loudly without it);
// tests may pass the oracle's path to exercise the host logic on a CPU box.
static vgh_aligner* make_aligner(const char* engine_lib, int device, int match, int mismatch,
                                 int gap_open, int gap_extend, int full_length_bonus, bool qual_adj) {
    try {
        int8_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? (int8_t)match : (int8_t)-mismatch;   // src/aligner.cpp:1395-1413
        auto eng = load_engine(engine_lib ? engine_lib : "");
        auto* h = new vgh_aligner();
        if (qual_adj) h->a = std::make_unique<QualAdjAligner>(m, (int8_t)gap_open, (int8_t)gap_extend, (int8_t)full_length_bonus, 0.5, eng, device);
        else h->a = std::make_unique<Aligner>(m, (int8_t)gap_open, (int8_t)gap_extend, (int8_t)full_length_bonus, 0.5, eng, device);
        return h;
    } catch (std::exception& e) { g_last_error = e.what(); return nullptr; }
}
vgh_aligner* vgh_aligner_create(const char* engine_lib, int device, int match, int mismatch,
                                int gap_open, int gap_extend, int full_length_bonus) {
    return make_aligner(engine_lib, device, match, mismatch, gap_open, gap_extend, full_length_bonus, false);
}
// AlignerClient::get_qual_adj_aligner(): quality-adjusted twin; reads then need qualities (vgh_align_q)
vgh_aligner* vgh_qual_adj_aligner_create(const char* engine_lib, int device, int match, int mismatch,
                                         int gap_open, int gap_extend, int full_length_bonus) {
    return make_aligner(engine_lib, device, match, mismatch, gap_open, gap_extend, full_length_bonus, true);
}
void vgh_aligner_destroy(vgh_aligner* a) { delete a; }
// X-drop calls with dozeu's band restated instead of the exact extension (GSSWAligner::xdrop_band)
void vgh_aligner_set_xdrop_band(vgh_aligner* a, int on) { a->a->xdrop_band = on != 0; }

static int emit(const Alignment& aln, char* out, size_t cap) {
    std::string js = alignment_to_json(aln);
    if (js.size() + 1 > cap) { g_last_error = "json buffer too small"; return -2; }
    std::memcpy(out, js.c_str(), js.size() + 1);
    return 0;
}

// call: 0 = align(traceback), 1 = align(score only), 2 = align_pinned, 3 = align_pinned_multi (JSON list of all alternates),
//       4 = align_pinned(xdrop = true, max_gap = max_alt_alns argument)
//       5 = align_global_banded(band_padding = max_alt_alns argument, permissive_banding = pin_left argument)
int vgh_align_q(vgh_aligner* a, vgh_graph* g, const char* read, const uint8_t* qual, int call, int pin_left, int max_alt_alns,
                char* json_out, size_t json_cap);
int vgh_align(vgh_aligner* a, vgh_graph* g, const char* read, int call, int pin_left, int max_alt_alns,
              char* json_out, size_t json_cap) {
    return vgh_align_q(a, g, read, nullptr, call, pin_left, max_alt_alns, json_out, json_cap);
}
// same with raw phred base qualities (one per read base) for a quality-adjusted aligner
int vgh_align_q(vgh_aligner* a, vgh_graph* g, const char* read, const uint8_t* qual, int call, int pin_left, int max_alt_alns,
                char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = read;
        if (qual) aln.quality.assign(reinterpret_cast<const char*>(qual), aln.sequence.size());
        switch (call) {
            case 0: a->a->align(aln, g->g, true); break;
            case 1: a->a->align(aln, g->g, false); break;
            case 2: a->a->align_pinned(aln, g->g, pin_left != 0); break;
            case 3: {
                std::vector<Alignment> alts; a->a->align_pinned_multi(aln, alts, g->g, pin_left != 0, max_alt_alns);
                std::string js = "[";                         // every alternate, best first
                for (size_t k = 0; k < alts.size(); ++k) js += std::string(k ? "," : "") + alignment_to_json(alts[k]);
                js += "]";
                if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
                std::memcpy(json_out, js.c_str(), js.size() + 1);
                return 0;
            }
            case 4: a->a->align_pinned(aln, g->g, pin_left != 0, true, (uint16_t)max_alt_alns); break;
            case 5: a->a->align_global_banded(aln, g->g, max_alt_alns, pin_left != 0); break;
            default: g_last_error = "unknown call"; return -1;
        }
        return emit(aln, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// Aligner::align(alignment, graph, topological_order) (src/aligner.hpp:180-181): the order names oriented nodes (2 * id + is_reverse)
// of a subgraph that may hold both strands; only a plain Aligner has this overload.
int vgh_align_order(vgh_aligner* a, vgh_graph* g, const char* read, const int64_t* order, int n_order, char* json_out, size_t json_cap) {
    try {
        Aligner* plain = dynamic_cast<Aligner*>(a->a.get());
        if (!plain) { g_last_error = "align(order) needs a plain Aligner"; return -1; }
        Alignment aln; aln.sequence = read;
        std::vector<handle_t> topological_order;
        for (int i = 0; i < n_order; ++i) topological_order.push_back(g->g.get_handle(order[i] >> 1, order[i] & 1));
        plain->align(aln, g->g, topological_order);
        return emit(aln, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// MatrixAlignmentScorer::longest_detectable_gap (src/alignment_scorer.cpp:264-271) of the aligner's scorer
int64_t vgh_longest_detectable_gap(vgh_aligner* a, int64_t read_length, int64_t read_pos) {
    return (int64_t)a->a->scorer->longest_detectable_gap((size_t)read_length, (size_t)read_pos);
}

// Aligner::align_global_banded_multi: JSON out = {"primary": alignment, "alternates": [alignment...]}
int vgh_align_banded_multi(vgh_aligner* a, vgh_graph* g, const char* read, const uint8_t* qual, int max_alt_alns, int band_padding, int permissive,
                           char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = read;
        if (qual) aln.quality.assign(reinterpret_cast<const char*>(qual), aln.sequence.size());
        std::vector<Alignment> alts;
        a->a->align_global_banded_multi(aln, alts, g->g, max_alt_alns, band_padding, permissive != 0);
        std::string js = "{\"primary\":" + alignment_to_json(aln) + ",\"alternates\":[";
        for (size_t i = 0; i < alts.size(); ++i) { if (i) js += ','; js += alignment_to_json(alts[i]); }
        js += "]}";
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// Aligner::align_xdrop with MEMs given as flat records [begin, end, node_id, offset, is_reverse] x n_mems
int vgh_align_xdrop(vgh_aligner* a, vgh_graph* g, const char* read, const int64_t* mems, int n_mems,
                    int reverse_complemented, int max_gap, char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = read;
        std::vector<MaximalExactMatch> ms;
        for (int i = 0; i < n_mems; ++i) {
            MaximalExactMatch m; m.begin = (size_t)mems[5 * i]; m.end = (size_t)mems[5 * i + 1];
            m.nodes.push_back({mems[5 * i + 2], (size_t)mems[5 * i + 3], mems[5 * i + 4] != 0});
            ms.push_back(m);
        }
        a->a->align_xdrop(aln, g->g, ms, reverse_complemented != 0, (uint16_t)max_gap);
        return emit(aln, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// Aligner::align_xdrop_many: n reads, each with its graph and its MEMs (mems flat as above, n_mems[k] of them for read k), followed — as
// MinimizerMapper::attempt_rescue follows align_xdrop — by fix_dozeu_score and fix_dozeu_end_deletions when `rescue_fixups` is set.
// JSON out: a list of the n alignments.
int vgh_align_xdrop_many(vgh_aligner* a, vgh_graph** graphs, const char** reads, const int64_t* mems, const int* n_mems, int n,
                         int reverse_complemented, int max_gap, int rescue_fixups, char* json_out, size_t json_cap) {
    try {
        std::deque<Alignment> alns((size_t)n);
        std::vector<Aligner::XdropRequest> requests((size_t)n);
        size_t at = 0;
        for (int k = 0; k < n; ++k) {
            alns[(size_t)k].sequence = reads[k];
            Aligner::XdropRequest& rq = requests[(size_t)k];
            rq.alignment = &alns[(size_t)k]; rq.graph = &graphs[k]->g; rq.reverse_complemented = reverse_complemented != 0; rq.max_gap_length = (uint16_t)max_gap;
            for (int i = 0; i < n_mems[k]; ++i, ++at) {
                MaximalExactMatch m; m.begin = (size_t)mems[5 * at]; m.end = (size_t)mems[5 * at + 1];
                m.nodes.push_back({mems[5 * at + 2], (size_t)mems[5 * at + 3], mems[5 * at + 4] != 0});
                rq.mems.push_back(m);
            }
        }
        a->a->align_xdrop_many(requests);
        std::string js = "[";
        for (int k = 0; k < n; ++k) {
            if (rescue_fixups) { fix_dozeu_score(alns[(size_t)k], *a->a, graphs[k]->g, requests[(size_t)k].order); fix_dozeu_end_deletions(alns[(size_t)k]); }
            js += std::string(k ? "," : "") + alignment_to_json(alns[(size_t)k]);
        }
        js += "]";
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// run_rescue_stage (rescue_stage.hpp): requests flat, 6 numbers each {node_lo, node_hi, seed_begin, seed_end, seed_node (-1: none), seed_offset}; reads flat
// with offsets; out[6 k ..] = {score, status, first_node, first_offset, n_mappings, aligned read bases}
int vgh_rescue_stage(vgh_aligner* a, uint32_t n_nodes, const uint32_t* node_len, const uint64_t* seq_off, const char* seq, const uint32_t* succ_off, const uint32_t* succ,
                     int n, const char* reads, const uint64_t* read_off, const int64_t* requests, uint64_t max_dozeu_cells, int host_threads, int64_t* out) {
    try {
        RescueGraph G; G.n_nodes = n_nodes; G.node_len = node_len; G.seq_off = seq_off; G.seq = seq; G.succ_off = succ_off; G.succ = succ;
        std::vector<RescueRequest> rq((size_t)n);
        for (int k = 0; k < n; ++k) {
            RescueRequest& r = rq[(size_t)k]; const int64_t* q = requests + 6 * (size_t)k;
            r.read = reads + read_off[k]; r.read_len = (uint32_t)(read_off[k + 1] - read_off[k]);
            r.node_lo = (uint32_t)q[0]; r.node_hi = (uint32_t)q[1]; r.seed_begin = q[2]; r.seed_end = q[3]; r.seed_node = q[4]; r.seed_offset = q[5];
        }
        std::vector<RescueResult> res;
        run_rescue_stage(*a->a, G, rq, max_dozeu_cells ? max_dozeu_cells : default_max_dozeu_cells, (unsigned)std::max(host_threads, 0), res);
        for (int k = 0; k < n; ++k) {
            const RescueResult& r = res[(size_t)k]; int64_t* o = out + 6 * (size_t)k;
            o[0] = r.score; o[1] = r.status; o[2] = r.first_node; o[3] = r.first_offset; o[4] = r.n_mappings; o[5] = r.aligned_read_bases;
        }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// the same with the final alignments as op runs: ops_begin[n + 1], ops up to ops_cap (*ops_written = what there is; -2 when that does not fit)
int vgh_rescue_stage_ops(vgh_aligner* a, uint32_t n_nodes, const uint32_t* node_len, const uint64_t* seq_off, const char* seq, const uint32_t* succ_off, const uint32_t* succ,
                         int n, const char* reads, const uint64_t* read_off, const int64_t* requests, uint64_t max_dozeu_cells, int host_threads, int64_t* out,
                         uint64_t* ops_begin, vgk_op* ops, uint64_t ops_cap, uint64_t* ops_written) {
    try {
        RescueGraph G; G.n_nodes = n_nodes; G.node_len = node_len; G.seq_off = seq_off; G.seq = seq; G.succ_off = succ_off; G.succ = succ;
        std::vector<RescueRequest> rq((size_t)n);
        for (int k = 0; k < n; ++k) {
            RescueRequest& r = rq[(size_t)k]; const int64_t* q = requests + 6 * (size_t)k;
            r.read = reads + read_off[k]; r.read_len = (uint32_t)(read_off[k + 1] - read_off[k]);
            r.node_lo = (uint32_t)q[0]; r.node_hi = (uint32_t)q[1]; r.seed_begin = q[2]; r.seed_end = q[3]; r.seed_node = q[4]; r.seed_offset = q[5];
        }
        std::vector<RescueResult> res; std::vector<vgk_op> o; std::vector<uint64_t> ob;
        run_rescue_stage(*a->a, G, rq, max_dozeu_cells ? max_dozeu_cells : default_max_dozeu_cells, (unsigned)std::max(host_threads, 0), res, &o, &ob);
        for (int k = 0; k < n; ++k) {
            const RescueResult& r = res[(size_t)k]; int64_t* q = out + 6 * (size_t)k;
            q[0] = r.score; q[1] = r.status; q[2] = r.first_node; q[3] = r.first_offset; q[4] = r.n_mappings; q[5] = r.aligned_read_bases;
        }
        if (ops_written) *ops_written = o.size();
        if (ops_begin) std::copy(ob.begin(), ob.end(), ops_begin);
        if (o.size() > ops_cap) { g_last_error = "vgh_rescue_stage_ops: op array too small"; return -2; }
        if (ops) std::copy(o.begin(), o.end(), ops);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// ---- the same stage on the resident graph (rescue_resident.hpp): the graph goes to the aligner's engine context once, a batch of requests is
// flat arrays: requests 6 numbers each as above, reads flat with offsets
struct vgh_rescue_graph { std::unique_ptr<ResidentRescueGraph> g; std::vector<uint32_t> node_len, pred_off, pred_idx; std::vector<char> seq; };
vgh_rescue_graph* vgh_rescue_graph_create(vgh_aligner* a, uint32_t n_nodes, const uint32_t* node_len, const char* seq, const uint32_t* pred_off, const uint32_t* pred_idx) {
    try {
        auto* h = new vgh_rescue_graph();
        std::unique_ptr<vgh_rescue_graph> hold(h);
        h->node_len.assign(node_len, node_len + n_nodes); h->pred_off.assign(pred_off, pred_off + n_nodes + 1);
        h->pred_idx.assign(pred_idx, pred_idx + pred_off[n_nodes]);
        uint64_t bases = 0; for (uint32_t v = 0; v < n_nodes; ++v) bases += node_len[v];
        h->seq.assign(seq, seq + bases);
        h->g = std::make_unique<ResidentRescueGraph>(*a->a, n_nodes, h->node_len.data(), h->seq.data(), h->pred_off.data(), h->pred_idx.empty() ? nullptr : h->pred_idx.data());
        return hold.release();
    } catch (std::exception& e) { g_last_error = e.what(); return nullptr; }
}
void vgh_rescue_graph_destroy(vgh_rescue_graph* g) { delete g; }
// laps (nullable, 6): classify, first pass, second pass, alignments + fix-ups, full-DP fallback, kernels of the three rounds (ms); counts (nullable, 6): first-pass extensions,
// scans, second-pass extensions, fallbacks, algorithmic bytes of the rounds' batches, their DP cells
int vgh_rescue_stage_resident(vgh_aligner* a, vgh_rescue_graph* g, int n, const char* reads, uint64_t reads_bytes, const uint64_t* read_off, const int64_t* requests,
                              uint64_t max_dozeu_cells, int host_threads, int64_t* out, uint64_t* ops_begin, vgk_op* ops, uint64_t ops_cap, uint64_t* ops_written,
                              double* laps, uint64_t* counts) {
    try {
        std::vector<RescueRequestFlat> rq((size_t)n);
        for (int k = 0; k < n; ++k) {
            RescueRequestFlat& r = rq[(size_t)k]; const int64_t* q = requests + 6 * (size_t)k;
            r.read_off = read_off[k]; r.read_len = (uint32_t)(read_off[k + 1] - read_off[k]);
            r.node_lo = (uint32_t)q[0]; r.node_hi = (uint32_t)q[1]; r.seed_begin = q[2]; r.seed_end = q[3]; r.seed_node = q[4]; r.seed_offset = q[5];
        }
        std::vector<RescueResult> res; std::vector<vgk_op> o; std::vector<uint64_t> ob; RescueTiming tm;
        const bool want_ops = ops_begin != nullptr;
        run_rescue_stage_resident(*a->a, *g->g, reads, (size_t)reads_bytes, rq, max_dozeu_cells ? max_dozeu_cells : default_max_dozeu_cells, (unsigned)std::max(host_threads, 0),
                                  res, want_ops ? &o : nullptr, want_ops ? &ob : nullptr, &tm);
        for (int k = 0; k < n; ++k) {
            const RescueResult& r = res[(size_t)k]; int64_t* q = out + 6 * (size_t)k;
            q[0] = r.score; q[1] = r.status; q[2] = r.first_node; q[3] = r.first_offset; q[4] = r.n_mappings; q[5] = r.aligned_read_bases;
        }
        if (laps) { laps[0] = tm.classify_ms; laps[1] = tm.first_pass_ms; laps[2] = tm.second_pass_ms; laps[3] = tm.finish_ms; laps[4] = tm.fallback_ms; }
        if (counts) { counts[0] = tm.first_pass; counts[1] = tm.scans; counts[2] = tm.second_pass; counts[3] = tm.fallbacks; counts[4] = tm.alg_bytes; counts[5] = tm.cells; }
        if (laps) laps[5] = tm.kernel_ms;
        if (ops_written) *ops_written = o.size();
        if (want_ops) {
            std::copy(ob.begin(), ob.end(), ops_begin);
            if (o.size() > ops_cap) { g_last_error = "vgh_rescue_stage_resident: op array too small"; return -2; }
            if (ops) std::copy(o.begin(), o.end(), ops);
        }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// the request table of a batch of pairs (rescue_requests.hpp) — returns the number of rescued pairs (-1: error); outputs sized for n_pairs entries
int64_t vgh_rescue_requests(uint32_t n_pairs, const vgk_gapless_result* results, const vgk_extension* extensions, const uint32_t* nodes, uint32_t n_nodes, const int64_t* col,
                            const char* reads, uint32_t read_len, double fragment_mean, double fragment_sd, double rescue_stdevs, int host_threads,
                            uint32_t* mapped, uint32_t* lost, int64_t* requests, char* rescue_reads) {
    try {
        RescueRequestTable t;
        build_rescue_requests(n_pairs, results, extensions, nodes, n_nodes, col, reads, read_len, fragment_mean, fragment_sd, rescue_stdevs, (unsigned)std::max(host_threads, 0), t);
        std::copy(t.mapped.begin(), t.mapped.end(), mapped); std::copy(t.lost.begin(), t.lost.end(), lost);
        std::copy(t.requests.begin(), t.requests.end(), requests); std::copy(t.reads.begin(), t.reads.end(), rescue_reads);
        return (int64_t)t.mapped.size();
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// the resident graph as the engine knows it (a vgk_dgraph of the aligner's context: vgk_rescue_requests takes it from any context of the same device)
const void* vgh_rescue_graph_dgraph(vgh_rescue_graph* g) { return g && g->g ? (const void*)g->g->dg : nullptr; }
// The table vgk_rescue_requests made on the device, in the form vgh_rescue_stage_resident takes (mapped, lost, 6 numbers per request as above) with the
// lost mates' reads as they read along the forward strand of their subgraphs (read_len each; reverse-complemented where the request says so): the
// one part of the table that needs the reads' own bytes, which are the host's (the engine keeps them masked).  Chunked host threads; a read is
// three cache lines and a table lookup per base.
int vgh_rescue_reads(uint32_t m, const vgk_rescue_request* table, const char* reads, uint32_t read_len, int host_threads,
                     uint32_t* mapped, uint32_t* lost, int64_t* requests, char* rescue_reads) {
    try {
        static const std::array<char, 256> comp = [] { std::array<char, 256> t{}; for (int c = 0; c < 256; ++c) t[(size_t)c] = (char)c; t['A'] = 'T'; t['C'] = 'G'; t['G'] = 'C'; t['T'] = 'A'; return t; }();
        unsigned threads = host_threads > 0 ? (unsigned)host_threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(m / 2048, 1));
        auto body = [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                if (k + 4 < hi) { const char* nx = reads + (size_t)table[k + 4].lost * read_len; __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); }
                const vgk_rescue_request& r = table[k];
                mapped[k] = r.mapped; lost[k] = r.lost;
                int64_t* q = requests + 6 * k;
                q[0] = r.node_lo; q[1] = r.node_hi; q[2] = r.seed_begin; q[3] = r.seed_end; q[4] = r.seed_node; q[5] = r.seed_offset;
                const char* src = reads + (size_t)r.lost * read_len; char* dst = rescue_reads + k * (size_t)read_len;
                if (r.reverse) for (uint32_t t = 0; t < read_len; ++t) dst[t] = comp[(unsigned char)src[read_len - 1 - t]]; else std::copy(src, src + read_len, dst);
            }
        };
        if (threads < 2) { body(0, m); return 0; }
        std::vector<std::thread> ts; const size_t per = ((size_t)m + threads - 1) / threads;
        std::mutex first_mutex; std::exception_ptr first;                // (as everywhere in the shim: a worker's exception comes back on the caller)
        auto guarded = [&](size_t lo, size_t hi) { try { if (lo < hi) body(lo, hi); } catch (...) { std::lock_guard<std::mutex> hold(first_mutex); if (!first) first = std::current_exception(); } };
        for (unsigned t = 1; t < threads; ++t) ts.emplace_back([&, t]() { const size_t lo = std::min<size_t>(m, t * per); guarded(lo, std::min<size_t>(m, lo + per)); });
        guarded(0, std::min<size_t>(m, per));
        for (auto& t : ts) t.join();
        if (first) std::rethrow_exception(first);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// ---- AlignmentBatch: the same calls, deferred; one engine launch per kernel family at flush ------------------------------------------
struct vgh_batch { std::unique_ptr<AlignmentBatch> b; std::deque<Alignment> alns; };
vgh_batch* vgh_batch_create(vgh_aligner* a) { auto* h = new vgh_batch(); h->b = std::make_unique<AlignmentBatch>(*a->a); return h; }
// one aligner (engine context) per device, flushes go to them in turn; max_pending > 0: the submission that fills the batch flushes it
vgh_batch* vgh_batch_create_multi(vgh_aligner** aligners, int n, int max_pending) {
    try {
        std::vector<const Aligner*> per_device;
        for (int i = 0; i < n; ++i) per_device.push_back(aligners[i]->a.get());
        auto* h = new vgh_batch(); h->b = std::make_unique<AlignmentBatch>(per_device, (size_t)std::max(max_pending, 0)); return h;
    } catch (std::exception& e) { g_last_error = e.what(); return nullptr; }
}
int vgh_batch_flushes(vgh_batch* b) { return (int)b->b->flushes(); }
void vgh_batch_destroy(vgh_batch* b) { delete b; }
// call codes as in vgh_align: 0 align(traceback), 1 align(score only), 2 align_pinned, 4 align_pinned(xdrop = true, max_gap = arg),
// 5 align_global_banded(band_padding = arg, permissive = pin_left)
int vgh_batch_add(vgh_batch* b, vgh_graph* g, const char* read, const uint8_t* qual, int call, int pin_left, int arg) {
    try {
        b->alns.emplace_back();
        Alignment& aln = b->alns.back(); aln.sequence = read;
        if (qual) aln.quality.assign(reinterpret_cast<const char*>(qual), aln.sequence.size());
        switch (call) {
            case 0: b->b->align(aln, g->g, true); break;
            case 1: b->b->align(aln, g->g, false); break;
            case 2: b->b->align_pinned(aln, g->g, pin_left != 0); break;
            case 4: b->b->align_pinned(aln, g->g, pin_left != 0, true, (uint16_t)arg); break;
            case 5: b->b->align_global_banded(aln, g->g, arg, pin_left != 0); break;
            default: b->alns.pop_back(); g_last_error = "unknown call"; return -1;
        }
        return 0;
    } catch (std::exception& e) { b->alns.pop_back(); g_last_error = e.what(); return -1; }
}
// runs everything added so far; JSON out = [alignment, ...] in submission order
int vgh_batch_flush(vgh_batch* b, char* json_out, size_t json_cap) {
    try {
        b->b->flush();
        std::string js = "[";
        for (size_t i = 0; i < b->alns.size(); ++i) { if (i) js += ','; js += alignment_to_json(b->alns[i]); }
        js += "]";
        b->alns.clear();
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { b->alns.clear(); g_last_error = e.what(); return -1; }
}

// ---- GaplessExtender (src/gbwt_extender.hpp:140-217) over a HaplotypeGraph built from `g` and explicit threads ----------
struct vgh_extender { std::unique_ptr<HaplotypeGraph> graph; std::unique_ptr<GaplessExtender> ext; };
// threads: oriented node ids (2 * id + is_reverse), thread t = thread_nodes[thread_off[t] .. thread_off[t + 1])
vgh_extender* vgh_gapless_create(vgh_aligner* a, vgh_graph* g, const int64_t* thread_nodes, const int32_t* thread_off, int n_threads) {
    try {
        std::vector<std::vector<handle_t>> threads((size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) for (int32_t k = thread_off[t]; k < thread_off[t + 1]; ++k) threads[t].push_back(g->g.get_handle(thread_nodes[k] >> 1, thread_nodes[k] & 1));
        auto* h = new vgh_extender();
        h->graph = std::make_unique<HaplotypeGraph>(g->g, threads);
        h->ext = std::make_unique<GaplessExtender>(*h->graph, *a->a);
        return h;
    } catch (std::exception& e) { g_last_error = e.what(); return nullptr; }
}
void vgh_gapless_destroy(vgh_extender* e) { delete e; }
// seeds: [node id, is_reverse, node offset, read offset] x n_seeds; JSON out: one object per extension
int vgh_gapless_extend(vgh_extender* x, const char* read, const int64_t* seeds, int n_seeds, int max_mismatches, double overlap_threshold,
                       int trim, char* json_out, size_t json_cap) {
    try {
        GaplessExtender::cluster_type cluster;
        for (int i = 0; i < n_seeds; ++i) {
            Position pos; pos.node_id = seeds[4 * i]; pos.is_reverse = seeds[4 * i + 1] != 0; pos.offset = seeds[4 * i + 2];
            cluster.push_back(GaplessExtender::to_seed(*x->graph, pos, (size_t)seeds[4 * i + 3]));
        }
        const std::string sequence(read);
        auto result = x->ext->extend(cluster, sequence, (size_t)max_mismatches, overlap_threshold, trim != 0);
        std::string js = "{\"full_length\":";
        js += GaplessExtender::full_length_extensions(result, (size_t)max_mismatches) ? "true" : "false";
        js += ",\"extensions\":[";
        for (size_t i = 0; i < result.size(); ++i) {
            const GaplessExtension& e = result[i];
            Alignment aln; aln.sequence = sequence; aln.path = e.to_path(*x->graph, sequence); aln.score = e.score;
            bool all = true; for (const auto& s : cluster) all = all && e.contains(*x->graph, s);
            if (i) js += ',';
            js += "{\"alignment\":" + alignment_to_json(aln) + ",\"read_begin\":" + std::to_string(e.read_interval.first) + ",\"read_end\":" + std::to_string(e.read_interval.second) +
                  ",\"mismatches\":" + std::to_string(e.mismatches()) + ",\"left_full\":" + (e.left_full ? "true" : "false") + ",\"right_full\":" + (e.right_full ? "true" : "false") +
                  ",\"contains_all_seeds\":" + (all ? "true" : "false") + ",\"tail_offset\":" + std::to_string(e.tail_offset(*x->graph)) + "}";
        }
        js += "]}";
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// get_tail_forest of extension number `which` of the set extend() returns for this cluster; JSON out:
// {"gap": n, "trees": [{"root_trim": n, "tree": [[parent, node id, is_reverse], ...]}, ...]}
int vgh_gapless_tail_forest(vgh_extender* x, const char* read, const int64_t* seeds, int n_seeds, int max_mismatches, int which, int left_tails, char* json_out, size_t json_cap) {
    try {
        GaplessExtender::cluster_type cluster;
        for (int i = 0; i < n_seeds; ++i) {
            Position pos; pos.node_id = seeds[4 * i]; pos.is_reverse = seeds[4 * i + 1] != 0; pos.offset = seeds[4 * i + 2];
            cluster.push_back(GaplessExtender::to_seed(*x->graph, pos, (size_t)seeds[4 * i + 3]));
        }
        const std::string sequence(read);
        auto result = x->ext->extend(cluster, sequence, (size_t)max_mismatches, GaplessExtender::OVERLAP_THRESHOLD, true);
        if (which < 0 || (size_t)which >= result.size()) { g_last_error = "no such extension"; return -3; }
        size_t gap = 0;
        auto forest = x->ext->get_tail_forest(result[(size_t)which], sequence.size(), left_tails != 0, &gap);
        std::string js = "{\"gap\":" + std::to_string(gap) + ",\"left_full\":" + (result[(size_t)which].left_full ? "true" : "false") + ",\"right_full\":" +
                         (result[(size_t)which].right_full ? "true" : "false") + ",\"trees\":[";
        for (size_t t = 0; t < forest.size(); ++t) {
            if (t) js += ',';
            js += "{\"root_trim\":" + std::to_string(forest[t].root_trim) + ",\"tree\":[";
            for (size_t k = 0; k < forest[t].tree.size(); ++k) {
                if (k) js += ',';
                js += "[" + std::to_string(forest[t].tree[k].first) + "," + std::to_string(x->graph->get_id(forest[t].tree[k].second)) + "," + (x->graph->get_is_reverse(forest[t].tree[k].second) ? "1" : "0") + "]";
            }
            js += "]}";
        }
        js += "]}";
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// ---- WFAExtender (src/gbwt_extender.hpp:346-461) -------------------------------------------------------------------------
struct vgh_wfa { std::unique_ptr<HaplotypeGraph> graph; WFAExtender::ErrorModel model; std::unique_ptr<WFAExtender> ext; std::shared_ptr<void> chain_out; };
// model: 4 x (per_base, min, max) for mismatches, gaps, gap_length, distance; nullptr = the default model
vgh_wfa* vgh_wfa_create(vgh_aligner* a, vgh_graph* g, const int64_t* thread_nodes, const int32_t* thread_off, int n_threads, const double* model) {
    try {
        std::vector<std::vector<handle_t>> threads((size_t)n_threads);
        for (int t = 0; t < n_threads; ++t) for (int32_t k = thread_off[t]; k < thread_off[t + 1]; ++k) threads[t].push_back(g->g.get_handle(thread_nodes[k] >> 1, thread_nodes[k] & 1));
        auto* h = new vgh_wfa();
        h->graph = std::make_unique<HaplotypeGraph>(g->g, threads);
        h->model = WFAExtender::default_error_model;
        if (model) {
            WFAExtender::ErrorModel::Event* ev[4] = { &h->model.mismatches, &h->model.gaps, &h->model.gap_length, &h->model.distance };
            for (int i = 0; i < 4; ++i) { ev[i]->per_base = model[3 * i]; ev[i]->min = (int32_t)model[3 * i + 1]; ev[i]->max = (int32_t)model[3 * i + 2]; }
        }
        h->ext = std::make_unique<WFAExtender>(*h->graph, *a->a, h->model);
        return h;
    } catch (std::exception& e) { g_last_error = e.what(); return nullptr; }
}
void vgh_wfa_destroy(vgh_wfa* w);       // (below: the chain stage's page-locked arrays go first)
// kind: 0 connect, 1 suffix, 2 prefix; from / to: [node id, is_reverse, offset]
int vgh_wfa_align(vgh_wfa* w, int kind, const char* seq, const int64_t* from, const int64_t* to, char* json_out, size_t json_cap) {
    try {
        Position f, t;
        if (from) { f.node_id = from[0]; f.is_reverse = from[1] != 0; f.offset = from[2]; }
        if (to) { t.node_id = to[0]; t.is_reverse = to[1] != 0; t.offset = to[2]; }
        const std::string sequence(seq);
        WFAAlignment r = kind == 0 ? w->ext->connect(sequence, f, t) : kind == 1 ? w->ext->suffix(sequence, f) : w->ext->prefix(sequence, t);
        std::string js = std::string("{\"ok\":") + (r ? "true" : "false") + ",\"score\":" + std::to_string(r.score) + ",\"node_offset\":" + std::to_string(r.node_offset) +
                         ",\"seq_offset\":" + std::to_string(r.seq_offset) + ",\"length\":" + std::to_string(r.length) + ",\"empty\":" + (r.empty() ? "true" : "false") +
                         ",\"unlocalized_insertion\":" + (r.unlocalized_insertion() ? "true" : "false") + ",\"path\":[";
        for (size_t i = 0; i < r.path.size(); ++i) js += std::string(i ? "," : "") + "[" + std::to_string(w->graph->get_id(r.path[i])) + "," + (w->graph->get_is_reverse(r.path[i]) ? "1" : "0") + "]";
        js += "],\"edits\":[";
        for (size_t i = 0; i < r.edits.size(); ++i) js += std::string(i ? "," : "") + "[" + std::to_string((int)r.edits[i].first) + "," + std::to_string(r.edits[i].second) + "]";
        js += "]";
        if (r) {
            js += ",\"final_offset\":" + std::to_string(r.final_offset(*w->graph));
            Alignment aln; aln.sequence = sequence; aln.path = r.to_path(*w->graph, sequence); aln.score = r.score;
            js += ",\"alignment\":" + alignment_to_json(aln);
        }
        js += "}";
        if (js.size() + 1 > json_cap) { g_last_error = "json buffer too small"; return -2; }
        std::memcpy(json_out, js.c_str(), js.size() + 1);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// The tails of a batch of gapless extensions (tail_stage.hpp) on an engine context and index the caller created through the C ABI of
// the same engine library.  ext_total: one per extension, read_score: one per read, stats: tails, trees, tree nodes, failed; ms[6].
int vgh_tail_stage(const char* engine_lib, void* ctx, const void* index, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                   const void* res, const void* ext, const uint32_t* nodes, const uint32_t* oriented_len, const int32_t scoring[4],
                   uint32_t ops_per_problem, int32_t* ext_total, uint64_t n_ext, int32_t* read_score, uint64_t stats[4], double ms[6]) {
    try {
        auto api = load_engine(engine_lib ? engine_lib : "");
        TailStageInput in{reads, read_off, n_reads, (const vgk_gapless_result*)res, (const vgk_extension*)ext, nodes, oriented_len,
                          scoring[0], scoring[1], scoring[2], scoring[3], ops_per_problem};
        TailStageOutput out;
        const int rc = run_tail_stage(*api, (vgk_ctx*)ctx, (const vgk_haplo*)index, in, out);
        if (rc) { g_last_error = api->strerror(rc); return rc; }
        if (out.ext_total.size() > n_ext) { g_last_error = "ext_total too small"; return VGK_EINVAL; }
        std::copy(out.ext_total.begin(), out.ext_total.end(), ext_total);
        std::copy(out.read_score.begin(), out.read_score.end(), read_score);
        if (stats) { stats[0] = out.n_tails; stats[1] = out.n_trees; stats[2] = out.tree_nodes; stats[3] = out.failed; }
        if (ms) for (int k = 0; k < 6; ++k) ms[k] = out.ms[k];
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

}  // extern "C"

// ---- the graph between / beyond anchors (chain_alignment.hpp; MinimizerMapper::align_sequence_between and friends) ------------------
#include "chain_alignment.hpp"
#include "cluster_alignment.hpp"
#include "rescue_fixups.hpp"
extern "C" {

// a bidirected graph, as the reference's tests build with HashGraph / json2graph
struct vgh_bigraph { LocalGraph g; };
vgh_bigraph* vgh_bigraph_create(void) { return new vgh_bigraph(); }
void vgh_bigraph_destroy(vgh_bigraph* g) { delete g; }
int vgh_bigraph_add_node(vgh_bigraph* g, int64_t id, const char* seq) {
    try { g->g.create_handle(seq, id); return 0; } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// vg's Edge message: from_start = the edge leaves the START of `from` (i.e. its reverse strand), to_end = it arrives at the END of `to`
int vgh_bigraph_add_edge(vgh_bigraph* g, int64_t from, int from_start, int64_t to, int to_end) {
    try { g->g.create_edge(g->g.get_handle(from, from_start != 0), g->g.get_handle(to, to_end != 0)); return 0; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

static Position as_position(const int64_t* p) { Position q; if (p) { q.node_id = p[0]; q.is_reverse = p[1] != 0; q.offset = p[2]; } return q; }
static int emit_string(const std::string& js, char* out, size_t cap) {
    if (js.size() + 1 > cap) { g_last_error = "json buffer too small"; return -2; }
    std::memcpy(out, js.c_str(), js.size() + 1);
    return 0;
}

// DozeuPinningOverlay (src/dozeu_pinning_overlay.hpp; the graph the X-drop aligner pins on: source / sink nodes without sequence are
// dropped and what they fed is duplicated) over a vgh_graph, as JSON for the tests that transcribe src/unittest/dozeu_pinning_overlay.cpp:
// {"performed_duplications": b, "node_count": n, "min_id": i, "max_id": i, "has_node": [ids in 1 .. 63 the overlay has],
//  "handles": [{"id", "sequence", "underlying": [id, is_reverse], "flip_underlying": [id, is_reverse], "flip_flip_is_self": b,
//               "right": [[id, rev]...], "left": [...], "flip_right": [...], "flip_left": [...]}]}   (follow_edges(h, false / true) of h and of flip(h))
int vgh_pinning_overlay(vgh_graph* g, int preserve_sinks, char* json_out, size_t json_cap) {
    try {
        DozeuPinningOverlay overlay(&g->g, preserve_sinks != 0);
        std::string js = "{\"performed_duplications\": ";
        js += overlay.performed_duplications() ? "true" : "false";
        js += ", \"node_count\": " + std::to_string(overlay.get_node_count()) + ", \"min_id\": " + std::to_string(overlay.min_node_id()) +
              ", \"max_id\": " + std::to_string(overlay.max_node_id()) + ", \"has_node\": [";
        bool firstn = true;
        for (nid_t i = 1; i < 64; ++i) if (overlay.has_node(i)) { js += (firstn ? "" : ", ") + std::to_string(i); firstn = false; }
        js += "], \"handles\": [";
        auto side = [&](const handle_t& h) { return "[" + std::to_string(overlay.get_id(h)) + ", " + (overlay.get_is_reverse(h) ? "true" : "false") + "]"; };
        auto under = [&](const handle_t& h) { const handle_t u = overlay.get_underlying_handle(h); return "[" + std::to_string(g->g.get_id(u)) + ", " + (g->g.get_is_reverse(u) ? "true" : "false") + "]"; };
        auto edges = [&](const handle_t& h, bool left) { std::string e = "["; bool f = true; overlay.follow_edges(h, left, [&](const handle_t& n) { e += (f ? "" : ", ") + side(n); f = false; return true; }); return e + "]"; };
        bool firsth = true;
        overlay.for_each_handle([&](const handle_t& h) {
            const handle_t r = overlay.flip(h);
            js += std::string(firsth ? "" : ", ") + "{\"id\": " + std::to_string(overlay.get_id(h)) + ", \"sequence\": \"" + overlay.get_sequence(h) + "\", \"underlying\": " + under(h) +
                  ", \"flip_underlying\": " + under(r) + ", \"flip_flip_is_self\": " + (overlay.flip(r) == h ? "true" : "false") +
                  ", \"right\": " + edges(h, false) + ", \"left\": " + edges(h, true) + ", \"flip_right\": " + edges(r, false) + ", \"flip_left\": " + edges(r, true) + "}";
            firsth = false;
            return true;
        });
        js += "]}";
        return emit_string(js, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// left / right: [node id, is_reverse, offset], node id 0 (or NULL) = no anchor on that side.  JSON out: {"did_align": b, "alignment": {...}}
// return 1: ChainAlignmentFailedError (message in vgh_last_error)
int vgh_align_sequence_between(vgh_aligner* a, vgh_bigraph* g, const char* read, const int64_t* left, const int64_t* right, int64_t max_path_length,
                               int64_t max_gap_length, int consistently, int64_t max_dp_cells, char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = read;
        const size_t cells = max_dp_cells < 0 ? std::numeric_limits<size_t>::max() : (size_t)max_dp_cells;
        const bool did = consistently ? align_sequence_between_consistently(as_position(left), as_position(right), (size_t)max_path_length, (size_t)max_gap_length, &g->g, a->a.get(), aln, nullptr, cells)
                                      : align_sequence_between(as_position(left), as_position(right), (size_t)max_path_length, (size_t)max_gap_length, &g->g, a->a.get(), aln, nullptr, cells);
        return emit_string(std::string("{\"did_align\":") + (did ? "true" : "false") + ",\"alignment\":" + alignment_to_json(aln) + "}", json_out, json_cap);
    } catch (ChainAlignmentFailedError& e) { g_last_error = e.what(); return 1; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// with_dagified_local_graph's view of one request.  JSON out: {"nodes": [[id, sequence, base id, base is_reverse], ...], "edges": [[from, to], ...],
// "tips": [[id, is_reverse], ...], "left_anchor": [id, is_reverse] | null, "right_anchor": ...}
int vgh_dagified_local_graph(vgh_bigraph* g, const int64_t* left, const int64_t* right, int64_t max_path_length, char* json_out, size_t json_cap) {
    try {
        std::string js;
        with_dagified_local_graph(as_position(left), as_position(right), (size_t)max_path_length, g->g,
            [&](LocalGraph& d, const handle_t& l, const handle_t& r, const std::function<std::pair<nid_t, bool>(const handle_t&)>& to_base) {
                js = "{\"nodes\":[";
                bool first = true;
                d.for_each_handle_v([&](const handle_t& h) {
                    const auto b = to_base(h);
                    js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(h)) + ",\"" + d.get_sequence(h) + "\"," + std::to_string(b.first) + "," + (b.second ? "1" : "0") + "]";
                    first = false;
                });
                js += "],\"edges\":[";
                first = true;
                d.for_each_handle_v([&](const handle_t& h) { d.follow_edges_v(h, false, [&](const handle_t& n) {
                    js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(h)) + "," + std::to_string(d.get_id(n)) + "]"; first = false; }); });
                js += "],\"tips\":[";
                first = true;
                for (const handle_t& t : handlealgs::find_tips(&d)) { js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(t)) + "," + (d.get_is_reverse(t) ? "1" : "0") + "]"; first = false; }
                auto anchor = [&](const int64_t* p, const handle_t& h) { return p && p[0] ? "[" + std::to_string(d.get_id(h)) + "," + (d.get_is_reverse(h) ? "1" : "0") + "]" : std::string("null"); };
                js += "],\"left_anchor\":" + anchor(left, l) + ",\"right_anchor\":" + anchor(right, r) + "}";
            });
        return emit_string(js, json_out, json_cap);
    } catch (ChainAlignmentFailedError& e) { g_last_error = e.what(); return 1; }
    catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// The graph algorithms of local_graph.hpp one by one, for the reference's own known-answer tests of them (src/unittest/dagify.cpp,
// src/unittest/vg_algorithms.cpp).  what / args:
//   "dagify" [min_preserved_path_length]                "dagify_from" [min_preserved_path_length, (node id, is_reverse) ...]
//   "split_strands" []                                  "properties" []   (-> {"acyclic": b, "single_stranded": b})
//   "extract_connecting" [max_len, id1, rev1, off1, id2, rev2, off2, strict_max_len]
//   "extract_extending"  [max_dist, id, rev, off, backward, preserve_cycles_on_src_node]
//   "extract_containing" [reversing_walk_length, n, (id, rev, off, forward length, backward length) x n]
// JSON out: {"nodes": [[id, sequence, source id, source is_reverse], ...], "edges": [[from, from_start, to, to_end], ...] (each edge once),
// "starts": [[id, is_reverse], ...], "acyclic": b, "single_stranded": b, "tips": [[id, is_reverse], ...]}
int vgh_graph_algorithm(vgh_bigraph* g, const char* what, const int64_t* args, int n_args, char* json_out, size_t json_cap) {
    try {
        const std::string op = what;
        LocalGraph out; const LocalGraph* shown = &out;
        std::unordered_map<nid_t, std::pair<nid_t, bool>> src;
        std::vector<handle_t> starts;
        auto need = [&](int k) { if (n_args < k) throw std::invalid_argument("vgh_graph_algorithm: too few arguments for " + op); };
        if (op == "dagify") { need(1); for (auto& kv : handlealgs::dagify(&g->g, &out, (size_t)args[0])) src[kv.first] = {kv.second, false}; }
        else if (op == "dagify_from") {
            need(3);
            std::vector<handle_t> from;
            for (int k = 1; k + 1 < n_args; k += 2) from.push_back(g->g.get_handle(args[k], args[k + 1] != 0));
            auto d = handlealgs::dagify_from(&g->g, from, &out, (size_t)args[0]);
            for (auto& kv : d.to_source) src[kv.first] = {kv.second, false};
            starts = d.starts;
        }
        else if (op == "split_strands") src = handlealgs::split_strands(&g->g, &out);
        else if (op == "properties") shown = &g->g;
        else if (op == "extract_connecting") {
            need(8);
            auto r = extract_connecting_graph(&g->g, &out, args[0], as_position(args + 1), as_position(args + 4), args[7] != 0);
            for (auto& kv : r.to_source) src[kv.first] = {kv.second, false};
        }
        else if (op == "extract_extending") {
            need(6);
            auto r = extract_extending_graph(&g->g, &out, args[0], as_position(args + 1), args[4] != 0, args[5] != 0);
            for (auto& kv : r.to_source) src[kv.first] = {kv.second, false};
        }
        else if (op == "extract_containing") {
            need(2);
            const int n = (int)args[1]; need(2 + 5 * n);
            std::vector<Position> pos; std::vector<size_t> fw, bw;
            for (int k = 0; k < n; ++k) { const int64_t* a = args + 2 + 5 * k; pos.push_back(as_position(a)); fw.push_back((size_t)a[3]); bw.push_back((size_t)a[4]); }
            extract_containing_graph(&g->g, &out, pos, fw, bw, (size_t)args[0]);
        }
        else throw std::invalid_argument("vgh_graph_algorithm: unknown algorithm " + op);
        const LocalGraph& d = *shown;
        std::string js = "{\"nodes\":[";
        bool first = true;
        d.for_each_handle_v([&](const handle_t& h) {
            auto found = src.find(d.get_id(h));
            const nid_t sid = found == src.end() ? d.get_id(h) : found->second.first; const bool srev = found != src.end() && found->second.second;
            js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(h)) + ",\"" + d.get_sequence(h) + "\"," + std::to_string(sid) + "," + (srev ? "1" : "0") + "]";
            first = false;
        });
        js += "],\"edges\":[";
        first = true;
        std::set<std::array<int64_t, 4>> edges;                                        // (from, from_start, to, to_end), the smaller of an edge's two readings
        d.for_each_handle_v([&](const handle_t& fwd) {
            for (int rev = 0; rev < 2; ++rev) {
                const handle_t h = rev ? d.flip(fwd) : fwd;
                d.follow_edges_v(h, false, [&](const handle_t& n) {
                    const std::array<int64_t, 4> a{d.get_id(h), d.get_is_reverse(h) ? 1 : 0, d.get_id(n), d.get_is_reverse(n) ? 1 : 0};
                    const std::array<int64_t, 4> b{a[2], a[3] ? 0 : 1, a[0], a[1] ? 0 : 1};
                    edges.insert(std::min(a, b));
                });
            }
        });
        for (const auto& e : edges) { js += std::string(first ? "" : ",") + "[" + std::to_string(e[0]) + "," + std::to_string(e[1]) + "," + std::to_string(e[2]) + "," + std::to_string(e[3]) + "]"; first = false; }
        js += "],\"starts\":[";
        first = true;
        for (const handle_t& t : starts) { js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(t)) + "," + (d.get_is_reverse(t) ? "1" : "0") + "]"; first = false; }
        js += "],\"tips\":[";
        first = true;
        for (const handle_t& t : handlealgs::find_tips(&d)) { js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(t)) + "," + (d.get_is_reverse(t) ? "1" : "0") + "]"; first = false; }
        js += std::string("],\"acyclic\":") + (handlealgs::is_acyclic(&d) ? "true" : "false") + ",\"single_stranded\":" + (handlealgs::is_single_stranded(&d) ? "true" : "false") + "}";
        return emit_string(js, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

// Mapper::align_to_graph (cluster_alignment.hpp).  flags: 1 do_flip, 2 traceback, 4 pinned, 8 pin_left, 16 banded_global, 32 keep_bonuses.
// JSON out: the alignment, positions on the caller's own nodes.
int vgh_align_to_graph(vgh_aligner* a, vgh_bigraph* g, const char* read, int flags, char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = read;
        const Alignment out = align_to_graph(aln, g->g, *a->a, flags & 1, flags & 2, flags & 4, flags & 8, flags & 16, flags & 32);
        return emit_string(alignment_to_json(out), json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// cluster_subgraph_containing: seeds = n x (read begin, read end, node id, is_reverse, offset).  JSON out: like vgh_graph_algorithm.
int vgh_cluster_subgraph(vgh_aligner* a, vgh_bigraph* g, int64_t read_length, const int64_t* seeds, int n, char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence.assign((size_t)read_length, 'A');
        std::vector<ClusterSeed> cluster;
        for (int k = 0; k < n; ++k) { ClusterSeed s; s.begin = (size_t)seeds[5 * k]; s.end = (size_t)seeds[5 * k + 1]; s.start = as_position(seeds + 5 * k + 2); cluster.push_back(s); }
        const LocalGraph d = cluster_subgraph_containing(g->g, aln, cluster, *a->a);
        std::string js = "{\"nodes\":[";
        bool first = true;
        d.for_each_handle_v([&](const handle_t& h) { js += std::string(first ? "" : ",") + "[" + std::to_string(d.get_id(h)) + ",\"" + d.get_sequence(h) + "\"]"; first = false; });
        js += "]}";
        return emit_string(js, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

int64_t vgh_longest_detectable_gap_in_range(vgh_aligner* a, int64_t read_length, int64_t begin, int64_t end) {
    Alignment aln; aln.sequence.assign((size_t)read_length, 'A');
    return (int64_t)longest_detectable_gap_in_range(aln, (size_t)begin, (size_t)end, a->a.get());
}

// ChainConnector: the same requests, many per engine flush
struct vgh_connector { std::unique_ptr<ChainConnector> c; std::deque<Alignment> alns; };
vgh_connector* vgh_connector_create(vgh_aligner* a, vgh_bigraph* g, int64_t max_dp_cells) {
    auto* h = new vgh_connector();
    h->c = std::make_unique<ChainConnector>(*a->a, g->g, max_dp_cells < 0 ? std::numeric_limits<size_t>::max() : (size_t)max_dp_cells);
    return h;
}
void vgh_connector_destroy(vgh_connector* c) { delete c; }
int vgh_connector_add(vgh_connector* c, const char* read, const int64_t* left, const int64_t* right, int64_t max_path_length, int64_t max_gap_length) {
    c->alns.emplace_back(); c->alns.back().sequence = read;
    return (int)c->c->add(as_position(left), as_position(right), (size_t)max_path_length, (size_t)max_gap_length, c->alns.back());
}
// JSON out: [{"status": n, "did_align": b, "message": "...", "alignment": {...}}, ...] for every request so far; ms[3]: extraction, engine flush, translation
int vgh_connector_run(vgh_connector* c, int threads, double* ms, char* json_out, size_t json_cap) {
    try {
        c->c->run((unsigned)std::max(threads, 0));
        if (ms) { ms[0] = c->c->last_extract_ms; ms[1] = c->c->last_align_ms; ms[2] = c->c->last_translate_ms; }
        if (!json_out) return 0;
        std::string js = "[";
        for (size_t i = 0; i < c->alns.size(); ++i) {
            const ChainConnector::Outcome& o = c->c->outcome(i);
            std::string msg; for (char ch : o.message) if (ch != '"' && ch != '\\' && ch != '\n') msg += ch;
            js += std::string(i ? "," : "") + "{\"status\":" + std::to_string((int)o.status) + ",\"did_align\":" + (o.did_align ? "true" : "false") + ",\"message\":\"" + msg +
                  "\",\"alignment\":" + alignment_to_json(c->alns[i]) + "}";
        }
        js += "]";
        return emit_string(js, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}

}  // extern "C"

// ---- the chain stage (chain_stage.hpp): every link of a batch of reads through WFA, the declined ones through align_sequence_between ----
#include "chain_stage.hpp"
#include <atomic>
#include <thread>
#include <mutex>
extern "C" {
// w: the WFA handle (haplotype graph + index + aligner) of vgh_wfa_create.  stats: declined, between, no graph, too big, failed, broken reads; ms[6].
// anchors (anchor_off ... anchor_nodes, all or none; chain_stage.hpp): with them one alignment per read is composed (vgk_chain_stitch) and stays with
// the handle until the next call: sizes = {mappings, edits}; vgh_chain_stage_view hands out the arrays (no copy).
int vgh_chain_stage(vgh_wfa* w, const char* seqs, const uint64_t* seq_off, uint32_t n_links, const uint32_t* mode, const uint32_t* from_node, const uint32_t* from_offset,
                    const uint32_t* to_node, const uint32_t* to_offset, const uint32_t* read_of, uint32_t n_reads, const uint32_t* graph_distance,
                    const uint32_t* read_begin, const uint32_t* read_length, const int64_t* anchor_score, int threads, int dp_for_tails,
                    int32_t* link_score, uint8_t* link_source, int32_t* wfa_status, int64_t* chain_score, uint64_t stats[6], double ms[6],
                    const uint64_t* anchor_off, const uint32_t* anchor_length, const uint32_t* anchor_node_offset, const uint64_t* anchor_path_off, const uint32_t* anchor_nodes,
                    uint64_t sizes[2]) {
    try {
        ChainStageInput in{};
        in.seqs = seqs; in.seq_off = seq_off; in.n_links = n_links; in.mode = mode; in.from_node = from_node; in.from_offset = from_offset;
        in.to_node = to_node; in.to_offset = to_offset; in.read_of = read_of; in.n_reads = n_reads; in.graph_distance = graph_distance;
        in.read_begin = read_begin; in.read_length = read_length; in.anchor_score = anchor_score; in.threads = (unsigned)std::max(threads, 0);
        in.dp_for_tails = dp_for_tails != 0;
        if (anchor_off) {
            if (!anchor_length || !anchor_node_offset || !anchor_path_off || !anchor_nodes) { g_last_error = "vgh_chain_stage: anchors need all five arrays"; return -1; }
            in.anchor_off = anchor_off; in.anchor_length = anchor_length; in.anchor_node_offset = anchor_node_offset; in.anchor_path_off = anchor_path_off; in.anchor_nodes = anchor_nodes;
        }
        if (!w->chain_out) w->chain_out = std::make_shared<ChainStageOutput>();       // (its arrays keep their room from batch to batch)
        ChainStageOutput& out = *static_cast<ChainStageOutput*>(w->chain_out.get());
        const Aligner& aligner = *w->ext->aligner;
        const int rc = run_chain_stage(aligner.engine_api(), aligner.engine_context(), w->ext->engine_index(), *w->graph, aligner, nullptr, in, out);
        if (rc) { g_last_error = aligner.engine_api().strerror(rc); return rc; }
        std::copy(out.link_score.begin(), out.link_score.end(), link_score);
        std::copy(out.link_source.begin(), out.link_source.end(), link_source);
        if (wfa_status) std::copy(out.wfa_status.begin(), out.wfa_status.end(), wfa_status);
        std::copy(out.chain_score.begin(), out.chain_score.end(), chain_score);
        if (stats) { stats[0] = out.n_declined; stats[1] = out.n_between; stats[2] = out.n_no_graph; stats[3] = out.n_too_big; stats[4] = out.n_failed; stats[5] = out.n_broken; }
        if (ms) for (int k = 0; k < 6; ++k) ms[k] = out.ms[k];
        if (sizes) { sizes[0] = anchor_off ? out.n_mappings : 0; sizes[1] = anchor_off ? out.n_edits : 0; }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
void vgh_wfa_destroy(vgh_wfa* w) {
    if (w && w->chain_out && w->ext) { const Aligner& aligner = *w->ext->aligner; static_cast<ChainStageOutput*>(w->chain_out.get())->release(aligner.engine_api(), aligner.engine_context()); }
    delete w;
}
// page-lock / release a caller's buffer through the WFA handle's engine context (a batch's sequence arena: it then goes up by DMA straight from the caller's pages)
int vgh_wfa_host_register(vgh_wfa* w, const void* ptr, uint64_t bytes, int on) {
    try {
        const Aligner& aligner = *w->ext->aligner; const EngineApi& api = aligner.engine_api();
        if (!api.host_register || !api.host_unregister) return 0;
        const int rc = on ? api.host_register(aligner.engine_context(), ptr, (size_t)bytes) : api.host_unregister(aligner.engine_context(), ptr);
        if (rc) { g_last_error = api.strerror(rc); return rc; }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// the composed alignments of the last vgh_chain_stage call with anchors: per read vgk_chain_result, the mappings, the edit runs, per read 1 = chain broken; valid until the next call
int vgh_chain_stage_view(vgh_wfa* w, const void** read_result, const void** mappings, const void** edits, const uint8_t** read_broken, double* stitch_kernel_ms) {
    if (!w || !w->chain_out) { g_last_error = "vgh_chain_stage_view: no chain stage has run on this handle"; return -1; }
    ChainStageOutput& out = *static_cast<ChainStageOutput*>(w->chain_out.get());
    if (read_result) *read_result = out.read_result.data();
    if (mappings) *mappings = out.mappings.data();
    if (edits) *edits = out.edits.data();
    if (read_broken) *read_broken = out.read_broken.data();
    if (stitch_kernel_ms) *stitch_kernel_ms = out.stitch_kernel_ms;
    return 0;
}

// algorithms::sample_minimal with should_beat(a, b) = goodness[a] > goodness[b]: sampled[i] = 1 for every element sampled
int vgh_sample_minimal(uint64_t count, uint64_t element_length, uint64_t window_size, uint64_t sequence_length, const uint64_t* starts, const int64_t* goodness, uint8_t* sampled) {
    try {
        for (uint64_t i = 0; i < count; ++i) sampled[i] = 0;
        sample_minimal((size_t)count, (size_t)element_length, (size_t)window_size, (size_t)sequence_length,
                       [&](size_t i) { return (size_t)starts[i]; }, [&](size_t a, size_t b) { return goodness[a] > goodness[b]; }, [&](size_t i) { sampled[i] = 1; });
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// MinimizerMapper::find_seeds' selection: minimizers in read order, 4 numbers each {key, forward offset, length, hits}; policy = {hit_cap,
// hard_hit_cap, max_unique_min, num_bp_per_min, exclude_overlapping_min, minimizer_coverage_flank, downsampling_window_count,
// downsampling_max_window_length}; verdict_out[i] = SeedFilter (0 = its hits become seeds); scores_out (nullable) = find_minimizers' scores
static int select_minimizers_impl(const uint64_t* minimizers, int n, uint64_t read_length, const uint64_t* policy, double score_fraction, uint8_t* verdict_out, double* scores_out,
                                  const std::string* sequence, uint64_t* order_out);
int vgh_select_minimizers(const uint64_t* minimizers, int n, uint64_t read_length, const uint64_t* policy, double score_fraction, uint8_t* verdict_out, double* scores_out) {
    return select_minimizers_impl(minimizers, n, read_length, policy, score_fraction, verdict_out, scores_out, nullptr, nullptr);
}
// the same with the read's sequence: the runs tied at the best score are shuffled as the reference shuffles them (sort_shuffling_ties); order_out (nullable, n):
// the minimizers' indices in the order the filters took them
int vgh_select_minimizers_of_read(const uint64_t* minimizers, int n, const char* sequence, uint64_t read_length, const uint64_t* policy, double score_fraction,
                                  uint8_t* verdict_out, double* scores_out, uint64_t* order_out) {
    const std::string seq(sequence, (size_t)read_length);
    return select_minimizers_impl(minimizers, n, read_length, policy, score_fraction, verdict_out, scores_out, &seq, order_out);
}
// the paired path (src/minimizer_mapper.cpp:1529-1541): ONE generator seeded from mate 1 + mate 2 orders mate 1's minimizers, then mate 2's.
// minimizers1 / minimizers2 as above; order1_out / order2_out (n1 / n2): the orders the filters took them in; verdict1_out / verdict2_out: SeedFilter per minimizer
int vgh_select_minimizers_of_pair(const uint64_t* minimizers1, int n1, const char* sequence1, uint64_t length1, const uint64_t* minimizers2, int n2, const char* sequence2, uint64_t length2,
                                  const uint64_t* policy, double score_fraction, uint8_t* verdict1_out, uint8_t* verdict2_out, uint64_t* order1_out, uint64_t* order2_out) {
    try {
        SeedPolicy P; P.hit_cap = (size_t)policy[0]; P.hard_hit_cap = (size_t)policy[1]; P.max_unique_min = (size_t)policy[2]; P.num_bp_per_min = (size_t)policy[3];
        P.exclude_overlapping_min = policy[4] != 0; P.minimizer_coverage_flank = (size_t)policy[5]; P.minimizer_downsampling_window_count = (size_t)policy[6];
        P.minimizer_downsampling_max_window_length = (size_t)policy[7]; P.minimizer_score_fraction = score_fraction;
        ReadRng rng(std::string(sequence1, (size_t)length1) + std::string(sequence2, (size_t)length2));
        for (int mate = 0; mate < 2; ++mate) {
            const uint64_t* in = mate ? minimizers2 : minimizers1; const int n = mate ? n2 : n1;
            std::vector<PolicyMinimizer> ms((size_t)n);
            for (int i = 0; i < n; ++i) { const uint64_t* q = in + 4 * (size_t)i; ms[(size_t)i].key = q[0]; ms[(size_t)i].forward_offset = (size_t)q[1]; ms[(size_t)i].length = (size_t)q[2]; ms[(size_t)i].hits = (size_t)q[3]; }
            score_minimizers(ms, P.hard_hit_cap);
            // (the order is drawn once per mate — sort_minimizers_by_score is called once per mate — and the filters run over that order)
            const std::vector<size_t> order = minimizers_by_score(ms, rng);
            const std::vector<uint8_t> v = select_minimizers_in_order(ms, (size_t)(mate ? length2 : length1), P, order);
            for (int i = 0; i < n; ++i) { (mate ? verdict2_out : verdict1_out)[i] = v[(size_t)i]; if (mate ? order2_out : order1_out) (mate ? order2_out : order1_out)[i] = order[(size_t)i]; }
        }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
static int select_minimizers_impl(const uint64_t* minimizers, int n, uint64_t read_length, const uint64_t* policy, double score_fraction, uint8_t* verdict_out, double* scores_out,
                                  const std::string* sequence, uint64_t* order_out) {
    try {
        std::vector<PolicyMinimizer> ms((size_t)n);
        for (int i = 0; i < n; ++i) { const uint64_t* q = minimizers + 4 * (size_t)i; ms[(size_t)i].key = q[0]; ms[(size_t)i].forward_offset = (size_t)q[1]; ms[(size_t)i].length = (size_t)q[2]; ms[(size_t)i].hits = (size_t)q[3]; }
        SeedPolicy P; P.hit_cap = (size_t)policy[0]; P.hard_hit_cap = (size_t)policy[1]; P.max_unique_min = (size_t)policy[2]; P.num_bp_per_min = (size_t)policy[3];
        P.exclude_overlapping_min = policy[4] != 0; P.minimizer_coverage_flank = (size_t)policy[5]; P.minimizer_downsampling_window_count = (size_t)policy[6];
        P.minimizer_downsampling_max_window_length = (size_t)policy[7]; P.minimizer_score_fraction = score_fraction;
        score_minimizers(ms, P.hard_hit_cap);
        const std::vector<uint8_t> v = select_minimizers(ms, (size_t)read_length, P, sequence);
        for (int i = 0; i < n; ++i) { verdict_out[i] = v[(size_t)i]; if (scores_out) scores_out[i] = ms[(size_t)i].score; }
        if (order_out) { const std::vector<size_t> order = minimizers_by_score(ms, sequence); for (int i = 0; i < n; ++i) order_out[i] = order[(size_t)i]; }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// MinimizerMapper::score_cluster: minimizers as for vgh_select_minimizers (4 numbers each; scores from find_minimizers' rule with policy[1] = hard_hit_cap);
// seed_sources[n_seeds]: the minimizer each seed of the cluster came from -> out[0] = score, out[1] = coverage; present_out (nullable, n): 1 per minimizer present
int vgh_score_cluster(const uint64_t* minimizers, int n, uint64_t hard_hit_cap, const uint64_t* seed_sources, int n_seeds, uint64_t seq_length, double out[2], uint8_t* present_out) {
    try {
        std::vector<PolicyMinimizer> ms((size_t)n);
        for (int i = 0; i < n; ++i) { const uint64_t* q = minimizers + 4 * (size_t)i; ms[(size_t)i].key = q[0]; ms[(size_t)i].forward_offset = (size_t)q[1]; ms[(size_t)i].length = (size_t)q[2]; ms[(size_t)i].hits = (size_t)q[3]; }
        score_minimizers(ms, (size_t)hard_hit_cap);
        const ClusterScore c = score_cluster(std::vector<size_t>(seed_sources, seed_sources + n_seeds), ms, (size_t)seq_length);
        out[0] = c.score; out[1] = c.coverage;
        if (present_out) for (int i = 0; i < n; ++i) present_out[i] = c.present[(size_t)i];
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// find_seeds' choice for a BATCH of reads of any length, over what vgk_minimizer_list answered (include/vgk.h: vgk_read_minimizer records behind each
// other, read r = [minimizer_off[r], minimizer_off[r + 1])): take_out[j] = 1 where minimizer j's hits become seeds (SeedFilter 0).  k: the index's k-mer
// length (a minimizer covers read bases [offset, offset + k)).  Reads on `threads` host threads (0 = all).  The reads' own bytes seed each read's shuffle.
int vgh_select_minimizers_of_reads(const vgk_read_minimizer* minimizers, const uint64_t* minimizer_off, uint32_t n_reads, const char* reads, const uint64_t* read_off, uint32_t k,
                                   const uint64_t* policy, double score_fraction, int threads, uint8_t* take_out) {
    try {
        SeedPolicy P; P.hit_cap = (size_t)policy[0]; P.hard_hit_cap = (size_t)policy[1]; P.max_unique_min = (size_t)policy[2]; P.num_bp_per_min = (size_t)policy[3];
        P.exclude_overlapping_min = policy[4] != 0; P.minimizer_coverage_flank = (size_t)policy[5]; P.minimizer_downsampling_window_count = (size_t)policy[6];
        P.minimizer_downsampling_max_window_length = (size_t)policy[7]; P.minimizer_score_fraction = score_fraction;
        unsigned T = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
        T = std::min<unsigned>(T, std::max<uint32_t>(1, n_reads));
        std::atomic<uint32_t> next{0}; std::atomic<bool> failed{false}; std::string what;
        std::mutex what_mu;
        auto work = [&]() {
            try {
                std::vector<PolicyMinimizer> ms;
                for (uint32_t r = next.fetch_add(1); r < n_reads; r = next.fetch_add(1)) {
                    const uint64_t a = minimizer_off[r], b = minimizer_off[r + 1];
                    ms.assign((size_t)(b - a), PolicyMinimizer());
                    for (uint64_t j = a; j < b; ++j) { PolicyMinimizer& m = ms[(size_t)(j - a)]; m.key = minimizers[j].key; m.forward_offset = minimizers[j].offset; m.length = k; m.hits = minimizers[j].hits; }
                    score_minimizers(ms, P.hard_hit_cap);
                    const std::string seq(reads + read_off[r], (size_t)(read_off[r + 1] - read_off[r]));
                    const std::vector<uint8_t> v = select_minimizers(ms, seq.size(), P, &seq);
                    for (uint64_t j = a; j < b; ++j) take_out[j] = v[(size_t)(j - a)] == 0 ? 1 : 0;
                }
            } catch (std::exception& e) { failed = true; std::lock_guard<std::mutex> lk(what_mu); what = e.what(); }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
        if (failed) { g_last_error = what; return -1; }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
}  // extern "C"
extern "C" {
// vgk_wfa_set_point_budgets on the context behind this WFA handle (0, 0 = no budget: the reference has none)
int vgh_wfa_set_point_budgets(vgh_wfa* w, uint32_t connect_points, uint32_t tail_points) {
    const Aligner& a = *w->ext->aligner;
    return a.engine_api().wfa_set_point_budgets(a.engine_context(), connect_points, tail_points);
}
double vgh_wfa_last_kernel_ms(vgh_wfa* w) { const Aligner& a = *w->ext->aligner; return a.engine_api().wfa_last_ms(a.engine_context()); }
double vgh_wfa_last_wave(vgh_wfa* w, int which) { const Aligner& a = *w->ext->aligner; return a.engine_api().wfa_last_wave(a.engine_context(), which); }
}
extern "C" {
// MappingQualityCalculator::maximum_mapping_quality_exact / _approx (static): -> the mapping quality; *max_idx = the chosen element
double vgh_maximum_mapping_quality(const double* scaled_scores, int n, int approx, int64_t* max_idx) {
    std::vector<double> s(scaled_scores, scaled_scores + n);
    size_t idx = 0;
    const double q = approx ? MappingQualityCalculator::maximum_mapping_quality_approx(s, &idx) : MappingQualityCalculator::maximum_mapping_quality_exact(s, &idx);
    if (max_idx) *max_idx = (int64_t)idx;
    return q;
}
// the aligner's mapq_calc member over raw scores (scaled by the recovered log base); first: the first score instead of the best
int32_t vgh_compute_mapping_quality(vgh_aligner* a, const double* scores, int n, int fast_approximation, int first) {
    std::vector<double> s(scores, scores + n);
    return first ? a->a->mapq_calc->compute_first_mapping_quality(s, fast_approximation != 0) : a->a->mapq_calc->compute_max_mapping_quality(s, fast_approximation != 0);
}
double vgh_log_base(vgh_aligner* a) { return a->a->scorer->get_log_base(); }
// The class shapes of aligner_client.hpp, exercised the way a mapper uses them: an AlignerClient whose scores come from a matrix stream,
// its regular / quality-adjusted aligner picked by get_aligner; an XdropAligner (or its quality-adjusted twin) given the bonus per call.
//   what = 0: AlignerClient(matrix text).get_aligner(have qualities)->align_pinned(xdrop = false)
//   what = 1: XdropAligner(matrix from the client's parse).align_pinned(..., full_length_bonus, max_gap)
//   what = 2: QualAdjXdropAligner likewise
int vgh_client_align_pinned(const char* engine_lib, const char* matrix_text, int gap_open, int gap_extend, int full_length_bonus, int adjust_for_quality, int what,
                            vgh_graph* g, const char* read, const uint8_t* qual, int pin_left, int max_gap, char* json_out, size_t json_cap) {
    try {
        auto eng = load_engine(engine_lib ? engine_lib : "");
        Alignment aln; aln.sequence = read;
        if (qual) aln.quality.assign(reinterpret_cast<const char*>(qual), aln.sequence.size());
        std::istringstream in(matrix_text);
        if (what == 0) {
            AlignerClient client(0.5, eng, 0);
            client.adjust_alignments_for_base_quality = adjust_for_quality != 0;
            client.set_alignment_scores(in, (int8_t)gap_open, (int8_t)gap_extend, (int8_t)full_length_bonus);
            client.get_aligner(qual != nullptr)->align_pinned(aln, g->g, pin_left != 0);
        } else {
            const std::vector<int8_t> m = AlignerClient::parse_matrix(in);
            if (what == 1) { XdropAligner x(m.data(), (int8_t)gap_open, (int8_t)gap_extend, eng, 0); x.align_pinned(aln, g->g, pin_left != 0, (int8_t)full_length_bonus, (uint16_t)max_gap); }
            else { QualAdjXdropAligner x(m.data(), (int8_t)gap_open, (int8_t)gap_extend, 0.5, eng, 0); x.align_pinned(aln, g->g, pin_left != 0, (int8_t)full_length_bonus, (uint16_t)max_gap); }
        }
        return emit(aln, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// MinimizerMapper::score_extension_group: intervals[3 k ..] = {read begin, read end, score} of extension k, in the extender's order
int vgh_score_extension_group(uint64_t read_length, const int64_t* intervals, int n, int full_length, int gap_open, int gap_extend) {
    std::vector<ScoredInterval> v((size_t)n);
    for (int k = 0; k < n; ++k) { v[(size_t)k].begin = (size_t)intervals[3 * k]; v[(size_t)k].end = (size_t)intervals[3 * k + 1]; v[(size_t)k].score = (int32_t)intervals[3 * k + 2]; }
    return score_extension_group((size_t)read_length, v, full_length != 0, gap_open, gap_extend);
}
// MinimizerMapper::faster_cap: minimizers flat, 6 numbers each {hash, offset, is_reverse, agglomeration_start, agglomeration_length, length};
// explored = indices into them; quality = raw Phred bytes (n_quality 0: none).  *cap_out = the cap (inf without qualities); rc -1 where the
// reference prints an error and exits.
int vgh_faster_cap(const uint64_t* minimizers, int n_minimizers, const uint64_t* explored, int n_explored, const char* sequence, const unsigned char* quality, int n_quality, double* cap_out) {
    try {
        std::vector<CapMinimizer> ms((size_t)n_minimizers);
        for (int i = 0; i < n_minimizers; ++i) {
            CapMinimizer& m = ms[(size_t)i]; const uint64_t* q = minimizers + 6 * (size_t)i;
            m.hash = q[0]; m.offset = (size_t)q[1]; m.is_reverse = q[2] != 0; m.agglomeration_start = (size_t)q[3]; m.agglomeration_length = (size_t)q[4]; m.length = (int32_t)q[5];
        }
        std::vector<size_t> ex(explored, explored + n_explored);
        for (size_t e : ex) if (e >= ms.size()) throw std::runtime_error("faster_cap: explored minimizer out of range");
        *cap_out = faster_cap(ms, ex, sequence, std::string((const char*)quality, (size_t)n_quality));
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// MinimizerMapper::fix_dozeu_end_deletions over an alignment given flat: positions[m] = {node id, offset, is_reverse} per mapping,
// edits[k] = {mapping index, from_length, to_length, has sequence}; JSON out = the alignment afterwards
static int fix_end_deletions_flat(const char* sequence, const int64_t* positions, int n_mappings, const int64_t* edits, int n_edits, bool reference_indexing, char* json_out, size_t json_cap);
int vgh_fix_dozeu_end_deletions(const char* sequence, const int64_t* positions, int n_mappings, const int64_t* edits, int n_edits, char* json_out, size_t json_cap) {
    return fix_end_deletions_flat(sequence, positions, n_mappings, edits, n_edits, false, json_out, json_cap);
}
// the same with the reference's own indexing of the mappings (rescue_fixups.hpp)
int vgh_fix_dozeu_end_deletions_as_written(const char* sequence, const int64_t* positions, int n_mappings, const int64_t* edits, int n_edits, char* json_out, size_t json_cap) {
    return fix_end_deletions_flat(sequence, positions, n_mappings, edits, n_edits, true, json_out, json_cap);
}
static int fix_end_deletions_flat(const char* sequence, const int64_t* positions, int n_mappings, const int64_t* edits, int n_edits, bool reference_indexing, char* json_out, size_t json_cap) {
    try {
        Alignment aln; aln.sequence = sequence;
        aln.path.mapping.resize((size_t)n_mappings);
        for (int m = 0; m < n_mappings; ++m) { Position& p = aln.path.mapping[(size_t)m].position; p.node_id = positions[3 * m]; p.offset = positions[3 * m + 1]; p.is_reverse = positions[3 * m + 2] != 0; }
        size_t at = 0;
        for (int k = 0; k < n_edits; ++k) {
            Edit e; e.from_length = (int32_t)edits[4 * k + 1]; e.to_length = (int32_t)edits[4 * k + 2];
            if (edits[4 * k + 3]) e.sequence = aln.sequence.substr(at, (size_t)e.to_length);
            at += (size_t)e.to_length;
            aln.path.mapping.at((size_t)edits[4 * k]).edit.push_back(e);
        }
        fix_dozeu_end_deletions(aln, reference_indexing);
        return emit(aln, json_out, json_cap);
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
// scorer->score_contiguous_alignment of an alignment given as flat mappings: edits[k] = {mapping index, from_length, to_length, has sequence}
int32_t vgh_score_contiguous_alignment(vgh_aligner* a, const char* sequence, const int64_t* edits, int n_edits) {
    Alignment aln; aln.sequence = sequence;
    size_t at = 0;
    for (int k = 0; k < n_edits; ++k) {
        const size_t m = (size_t)edits[4 * k];
        while (aln.path.mapping.size() <= m) aln.path.mapping.emplace_back();
        Edit e; e.from_length = (int32_t)edits[4 * k + 1]; e.to_length = (int32_t)edits[4 * k + 2];
        if (edits[4 * k + 3]) e.sequence = aln.sequence.substr(at, (size_t)e.to_length);
        at += (size_t)e.to_length;
        aln.path.mapping[m].edit.push_back(e);
    }
    return a->a->scorer->score_contiguous_alignment(aln);
}
}
extern "C" {
// AlignmentBatch::align_xdrop: slot as in vgh_batch_add_slot; MEMs flat as in vgh_align_xdrop
int vgh_batch_add_xdrop_slot(vgh_batch* b, int slot, vgh_graph* g, const char* read, const int64_t* mems, int n_mems, int reverse_complemented, int max_gap) {
    try {
        Alignment& aln = b->alns[(size_t)slot]; aln = Alignment(); aln.sequence = read;
        std::vector<MaximalExactMatch> ms;
        for (int i = 0; i < n_mems; ++i) {
            MaximalExactMatch m; m.begin = (size_t)mems[5 * i]; m.end = (size_t)mems[5 * i + 1];
            m.nodes.push_back({mems[5 * i + 2], (size_t)mems[5 * i + 3], mems[5 * i + 4] != 0});
            ms.push_back(m);
        }
        b->b->align_xdrop(aln, g->g, ms, reverse_complemented != 0, (uint16_t)max_gap);
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
}
extern "C" {
// Submissions from many threads at once: the caller reserves n slots, every thread submits into slots of its own (no lock on the
// caller's side: AlignmentBatch's own lock orders the submissions), the flush answers in slot order.
int vgh_batch_reserve(vgh_batch* b, int n) { b->alns.clear(); b->alns.resize((size_t)n); return 0; }
int vgh_batch_add_slot(vgh_batch* b, int slot, vgh_graph* g, const char* read, int call, int pin_left, int arg) {
    try {
        Alignment& aln = b->alns[(size_t)slot]; aln = Alignment(); aln.sequence = read;
        switch (call) {
            case 0: b->b->align(aln, g->g, true); break;
            case 2: b->b->align_pinned(aln, g->g, pin_left != 0); break;
            case 4: b->b->align_pinned(aln, g->g, pin_left != 0, true, (uint16_t)arg); break;
            case 5: b->b->align_global_banded(aln, g->g, arg, pin_left != 0); break;
            default: g_last_error = "unknown call"; return -1;
        }
        return 0;
    } catch (std::exception& e) { g_last_error = e.what(); return -1; }
}
}
