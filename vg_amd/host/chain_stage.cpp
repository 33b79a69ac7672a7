// chain_stage.cpp — see chain_stage.hpp.
#include "chain_stage.hpp"
#include <algorithm>
#include <chrono>
#include <deque>

namespace vgamd {

void ChainStageOutput::release(const EngineApi& api, vgk_ctx* ctx) {
    if (api.host_unregister) { if (pinned_m) api.host_unregister(ctx, pinned_m); if (pinned_e) api.host_unregister(ctx, pinned_e); }
    pinned_m = pinned_e = nullptr;
}

int run_chain_stage(const EngineApi& api, vgk_ctx* ctx, const vgk_haplo* index, const HaplotypeGraph& graph, const Aligner& aligner,
                    const vgk_wfa_error_model* model, const ChainStageInput& in, ChainStageOutput& out) {
    using clock = std::chrono::steady_clock;
    auto t0 = clock::now();
    auto lap = [&](int k) { const auto t = clock::now(); out.ms[k] += std::chrono::duration<double, std::milli>(t - t0).count(); t0 = t; };
    const uint32_t n = in.n_links;
    out.link_score.assign(n, 0); out.link_source.assign(n, ChainStageOutput::NONE); out.wfa_status.assign(n, VGK_OK);
    out.chain_score.assign(in.n_reads, 0);
    out.n_declined = out.n_between = out.n_no_graph = out.n_too_big = out.n_failed = 0;
    for (double& m : out.ms) m = 0;
    out.read_result.clear(); out.read_broken.clear(); out.n_broken = 0; out.stitch_kernel_ms = 0;
    // 1. everything through WFAExtender, one engine call.  A read's links are long and a percent of them carry a long gap: the wavefront kernel
    //    throughout, so that the heavy ones start at once (include/vgk.h, vgk_wfa_set_form)
    //    The aligner's context is shared: the form in force comes back after the call.
    const int form_before = api.wfa_get_form ? api.wfa_get_form(ctx) : VGK_WFA_FORM_HYBRID;
    struct RestoreForm { const EngineApi& api; vgk_ctx* ctx; int form; ~RestoreForm() { if (api.wfa_set_form && form >= 0) api.wfa_set_form(ctx, form); } } restore_form{api, ctx, form_before};
    if (api.wfa_set_form) api.wfa_set_form(ctx, in.wfa_form);
    std::vector<vgk_wfa_problem> problems(n);
    uint64_t bases = 0;
    for (uint32_t i = 0; i < n; ++i) {
        vgk_wfa_problem& p = problems[i];
        p.seq = in.seqs + in.seq_off[i]; p.seq_len = (uint32_t)(in.seq_off[i + 1] - in.seq_off[i]); p.mode = in.mode[i];
        p.from_node = in.from_node[i]; p.from_offset = in.from_offset[i]; p.to_node = in.to_node[i]; p.to_offset = in.to_offset[i];
        bases += p.seq_len;
    }
    std::vector<vgk_wfa_result> results(n);
    size_t written[2] = {0, 0};
    (void)bases;
    // The chainer knows how far apart two anchors are in the graph: a connect whose sequence differs from that by g bases holds a gap of
    // g bases and costs WFA what a much longer link costs.  The hint puts such links first in the launch, which otherwise ends ~17 ms
    // after the LAST of them happens to be handed out.  It changes the order only.
    if (in.graph_distance && n && api.wfa_set_cost_hints) {
        std::vector<uint32_t> hints(n, 0);
        for (uint32_t i = 0; i < n; ++i) {
            if (in.mode[i] != VGK_WFA_CONNECT) continue;
            const int64_t gap = (int64_t)problems[i].seq_len - (int64_t)in.graph_distance[i];
            hints[i] = (uint32_t)std::min<int64_t>(64 * (gap < 0 ? -gap : gap), 60000);
        }
        api.wfa_set_cost_hints(ctx, hints.data(), n);
    }
    int rc = n ? api.wfa_extend(ctx, index, model, problems.data(), n, results.data(), nullptr, 0, nullptr, 0, written) : VGK_OK;        // scores only
    if (rc != VGK_OK && rc != VGK_ETOOBIG) return rc;                         // (single declined problems are in results[].status)
    lap(0);
    // 2. the declined links as align_sequence_between requests
    std::deque<Alignment> alignments; std::vector<uint32_t> link_of;
    Alignment whole;
    ChainConnector connector(aligner, graph, in.max_dp_cells);
    auto position = [&](uint32_t oriented, int64_t offset) { Position p; const handle_t h = graph.handle_of(oriented); p.node_id = graph.get_id(h); p.is_reverse = graph.get_is_reverse(h); p.offset = offset; return p; };
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_wfa_result& r = results[i];
        if (r.status == VGK_OK && r.ok) { out.link_score[i] = r.score; out.link_source[i] = ChainStageOutput::WFA; continue; }
        out.wfa_status[i] = r.status != VGK_OK ? r.status : VGK_ENOBAND;
        ++out.n_declined;
        const bool connect = in.mode[i] == VGK_WFA_CONNECT;
        if (connect && in.graph_distance && in.graph_distance[i] == 0 && problems[i].seq_len) {       // nothing lies between the anchors: the link is an insertion (:2996-3008)
            const int32_t len = (int32_t)problems[i].seq_len;
            out.link_score[i] = -((int32_t)aligner.scorer->gap_open + (len - 1) * (int32_t)aligner.scorer->gap_extension);      // score_gap (src/alignment_scorer.cpp:24-26)
            out.link_source[i] = ChainStageOutput::UNLOCALIZED; continue;
        }
        if (!connect && !in.dp_for_tails) continue;
        // WFA's endpoints are the bases next to the link; align_sequence_between's are the gaps between bases (:3074: graph_end / graph_start)
        const Position left = in.mode[i] == VGK_WFA_PREFIX ? Position() : position(in.from_node[i], (int64_t)in.from_offset[i] + 1);
        const Position right = in.mode[i] == VGK_WFA_SUFFIX ? Position() : position(in.to_node[i], (int64_t)in.to_offset[i]);
        const size_t link_length = problems[i].seq_len;
        // the longest gap a read of this length can detect in this stretch (:3067, :2700, :3250)
        const size_t whole_length = in.read_length ? in.read_length[i] : link_length;        // (only the read's length matters: one buffer, resized when it changes)
        if (whole.sequence.size() != whole_length) whole.sequence.assign(whole_length, 'N');
        const size_t begin = in.read_begin ? in.read_begin[i] : 0;
        const size_t gap = std::min(connect ? in.max_middle_gap : in.max_tail_gap, longest_detectable_gap_in_range(whole, begin, begin + link_length, &aligner));
        const size_t path_length = std::max<size_t>(in.graph_distance ? in.graph_distance[i] : link_length, link_length) + gap;
        alignments.emplace_back();
        alignments.back().sequence.assign(problems[i].seq, link_length);
        connector.add(left, right, path_length, gap, alignments.back(), true);
        link_of.push_back(i);
    }
    lap(1);
    // 3. local graphs on the host threads, one flush of DP problems, answers back in base-graph space
    if (!link_of.empty()) connector.run(in.threads);
    out.ms[2] += connector.last_extract_ms; out.ms[3] += connector.last_align_ms;
    t0 = clock::now();
    for (size_t k = 0; k < link_of.size(); ++k) {
        const uint32_t i = link_of[k];
        const ChainConnector::Outcome& o = connector.outcome(k);
        switch (o.status) {
            case ChainConnector::ALIGNED:
                if (alignments[k].path.mapping.empty() && !alignments[k].sequence.empty()) { ++out.n_too_big; break; }
                out.link_score[i] = alignments[k].score; out.link_source[i] = ChainStageOutput::BETWEEN; ++out.n_between; break;
            case ChainConnector::NO_GRAPH: ++out.n_no_graph; break;
            case ChainConnector::TOO_BIG: ++out.n_too_big; break;
            default: ++out.n_failed; break;
        }
    }
    for (uint32_t i = 0; i < n; ++i) out.chain_score[in.read_of[i]] += out.link_score[i];
    if (in.anchor_score) for (uint32_t r = 0; r < in.n_reads; ++r) out.chain_score[r] += in.anchor_score[r];
    out.ms[4] += connector.last_translate_ms;
    lap(4);
    if (!in.anchor_off) return VGK_OK;
    // 4. one alignment per read: the pieces in read order, composed on the device
    if (!api.chain_stitch) return VGK_EUNSUPPORTED;
    std::vector<int64_t> request_of(n, -1);
    for (size_t k = 0; k < link_of.size(); ++k) request_of[link_of[k]] = (int64_t)k;
    std::vector<vgk_chain_piece> pieces; std::vector<uint64_t> piece_off((size_t)in.n_reads + 1, 0);
    std::vector<uint32_t> p_edits; std::vector<vgk_chain_mapping> p_maps;
    pieces.reserve((size_t)n * 2 + in.n_reads); p_edits.reserve((size_t)n + in.anchor_off[in.n_reads]);
    out.read_broken.assign(in.n_reads, 0);
    auto unlocalized = [&](uint64_t length) {                                   // WFAAlignment::make_unlocalized_insertion: no path, one insertion
        if (!length) return;
        vgk_chain_piece pc{}; pc.kind = VGK_PIECE_ALIGNMENT; pc.edit_begin = (uint32_t)p_edits.size(); pc.n_edits = 1;
        p_edits.push_back((uint32_t)length << 2 | VGK_WFA_INSERTION); pieces.push_back(pc);
    };
    auto anchor = [&](uint64_t a) {                                             // to_wfa_alignment (:4083-4104): one match run
        vgk_chain_piece pc{}; pc.kind = VGK_PIECE_ALIGNMENT; pc.node_offset = in.anchor_node_offset[a];
        pc.path_begin = (uint32_t)in.anchor_path_off[a]; pc.path_len = (uint32_t)(in.anchor_path_off[a + 1] - in.anchor_path_off[a]);
        pc.edit_begin = (uint32_t)p_edits.size(); pc.n_edits = 1;
        p_edits.push_back(in.anchor_length[a] << 2 | VGK_WFA_MATCH); pieces.push_back(pc);
    };
    uint32_t i = 0;
    for (uint32_t r = 0; r < in.n_reads; ++r) {
        uint64_t a = in.anchor_off[r]; const uint64_t a_end = in.anchor_off[r + 1];
        const uint32_t first = i;
        uint32_t last = i; while (last < n && in.read_of[last] == r) ++last;
        bool broken = false;
        if ((first == last || in.mode[first] != VGK_WFA_PREFIX) && a < a_end) anchor(a++);
        for (i = first; i < last && !broken; ++i) {
            const vgk_wfa_result& w = results[i];
            const uint32_t len = problems[i].seq_len;
            switch (out.link_source[i]) {
                case ChainStageOutput::WFA: {
                    vgk_chain_piece pc{}; pc.kind = VGK_PIECE_LINK; pc.link = i;
                    if (in.mode[i] == VGK_WFA_PREFIX) unlocalized(w.seq_offset);                                        // the tail's unaligned start is soft-clipped (:2632-2640)
                    pieces.push_back(pc);
                    if (in.mode[i] == VGK_WFA_SUFFIX && w.seq_offset + w.length < len) unlocalized(len - (w.seq_offset + w.length));       // (:3176-3181)
                    break; }
                case ChainStageOutput::BETWEEN: {
                    const Alignment& aln = alignments[(size_t)request_of[i]];
                    vgk_chain_piece pc{}; pc.kind = VGK_PIECE_PATH; pc.path_begin = (uint32_t)p_maps.size(); pc.path_len = (uint32_t)aln.path.mapping.size();
                    for (const Mapping& m : aln.path.mapping) {
                        vgk_chain_mapping fm; fm.node = m.position.node_id ? graph.oriented(graph.get_handle(m.position.node_id, m.position.is_reverse)) : VGK_WFA_NO_NODE;
                        fm.offset = (uint32_t)m.position.offset; fm.edit_begin = (uint32_t)p_edits.size(); fm.n_edits = (uint32_t)m.edit.size();
                        for (const Edit& e : m.edit) {
                            const uint32_t kind = e.from_length == e.to_length ? (e.sequence.empty() ? VGK_WFA_MATCH : VGK_WFA_MISMATCH) : e.from_length == 0 ? VGK_WFA_INSERTION : VGK_WFA_DELETION;
                            if (e.from_length != e.to_length && e.from_length && e.to_length) broken = true;              // (no aligner of this path makes such an edit)
                            p_edits.push_back((uint32_t)std::max(e.from_length, e.to_length) << 2 | kind);
                        }
                        p_maps.push_back(fm);
                    }
                    pieces.push_back(pc);
                    break; }
                case ChainStageOutput::UNLOCALIZED: unlocalized(len); break;
                default:                                                                                                // nothing aligned this link
                    if (in.mode[i] == VGK_WFA_CONNECT) broken = true;                                                    // the chain stops here (:3057, :3093)
                    else unlocalized(len);                                                                               // a tail left unaligned: soft clip (:2674, :3217)
                    break;
            }
            if (broken) break;
            if (in.mode[i] != VGK_WFA_SUFFIX && a < a_end) anchor(a++);
        }
        if (broken) {                                                           // link i and everything behind it in the read: one insertion without a position
            uint64_t rest = 0;
            for (uint32_t k = i; k < last; ++k) rest += problems[k].seq_len;
            for (; a < a_end; ++a) rest += in.anchor_length[a];
            unlocalized(rest);
            out.read_broken[r] = 1; ++out.n_broken;
        } else if (a != a_end) return VGK_EINVAL;                               // anchors and links do not interleave as stated (chain_stage.hpp)
        i = last;
        piece_off[r + 1] = pieces.size();
    }
    out.read_result.resize(in.n_reads);
    size_t stitched[2] = {0, 0};
    if (out.mappings.size() < 1024) out.mappings.resize(std::max<size_t>(1024, (bases + in.anchor_off[in.n_reads] * 32) / 12));       // (a first guess; afterwards the last batch's size and a quarter)
    if (out.edits.size() < 1024) out.edits.resize(std::max<size_t>(1024, (bases + in.anchor_off[in.n_reads] * 32) / 10));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (api.host_register && (out.pinned_m != out.mappings.data() || out.pinned_e != out.edits.data())) {
            out.release(api, ctx);
            if (api.host_register(ctx, out.mappings.data(), out.mappings.size() * sizeof(vgk_chain_mapping)) == VGK_OK) out.pinned_m = out.mappings.data();
            if (api.host_register(ctx, out.edits.data(), out.edits.size() * sizeof(uint32_t)) == VGK_OK) out.pinned_e = out.edits.data();
        }
        rc = api.chain_stitch(ctx, index, pieces.data(), piece_off.data(), in.n_reads, in.anchor_nodes, (size_t)in.anchor_path_off[in.anchor_off[in.n_reads]],
                              p_maps.data(), p_maps.size(), p_edits.data(), p_edits.size(), out.read_result.data(), out.mappings.data(), out.mappings.size(),
                              out.edits.data(), out.edits.size(), stitched);
        if (rc != VGK_EOPS) break;
        out.release(api, ctx);
        out.mappings.resize(stitched[0] + stitched[0] / 4 + 1024); out.edits.resize(stitched[1] + stitched[1] / 4 + 1024);
    }
    if (rc != VGK_OK) return rc;
    out.n_mappings = stitched[0]; out.n_edits = stitched[1];
    if (api.chain_stitch_last_ms) out.stitch_kernel_ms = api.chain_stitch_last_ms(ctx);
    lap(5);
    return VGK_OK;
}

}  // namespace vgamd
