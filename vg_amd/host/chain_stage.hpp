// chain_stage.hpp — what giraffe does with a chain of anchors (MinimizerMapper::find_chain_alignment, reference:
// src/minimizer_mapper_from_chains.cpp:2560-3330): every stretch of the read between two anchors, before the first and behind the
// last one goes to WFAExtender (connect / prefix / suffix); what WFA declines goes to align_sequence_between_consistently — the local
// graph between (beyond) the anchors cut out of the GBWTGraph, dagified, and aligned by BandedGlobalAligner (pinned X-drop for a tail)
// — and the pieces' scores add up to the chain's.  Here for a whole batch of reads at once: one vgk_wfa_extend call, the local graphs
// of the declined links on host threads (ChainConnector), one flush of banded / X-drop problems.
//
// This replaces the Python glue (vg_amd/pipeline.py chain_stage) that round 2's configs[4] leg ran on, including its stand-in for
// the extraction: the subgraphs now come from extract_connecting_graph / extract_extending_graph inside the timed step.
#pragma once
#include <cstdint>
#include <vector>
#include "chain_alignment.hpp"
#include "gbwt_extender.hpp"

namespace vgamd {

struct ChainStageInput {
    const char* seqs; const uint64_t* seq_off; uint32_t n_links;        // link i = seqs[seq_off[i], seq_off[i + 1])
    const uint32_t* mode;                                               // VGK_WFA_CONNECT / VGK_WFA_SUFFIX / VGK_WFA_PREFIX
    const uint32_t *from_node, *from_offset, *to_node, *to_offset;      // WFAExtender's exclusive endpoints: oriented nodes of the index, base offsets
    const uint32_t* read_of; uint32_t n_reads;                          // the read a link belongs to
    const uint32_t* graph_distance;                                     // per link, nullable: the chain's distance between the two anchors (:2950; default: the link's length)
    const uint32_t *read_begin, *read_length;                           // per link, nullable: where the link starts in its read and that read's length (for longest_detectable_gap_in_range)
    const int64_t* anchor_score;                                        // per read, nullable: what the anchors themselves contribute
    size_t max_dp_cells = SIZE_MAX;
    size_t max_tail_gap = SIZE_MAX, max_middle_gap = SIZE_MAX;          // MinimizerMapper::max_tail_gap / max_middle_gap
    unsigned threads = 0;
    int wfa_form = VGK_WFA_FORM_WAVE;                                   // vgk_wfa_set_form for the stage's WFA call
    bool dp_for_tails = true;                                           // a declined prefix / suffix goes to pinned X-drop (:2713, :3261); false: it scores 0
    // The anchors themselves, read by read in read order (nullable: scores only).  With them the stage composes ONE alignment per read as
    // find_chain_alignment does (:2606-3295): left tail, anchor, link, anchor, ..., right tail, each piece's Path appended, the whole simplified
    // (vgk_chain_stitch, on the device: the WFA paths and edit runs never come down).  A read's links are PREFIX?, CONNECT*, SUFFIX? in read order and
    // an anchor follows every PREFIX / CONNECT link (and leads the read when it has no PREFIX): anchors of read r = [anchor_off[r], anchor_off[r + 1]).
    // Anchor a is an exact match of anchor_length[a] bases from anchor_node_offset[a] in the first of the oriented nodes
    // anchor_nodes[anchor_path_off[a] .. anchor_path_off[a + 1]) (to_wfa_alignment :4083-4104 makes it of one node; several are allowed).
    const uint64_t* anchor_off = nullptr; const uint32_t* anchor_length = nullptr; const uint32_t* anchor_node_offset = nullptr;
    const uint64_t* anchor_path_off = nullptr; const uint32_t* anchor_nodes = nullptr;
};
struct ChainStageOutput {
    enum Source : uint8_t { WFA = 0, BETWEEN = 1 /* align_sequence_between */, NONE = 2 /* nothing aligned: scores 0 */,
                            UNLOCALIZED = 3 /* a connect WFA declined between anchors at graph distance 0: an insertion without a position, score_gap (:2996-3008) */ };
    std::vector<int32_t> link_score; std::vector<uint8_t> link_source;
    std::vector<int32_t> wfa_status;                                    // per link: vgk_wfa_result.status (declined links: VGK_ETOOBIG ...), ok folded in as VGK_ENOBAND when !ok
    std::vector<int64_t> chain_score;                                   // per read
    uint64_t n_declined = 0, n_between = 0, n_no_graph = 0, n_too_big = 0, n_failed = 0;
    double ms[6] = {0, 0, 0, 0, 0, 0};                                  // wfa call | requests made | local graphs (host threads) | banded + X-drop flush | translation + totals | pieces + vgk_chain_stitch
    // with anchors: the composed alignments, dense and in read order (include/vgk.h: vgk_chain_result / vgk_chain_mapping / edit runs length << 2 | VGK_WFA_*)
    std::vector<vgk_chain_result> read_result; std::vector<vgk_chain_mapping> mappings; std::vector<uint32_t> edits;      // mappings / edits: the first n_mappings / n_edits entries count (the vectors keep their room between batches)
    size_t n_mappings = 0, n_edits = 0;
    // A read whose chain broke at a link nothing could align (the reference leaves the loop there, :3057 / :3093, and treats the rest of the read as
    // its right tail): here the rest of the read — that link, every later anchor and link — becomes one insertion without a position (what the
    // reference makes of a tail beyond max_tail_dp_length, :3217), and the read is flagged
    std::vector<uint8_t> read_broken; uint64_t n_broken = 0;
    double stitch_kernel_ms = 0;
    // mappings / edits are page-locked through the engine while they keep their size (full-rate DMA on the way back: 110 MB per batch of 8 000
    // reads); release() before the engine context goes
    const void* pinned_m = nullptr; const void* pinned_e = nullptr;
    void release(const EngineApi& api, vgk_ctx* ctx);
};

// graph: the HandleGraph the index was built over (HaplotypeGraph: node ids in index order).  -> a VGK_* code
int run_chain_stage(const EngineApi& api, vgk_ctx* ctx, const vgk_haplo* index, const HaplotypeGraph& graph, const Aligner& aligner,
                    const vgk_wfa_error_model* model, const ChainStageInput& in, ChainStageOutput& out);

}  // namespace vgamd
