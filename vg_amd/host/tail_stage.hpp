// tail_stage.hpp — the tails of a batch of gapless extensions, aligned: what MinimizerMapper does per extension at
// src/minimizer_mapper.cpp:5480-5535 (get_tail_forest for either open end, get_best_alignment_against_any_tree, the total score),
// for the whole batch at once over the engine: one vgk_tail_forest call, one batch of window problems (one per tree).
//
// The inputs are vgk_gapless_extend's outputs as they are; this is host glue (which tails exist, where they start, their bases —
// a right tail as it lies in the read, a left tail reverse-complemented, :5660) run on the caller's threads.
#pragma once
#include <cstdint>
#include <vector>
#include "engine.hpp"

namespace vgamd {

struct TailStageInput {
    const char* reads; const uint64_t* read_off; uint32_t n_reads;      // read i = reads[read_off[i], read_off[i + 1])
    const vgk_gapless_result* res; const vgk_extension* ext; const uint32_t* nodes;
    const uint32_t* oriented_len;                                        // length of every oriented node of the index
    int match, gap_open, gap_extend, bonus;                              // for EditAlignmentScorer::longest_detectable_gap (src/alignment_scorer.cpp:264-271)
    uint32_t ops_per_problem;
};
struct TailStageOutput {
    std::vector<int32_t> ext_total;      // per extension: its score + the best alignment of either open tail (:5530)
    std::vector<int32_t> read_score;     // per read: the best total over its extensions
    std::vector<int32_t> tail_score;     // per tail, in the order right tails (by extension) then left tails
    uint64_t n_tails = 0, n_trees = 0, tree_nodes = 0, failed = 0;
    double ms[6] = {0, 0, 0, 0, 0, 0};   // tails derived | vgk_tail_forest | windows + bases | pack | kernels | fetch + totals
};

// returns a VGK_* code; engine errors of single problems are counted in out.failed (their tail scores stay 0: a soft clip)
int run_tail_stage(const EngineApi& api, vgk_ctx* ctx, const vgk_haplo* index, const TailStageInput& in, TailStageOutput& out);

}  // namespace vgamd
