// rescue_stage.hpp — the rescue half of giraffe's paired-end path for many pairs at once: what MinimizerMapper::attempt_rescue does to the
// mate that found no full-length extension (reference src/minimizer_mapper.cpp:3264-3440), from the point where it HAS its rescue nodes:
//   the subgraph's topological order and its bases (:3352-3361), the max_dozeu_cells guard (:3372-3381), the best gapless extension inside the
//   subgraph as dozeu's seed (:3322-3348), Aligner::align_xdrop over the order with longest_detectable_gap as the gap limit (:3383-3386),
//   fix_dozeu_score and fix_dozeu_end_deletions (:3387-3388).
// Here every pair's X-drop passes go to the engine together (Aligner::align_xdrop_many), the subgraphs are built and the fix-ups applied
// on host threads.  What comes BEFORE — which nodes lie at the fragment's distance from the mapped mate (subgraph_in_distance_range over the
// SnarlDistanceIndex, absent from the snapshot) — is the caller's: it names a run of nodes of a graph whose node order is topological.
#pragma once
#include <cstdint>
#include <vector>
#include "aligner.hpp"

namespace vgamd {

struct RescueGraph {                    // the base graph, forward strand; node i has id i + 1; successors as CSR over node indices; edges go up
    uint32_t n_nodes = 0; const uint32_t* node_len = nullptr; const uint64_t* seq_off = nullptr; const char* seq = nullptr;
    const uint32_t* succ_off = nullptr; const uint32_t* succ = nullptr;
};
struct RescueRequest {
    const char* read = nullptr; uint32_t read_len = 0;      // the mate to rescue, as it reads along the forward strand of the subgraph
    uint32_t node_lo = 0, node_hi = 0;                      // rescue nodes = [node_lo, node_hi)
    // the best gapless extension inside the subgraph (dozeu's seed), or seed_node < 0 for none: read interval, first node of its path (index), offset there
    int64_t seed_begin = 0, seed_end = 0, seed_node = -1, seed_offset = 0;
};
// [PARITY deviation, stated: the reference hands longest_detectable_gap (a size_t) to align_xdrop's uint16_t parameter (src/minimizer_mapper.cpp:3383-3385),
// where a value above 65 535 wraps modulo 65 536; both rescue paths here clamp it to 65 535 instead — the saner reading, and the same for every
// scoring and read length a 16-bit score range admits (150-base reads: 81).]
struct RescueResult {
    int32_t score = 0; int32_t status = 0;                  // status: 0 aligned (score may be 0), 1 refused by the cell budget, 2 empty subgraph, 3 the engine declined every route for this mate (first_node = the VGK_E* code; rescue_resident.cpp)
    int64_t first_node = -1, first_offset = 0; uint32_t n_mappings = 0, aligned_read_bases = 0;
};
constexpr uint64_t default_max_dozeu_cells = (uint64_t)(1.5 * 1024 * 1024);      // src/minimizer_mapper.hpp:471

// host_threads 0 = as many as the machine grants.  ops / ops_begin (nullable): the final alignments as (node index, VGK_OP_M / I / D, length) runs,
// mapping by mapping — a stretch of match and substitution edits is one M run, every deletion and insertion edit a run of its own; request k's
// are ops[ops_begin[k] .. ops_begin[k + 1]) (what rescue_resident.hpp's path reports in the same form: the two are held against each other)
void run_rescue_stage(const Aligner& aligner, const RescueGraph& graph, const std::vector<RescueRequest>& requests, uint64_t max_dozeu_cells,
                      unsigned host_threads, std::vector<RescueResult>& results, std::vector<vgk_op>* ops = nullptr, std::vector<uint64_t>* ops_begin = nullptr);

}  // namespace vgamd
