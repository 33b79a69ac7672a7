// rescue_resident.hpp — giraffe's mate rescue for many pairs at once ON THE RESIDENT GRAPH: what rescue_stage.hpp does with one HashGraph, one
// Alignment and three engine calls full of per-problem graphs per batch, done over flat arrays and windows of ONE graph that lives in HBM
// (MinimizerMapper::attempt_rescue, reference src/minimizer_mapper.cpp:3264-3440, from the point where it has its rescue nodes):
//   * the two X-drop passes of Aligner::align_xdrop (DozeuInterface::align, src/dozeu_interface.cpp:608-685) are EXTENSION WINDOWS
//     (vgk_gssw_pack_extensions): {read, node range, start position, direction} — the sub-DAG of each pass is derived on the device;
//   * dozeu's 15-base scan for a mate without a seed (scan_seed_position, :143-208) and the full DP that fix_dozeu_score / align_xdrop fall back
//     to (src/minimizer_mapper.cpp:3510-3515, src/aligner.cpp:848-854) are plain windows (vgk_gssw_pack_windows);
//   * the head position, the conversion of the traced pass into the alignment (unreversal, match / mismatch edits by character, the read part
//     beyond the head as an insertion: :498-526), MinimizerMapper::fix_dozeu_score (:3502-3517) and fix_dozeu_end_deletions (:3519-3565) run on
//     chunked host threads over the flat results: a few operations per mate, no graph object, no string.
// The host's share per mate is O(read length + CIGAR), the engine's three rounds of kernels see node ranges, never node sequences from the host.
// Results are those of run_rescue_stage (rescue_stage.hpp) on the same requests — tests/test_paired_stage.py holds the two against each other
// alignment by alignment, this path on the HIP engine / the emulator, that one over the oracle.
#pragma once
#include <cstdint>
#include <vector>
#include "aligner.hpp"
#include "rescue_stage.hpp"

namespace vgamd {

// the graph, resident in the aligner's engine context (vgk_graph_create) beside the host's view of it (borrowed: node_len, seq, pred CSR must
// outlive this object); node i has id i + 1, nodes in topological order, predecessor lists ascending
struct ResidentRescueGraph {
    const Aligner* aligner = nullptr;
    vgk_dgraph* dg = nullptr;
    uint32_t n_nodes = 0; const uint32_t* node_len = nullptr; const char* seq = nullptr;
    std::vector<uint64_t> seq_off;                       // [n_nodes + 1]
    ResidentRescueGraph(const Aligner& aligner, uint32_t n_nodes, const uint32_t* node_len, const char* seq, const uint32_t* pred_off, const uint32_t* pred_idx);
    ~ResidentRescueGraph();
    ResidentRescueGraph(const ResidentRescueGraph&) = delete;
    ResidentRescueGraph& operator=(const ResidentRescueGraph&) = delete;
};

struct RescueRequestFlat {
    uint64_t read_off = 0; uint32_t read_len = 0;           // the mate in `reads`, as it reads along the forward strand of the subgraph
    uint32_t node_lo = 0, node_hi = 0;                      // rescue nodes = [node_lo, node_hi)
    int64_t seed_begin = 0, seed_end = 0, seed_node = -1, seed_offset = 0;      // as RescueRequest
};
struct RescueTiming { double classify_ms = 0, first_pass_ms = 0, second_pass_ms = 0, finish_ms = 0, fallback_ms = 0; uint64_t first_pass = 0, scans = 0, second_pass = 0, fallbacks = 0;
                      double kernel_ms = 0; uint64_t alg_bytes = 0, cells = 0; };      // kernels of the three rounds (HIP events), their algorithmic bytes and DP cells

// results[k] as run_rescue_stage fills them; ops (nullable): the final alignments as (node index, VGK_OP_M / I / D, length) runs, mapping by
// mapping, ops_begin[k] .. ops_begin[k + 1] those of request k (a match / mismatch stretch is one M run; soft clips are I)
void run_rescue_stage_resident(const Aligner& aligner, const ResidentRescueGraph& graph, const char* reads, size_t reads_bytes,
                               const std::vector<RescueRequestFlat>& requests, uint64_t max_dozeu_cells, unsigned host_threads,
                               std::vector<RescueResult>& results, std::vector<vgk_op>* ops, std::vector<uint64_t>* ops_begin, RescueTiming* timing = nullptr);

}  // namespace vgamd
