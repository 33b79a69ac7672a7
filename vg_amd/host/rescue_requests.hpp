// rescue_requests.hpp — which mates of a batch of pairs are rescued, from where, with which seed: the table the rescue stage takes, built from
// what the alignment stage left (the extension sets of all 2 n reads) on chunked host threads over flat arrays.
// MinimizerMapper::map_paired decides it per pair (reference src/minimizer_mapper.cpp:1793-1901: a pair with alignments for one end only is a
// rescue candidate) and attempt_rescue (:3264-3440) finds the rescue nodes — subgraph_in_distance_range over the SnarlDistanceIndex, which is an
// absent dependency: here, as in vg_amd/pipeline.py's statement of the same rule, the nodes whose columns lie at the fragment's distance from the
// mapped mate on a graph whose node order is topological [stand-in, stated in DESIGN.md] — and takes the best gapless extension of the lost mate
// inside them as dozeu's seed (:3322-3348).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/vgk.h"

namespace vgamd {

struct RescueRequestTable {
    std::vector<uint32_t> mapped, lost;          // read indices, one entry per rescued pair, in pair order
    std::vector<int64_t> requests;               // 6 per entry: node_lo, node_hi, seed_begin, seed_end, seed_node (-1: none), seed_offset
    std::vector<char> reads;                     // the lost mates as they read along the forward strand of their subgraphs, read_len each
};

// results / extensions / nodes: vgk_gapless_extend's outputs for the 2 n_pairs reads (read 2 i and 2 i + 1 are a pair); col[v] = first column of
// node v (col[n_nodes] = all bases); reads: read_len bases each
void build_rescue_requests(uint32_t n_pairs, const vgk_gapless_result* results, const vgk_extension* extensions, const uint32_t* nodes,
                           uint32_t n_nodes, const int64_t* col, const char* reads, uint32_t read_len, double fragment_mean, double fragment_sd,
                           double rescue_stdevs, unsigned host_threads, RescueRequestTable& out);

}  // namespace vgamd
