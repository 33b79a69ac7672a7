// chain_alignment.hpp — giraffe's alignment of the read bases BETWEEN two anchors of a chain, or beyond its first / last anchor, when
// WFAExtender has declined them (reference: MinimizerMapper::with_dagified_local_graph / align_sequence_between /
// align_sequence_between_consistently / longest_detectable_gap_in_range, src/minimizer_mapper_from_chains.cpp:3342-3920; band
// padding: src/algorithms/pad_band.cpp).  Same names, argument meaning and error behaviour as the reference's static members.
//
// MI355X-first addition: ChainConnector.  vg calls align_sequence_between once per gap from one OpenMP thread per read; the engine
// wants thousands of problems per launch.  A ChainConnector takes the same requests, cuts out and dagifies their local graphs on
// host threads, hands every DP problem to one AlignmentBatch flush (one launch per kernel family) and then translates every
// alignment back, exactly as the direct call would have.
#pragma once
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "aligner.hpp"
#include "local_graph.hpp"

namespace vgamd {

// thrown when no acceptable graph lies between two anchors (src/minimizer_mapper.hpp: ChainAlignmentFailedError)
class ChainAlignmentFailedError : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

using BandPaddingFunction = std::function<size_t(const Alignment&, const HandleGraph&)>;
// multiplier * sqrt(read length) + 1, capped (src/algorithms/pad_band.cpp:17-41); the _min_ form takes the shorter of read and graph
BandPaddingFunction pad_band_random_walk(double band_padding_multiplier = 1.0, size_t band_padding_memo_size = 2000,
                                         size_t max_padding = std::numeric_limits<size_t>::max());
BandPaddingFunction pad_band_min_random_walk(double band_padding_multiplier = 1.0, size_t band_padding_memo_size = 2000,
                                             size_t max_padding = std::numeric_limits<size_t>::max());
BandPaddingFunction pad_band_constant(size_t band_padding);

// The local graph of one request, made alignable: cut out (extract_connecting_graph / extract_extending_graph), strands split,
// dagified from the anchors, with the translation back to the base graph.
class DagifiedLocalGraph {
public:
    // throws ChainAlignmentFailedError when both anchors are empty or nothing connects them within max_path_length
    DagifiedLocalGraph(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, const HandleGraph& graph);
    DagifiedLocalGraph(const DagifiedLocalGraph&) = delete;
    LocalGraph dagified;
    handle_t left_anchor_handle{}, right_anchor_handle{};                     // as the callback of with_dagified_local_graph gets them
    std::pair<nid_t, bool> to_base(const handle_t& dagified_handle) const;      // (base node id, is_reverse)
    // remove every tip that is not an acceptable source / sink, again and again (:3674-3737) -> how many rounds removed something
    size_t trim_tips();
private:
    LocalGraph local_;
    StrandSplitView split_{&local_};
    std::unordered_map<nid_t, nid_t> local_to_base_, dagified_to_split_;
    bool has_left_, has_right_;
};

using DagifiedCallback = std::function<void(LocalGraph&, const handle_t&, const handle_t&, const std::function<std::pair<nid_t, bool>(const handle_t&)>&)>;
void with_dagified_local_graph(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, const HandleGraph& graph,
                               const DagifiedCallback& callback);

// [sequence_begin, sequence_end) as offsets into aln.sequence (the reference takes iterators)
size_t longest_detectable_gap_in_range(const Alignment& aln, size_t sequence_begin, size_t sequence_end, const GSSWAligner* aligner);

bool align_sequence_between(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length,
                            const HandleGraph* graph, const Aligner* aligner, Alignment& alignment, const std::string* alignment_name = nullptr,
                            size_t max_dp_cells = std::numeric_limits<size_t>::max(),
                            const BandPaddingFunction& choose_band_padding = pad_band_random_walk());
bool align_sequence_between_consistently(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length,
                                         const HandleGraph* graph, const Aligner* aligner, Alignment& alignment,
                                         const std::string* alignment_name = nullptr, size_t max_dp_cells = std::numeric_limits<size_t>::max(),
                                         const BandPaddingFunction& choose_band_padding = pad_band_random_walk());

// reverse_complement_alignment (src/alignment.cpp:3316-3336, src/path.cpp:1791-1882)
std::string reverse_complement(const std::string& seq);
Alignment reverse_complement_alignment(const Alignment& aln, const std::function<int64_t(nid_t)>& node_length);

// Many align_sequence_between requests answered by one flush.  add() only records; run() answers the requests added since the last
// run(): the host work on `threads` threads, one engine flush, the translation back.  After run(): outcome(i).  The Alignment objects and the graph must outlive run().
class ChainConnector {
public:
    enum Status { ALIGNED, NO_GRAPH /* ChainAlignmentFailedError */, TOO_BIG /* band matrices / X-drop cells over max_dp_cells */,
                  NO_ALIGNMENT_IN_BAND, FAILED /* anything else; see message */ };
    struct Outcome { Status status = FAILED; std::string message; bool did_align = false; size_t trims = 0; };
    ChainConnector(const Aligner& aligner, const HandleGraph& graph, size_t max_dp_cells = std::numeric_limits<size_t>::max(),
                   BandPaddingFunction choose_band_padding = pad_band_random_walk());
    ~ChainConnector();
    // consistently: answer as align_sequence_between_consistently does (the strand to align on chosen from the anchors, not the caller)
    size_t add(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length, Alignment& alignment,
               bool consistently = false);
    void run(unsigned threads = 0);
    size_t size() const { return requests_.size(); }
    const Outcome& outcome(size_t i) const { return outcomes_[i]; }
    double last_extract_ms = 0, last_align_ms = 0, last_translate_ms = 0;
private:
    struct Request;
    const Aligner& aligner_; const HandleGraph& graph_; size_t max_dp_cells_; BandPaddingFunction choose_band_padding_;
    std::vector<std::unique_ptr<Request>> requests_;
    std::vector<Outcome> outcomes_;
    size_t answered_ = 0;
};

}  // namespace vgamd
