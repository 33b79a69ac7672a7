// gbwt_extender.hpp — host-side mirror of vg's haplotype-aware gapless seed extension
// (reference: src/gbwt_extender.hpp:30-217, src/gbwt_extender.cpp:17-176, :533-737).  `GaplessExtender::extend`
// hands the per-seed best-first searches and the set rules to the MI355X engine (vgk_gapless_extend); the value
// type `GaplessExtension` keeps the reference's members and helpers.
//
// gbwt / gbwtgraph are empty submodules in the reference snapshot, so `HaplotypeGraph` stands in for
// gbwtgraph::GBWTGraph: node sequences plus the haplotype threads, indexed on the device when an extender is built.
#pragma once
#include <algorithm>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "aligner.hpp"

namespace vgamd {

// What gbwtgraph::GBWTGraph(gbwt_index, graph) is to the extender: the nodes the threads touch and the threads.
class HaplotypeGraph : public HandleGraph {
public:
    HaplotypeGraph(const HandleGraph& graph, const std::vector<std::vector<handle_t>>& threads);
    bool has_node(nid_t id) const override { return index_of_.count(id) != 0; }
    size_t get_length(const handle_t& h) const override { return seqs_[index_of_.at(get_id(h))].size(); }
    std::string get_sequence(const handle_t& h) const override;
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return ids_.size(); }
    nid_t min_node_id() const override { return ids_.empty() ? 0 : ids_.front(); }
    nid_t max_node_id() const override { return ids_.empty() ? 0 : ids_.back(); }
    // the engine's view: nodes in id order, oriented node = 2 * index + is_reverse
    uint32_t oriented(const handle_t& h) const { return 2u * (uint32_t)index_of_.at(get_id(h)) + (get_is_reverse(h) ? 1u : 0u); }
    handle_t handle_of(uint32_t oriented_node) const { return get_handle(ids_[oriented_node >> 1], oriented_node & 1u); }
    const std::vector<std::string>& sequences() const { return seqs_; }
    const std::vector<std::vector<uint32_t>>& threads() const { return threads_; }
private:
    std::vector<nid_t> ids_;                          // ascending
    std::vector<std::string> seqs_;
    std::unordered_map<nid_t, size_t> index_of_;
    std::vector<std::vector<uint32_t>> threads_;      // oriented nodes
    std::set<std::pair<int64_t, int64_t>> edges_;     // handle pairs that some thread crosses
    std::vector<std::vector<int64_t>> next_, prev_;   // the same as adjacency lists, by oriented node
};

// src/gbwt_extender.hpp:30-90
struct GaplessExtension {
    typedef std::pair<handle_t, int64_t> seed_type;          // (handle, read_offset - node_offset)
    struct SearchState { int64_t node = 0; std::pair<size_t, size_t> range{0, 0}; };
    struct BidirectionalState { SearchState forward, backward; bool operator==(const BidirectionalState& o) const {
        return forward.node == o.forward.node && forward.range == o.forward.range && backward.node == o.backward.node && backward.range == o.backward.range; } };

    std::vector<handle_t>     path;
    size_t                    offset = 0;
    BidirectionalState        state;                          // the engine's oriented-node numbering
    std::pair<size_t, size_t> read_interval{0, 0};
    std::vector<size_t>       mismatch_positions;
    int32_t                   score = 0;
    bool                      left_full = false, right_full = false;

    size_t length() const { return read_interval.second - read_interval.first; }
    bool empty() const { return length() == 0; }
    bool full() const { return left_full & right_full; }
    bool exact() const { return mismatch_positions.empty(); }
    size_t mismatches() const { return mismatch_positions.size(); }
    bool contains(const HandleGraph& graph, const seed_type& seed) const;
    Position starting_position(const HandleGraph& graph) const;
    Position tail_position(const HandleGraph& graph) const;
    size_t tail_offset(const HandleGraph& graph) const;
    size_t overlap(const HandleGraph& graph, const GaplessExtension& another) const;
    Path to_path(const HandleGraph& graph, const std::string& sequence) const;
    bool operator<(const GaplessExtension& another) const { return score < another.score; }
    bool operator==(const GaplessExtension& another) const { return read_interval == another.read_interval && state == another.state && offset == another.offset; }
    bool operator!=(const GaplessExtension& another) const { return !(*this == another); }
};

// src/gbwt_extender.hpp:140-217
class GaplessExtender {
public:
    typedef GaplessExtension::seed_type seed_type;
    // the reference's cluster is a hash set of seeds; here an ordered list of distinct seeds, visited in that order
    typedef std::vector<seed_type> cluster_type;
    constexpr static size_t MAX_MISMATCHES = 4;
    constexpr static double OVERLAP_THRESHOLD = 0.8;

    GaplessExtender(const HaplotypeGraph& graph, const Aligner& aligner);    // uploads the haplotype index
    ~GaplessExtender();
    GaplessExtender(const GaplessExtender&) = delete;
    GaplessExtender& operator=(const GaplessExtender&) = delete;

    static seed_type to_seed(const HandleGraph& graph, const Position& pos, size_t read_offset) {
        return seed_type(graph.get_handle(pos.node_id, pos.is_reverse), (int64_t)read_offset - pos.offset);
    }
    static handle_t get_handle(seed_type seed) { return seed.first; }
    static size_t get_node_offset(seed_type seed) { return seed.second < 0 ? (size_t)(-seed.second) : 0; }
    static size_t get_read_offset(seed_type seed) { return seed.second < 0 ? 0 : (size_t)seed.second; }

    std::vector<GaplessExtension> extend(const cluster_type& cluster, std::string sequence, size_t max_mismatches = MAX_MISMATCHES,
                                         double overlap_threshold = OVERLAP_THRESHOLD, bool trim = true) const;
    static bool full_length_extensions(const std::vector<GaplessExtension>& result, size_t max_mismatches = MAX_MISMATCHES);

    // MinimizerMapper::get_tail_forest (src/minimizer_mapper.cpp:5745-5860; a member of the mapper there, here beside the extender whose
    // haplotype index the walk reads): the trees a tail of `extended_seed` can align to, each as the (parent index, handle) list and the
    // root trim that TreeSubgraph's constructor takes (:5838); *longest_detectable_gap as the reference computes it (:5809).  The walk
    // itself (dfs_gbwt :5909-6013) runs on the engine (vgk_tail_forest).
    struct TailTree { std::vector<std::pair<int64_t, handle_t>> tree; size_t root_trim = 0; };
    std::vector<TailTree> get_tail_forest(const GaplessExtension& extended_seed, size_t read_length, bool left_tails, size_t* longest_detectable_gap = nullptr) const;

    const HaplotypeGraph* graph;
    const Aligner*        aligner;
private:
    vgk_haplo* index = nullptr;
};

// src/gbwt_extender.hpp:233-307
struct WFAAlignment {
    enum Edit { match, mismatch, insertion, deletion };

    static WFAAlignment from_extension(const GaplessExtension& extension);
    static WFAAlignment make_unlocalized_insertion(size_t sequence_offset, size_t length, int score);
    static WFAAlignment make_empty();

    std::vector<handle_t> path;
    std::vector<std::pair<Edit, uint32_t>> edits;
    uint32_t node_offset = 0;
    uint32_t seq_offset = 0;
    uint32_t length = 0;
    int32_t  score = 0;
    bool     ok = false;

    operator bool() const { return ok; }
    bool empty() const { return path.empty() && edits.empty(); }
    bool unlocalized_insertion() const;
    int64_t final_offset(const HandleGraph& graph) const;
    void flip(const HandleGraph& graph, const std::string& sequence);
    void append(Edit edit, uint32_t length);
    void join(const WFAAlignment& second);
    Path to_path(const HandleGraph& graph, const std::string& sequence) const;
};

// src/gbwt_extender.hpp:346-461.  connect / suffix / prefix run on the MI355X engine (vgk_wfa_extend); the batch overload
// hands a whole set of problems over in one call, which is how the engine is meant to be fed.
class WFAExtender {
public:
    struct ErrorModel {
        struct Event {
            double per_base; int32_t min, max;
            int32_t evaluate(size_t length) const { return std::min(max, (int32_t)(per_base * length) + min); }
        };
        Event mismatches, gaps, gap_length, distance;
        constexpr static Event default_mismatches() { return {0.03, 1, 6}; }
        constexpr static Event default_gaps() { return {0.05, 1, 10}; }
        constexpr static Event default_gap_length() { return {0.1, 1, 20}; }
        constexpr static Event default_distance() { return {0.1, 10, 200}; }
    };
    static const ErrorModel default_error_model;

    WFAExtender(const HaplotypeGraph& graph, const Aligner& aligner, const ErrorModel& error_model = default_error_model);   // uploads the haplotype index
    ~WFAExtender();
    WFAExtender(const WFAExtender&) = delete;
    WFAExtender& operator=(const WFAExtender&) = delete;

    // `from` and `to` are exclusive: the alignment starts one base after `from` and ends one base before `to`
    WFAAlignment connect(std::string sequence, Position from, Position to) const;
    WFAAlignment suffix(const std::string& sequence, Position from) const;
    WFAAlignment prefix(const std::string& sequence, Position to) const;

    struct Problem { enum Kind { CONNECT, SUFFIX, PREFIX } kind; std::string sequence; Position from, to; };
    std::vector<WFAAlignment> extend(const std::vector<Problem>& problems) const;      // one engine call for all of them

    const HaplotypeGraph* graph;
    const Aligner*        aligner;
    const ErrorModel*     error_model;
    const vgk_haplo* engine_index() const { return index; }     // the haplotype index in HBM, for batch stages that call the engine themselves (chain_stage.hpp)
private:
    vgk_haplo* index = nullptr;
};

}  // namespace vgamd
