// chain_alignment.cpp — see chain_alignment.hpp.
#include "chain_alignment.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <iostream>
#include <sstream>
#include <thread>
#include <tuple>
#include <unordered_set>

namespace vgamd {

// ---- band padding (src/algorithms/pad_band.cpp) -------------------------------------------------------------------------------------
namespace {
BandPaddingFunction pad_band_by(double multiplier, size_t memo_size, size_t max_padding, std::function<size_t(const Alignment&, const HandleGraph&)> size_of) {
    auto padding_for = [multiplier, max_padding](size_t size) { return std::min<size_t>(max_padding, (size_t)(multiplier * std::sqrt((double)size)) + 1); };
    std::vector<size_t> memo(memo_size);
    for (size_t i = 0; i < memo.size(); ++i) memo[i] = padding_for(i);
    return [memo, padding_for, size_of](const Alignment& aln, const HandleGraph& g) {
        const size_t size = size_of(aln, g);
        return size < memo.size() ? memo[size] : padding_for(size);
    };
}
size_t total_length(const HandleGraph& g) { size_t n = 0; g.for_each_handle_v([&](const handle_t& h) { n += g.get_length(h); }); return n; }
std::string describe(const Position& p) { std::ostringstream s; s << p.node_id << (p.is_reverse ? "-" : "+") << p.offset; return s.str(); }
}  // namespace

BandPaddingFunction pad_band_random_walk(double multiplier, size_t memo_size, size_t max_padding) {
    return pad_band_by(multiplier, memo_size, max_padding, [](const Alignment& aln, const HandleGraph&) { return aln.sequence.size(); });
}
BandPaddingFunction pad_band_min_random_walk(double multiplier, size_t memo_size, size_t max_padding) {
    return pad_band_by(multiplier, memo_size, max_padding, [](const Alignment& aln, const HandleGraph& g) { return std::min(aln.sequence.size(), total_length(g)); });
}
BandPaddingFunction pad_band_constant(size_t band_padding) { return [band_padding](const Alignment&, const HandleGraph&) { return band_padding; }; }

// ---- the local graph of one request ---------------------------------------------------------------------------------------------------
DagifiedLocalGraph::DagifiedLocalGraph(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, const HandleGraph& graph)
    : has_left_(!is_empty(left_anchor)), has_right_(!is_empty(right_anchor)) {
    if (!has_left_ && !has_right_) throw ChainAlignmentFailedError("Cannot align sequence between two unset positions");
    nid_t local_left = 0, local_right = 0;
    if (has_left_ && has_right_) {
        // strictly within max_path_length: looser pruning leaves extra tips (:3352-3360)
        ConnectingGraph cut = extract_connecting_graph(&graph, &local_, (int64_t)max_path_length, left_anchor, right_anchor, true);
        if (cut.to_source.empty())
            throw ChainAlignmentFailedError("Cannot find an acceptable path from " + describe(left_anchor) + " to " + describe(right_anchor) +
                                            " with max path length of " + std::to_string(max_path_length));
        local_to_base_ = std::move(cut.to_source); local_left = cut.left_id; local_right = cut.right_id;
    } else {
        ExtendingGraph cut = extract_extending_graph(&graph, &local_, (int64_t)max_path_length, has_left_ ? left_anchor : right_anchor, !has_left_, false);
        local_to_base_ = std::move(cut.to_source);
        (has_left_ ? local_left : local_right) = cut.cut_id;
    }
    if ((has_left_ && !local_.has_node(local_left)) || (has_right_ && !local_.has_node(local_right)))
        throw std::runtime_error("Extracted graph of " + std::to_string(local_.get_node_count()) + " nodes from " + describe(left_anchor) + " to " +
                                 describe(right_anchor) + " with max path length of " + std::to_string(max_path_length) + " but an anchor's node did not come through");
    // one strand per node, then acyclic from the anchors inwards: the left one read forwards, the right one backwards (:3512-3594)
    std::vector<handle_t> bounding;
    if (has_left_) bounding.push_back(split_.get_overlay_handle(local_.get_handle(local_left, left_anchor.is_reverse)));
    if (has_right_) bounding.push_back(split_.flip(split_.get_overlay_handle(local_.get_handle(local_right, right_anchor.is_reverse))));
    handlealgs::Dagified dag = handlealgs::dagify_from(&split_, bounding, &dagified, max_path_length);
    dagified_to_split_ = std::move(dag.to_source);
    if (has_left_) left_anchor_handle = dag.starts.front();
    if (has_right_) right_anchor_handle = dagified.flip(dag.starts.back());     // facing out of the graph again, like the position
}

std::pair<nid_t, bool> DagifiedLocalGraph::to_base(const handle_t& h) const {
    const auto in_split = dagified_to_split_.find(dagified.get_id(h));
    if (in_split == dagified_to_split_.end()) throw std::runtime_error("ID " + std::to_string(dagified.get_id(h)) + " from dagified graph not found in strand-split graph");
    const handle_t local_handle = split_.get_underlying_handle(split_.get_handle(in_split->second, dagified.get_is_reverse(h)));
    const auto in_base = local_to_base_.find(local_.get_id(local_handle));
    if (in_base == local_to_base_.end()) throw std::runtime_error("ID " + std::to_string(local_.get_id(local_handle)) + " from local graph not found in full base graph");
    return {in_base->second, local_.get_is_reverse(local_handle)};
}

size_t DagifiedLocalGraph::trim_tips() {
    size_t rounds = 0;
    for (;;) {
        std::vector<nid_t> doomed;
        for (const handle_t& tip : handlealgs::find_tips(&dagified)) {
            const bool inward_forward = !dagified.get_is_reverse(tip);
            const bool good_source = inward_forward && (!has_left_ || tip == left_anchor_handle);
            const bool good_sink = !inward_forward && (!has_right_ || tip == dagified.flip(right_anchor_handle));
            // anything else is the wrong orientation or another copy of an anchor's node, or a dead end the dagification left
            if (!good_source && !good_sink && !std::count(doomed.begin(), doomed.end(), dagified.get_id(tip))) doomed.push_back(dagified.get_id(tip));
        }
        if (doomed.empty()) return rounds;
        for (nid_t id : doomed) dagified.destroy_handle(dagified.get_handle(id, false));
        ++rounds;
    }
}

void with_dagified_local_graph(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, const HandleGraph& graph,
                               const DagifiedCallback& callback) {
    DagifiedLocalGraph d(left_anchor, right_anchor, max_path_length, graph);
    callback(d.dagified, d.left_anchor_handle, d.right_anchor_handle, [&](const handle_t& h) { return d.to_base(h); });
}

size_t longest_detectable_gap_in_range(const Alignment& aln, size_t begin_index, size_t end_index, const GSSWAligner* aligner) {
    // the read's middle allows the longest gap; a range on one side of it is bounded by its end nearer the middle (:3630-3653)
    const size_t length = aln.sequence.size(), middle_index = length / 2;
    if (end_index > middle_index && begin_index <= middle_index) return aligner->scorer->longest_detectable_gap(length, middle_index);
    return aligner->scorer->longest_detectable_gap(length, begin_index > middle_index ? begin_index : end_index);
}

// ---- align_sequence_between, in three steps so that many requests can share one engine flush ----------------------------------------
namespace {

enum class Route { BANDED, PINNED, SOFTCLIP };

void warn_trimmed(size_t rounds, const Position& left_anchor, const Position& right_anchor, const DagifiedLocalGraph& d, const std::string* name) {
    if (!rounds) return;
    std::ostringstream msg;
    msg << "warning[MinimizerMapper::align_sequence_between]: Trimmed back tips " << rounds << " times on graph between " << describe(left_anchor)
        << " and " << describe(right_anchor) << " leaving " << d.dagified.get_node_count() << " nodes";
    if (name) msg << " for read " << *name;
    msg << "\n";
    std::cerr << msg.str();
}

// which DP a prepared request takes; a pinned problem too large for X-drop is answered at once with a soft clip in base-graph space
// (:3786-3807)
Route choose_route(const Position& left_anchor, const Position& right_anchor, const DagifiedLocalGraph& d, Alignment& alignment, size_t max_dp_cells,
                   const std::string* name) {
    if (!is_empty(left_anchor) && !is_empty(right_anchor)) return Route::BANDED;
    const size_t cell_count = d.dagified.get_total_length() * alignment.sequence.size();
    if (cell_count <= max_dp_cells) return Route::PINNED;
    std::ostringstream msg;
    msg << "warning[MinimizerMapper::align_sequence_between]: Refusing to fill " << cell_count << " DP cells in tail with Xdrop";
    if (name) msg << " for read " << *name;
    msg << "\n";
    std::cerr << msg.str();
    const Position& at = is_empty(left_anchor) ? right_anchor : left_anchor;
    alignment.clear_path();
    Mapping m; m.position = at;
    Edit e; e.to_length = (int32_t)alignment.sequence.size(); e.sequence = alignment.sequence;
    m.edit.push_back(e);
    alignment.path.mapping.push_back(m);
    return Route::SOFTCLIP;
}

// the alignment, in dagified-graph coordinates, back into the base graph's (:3817-3864)
void translate_back(const Position& left_anchor, const Position& right_anchor, const DagifiedLocalGraph& d, const HandleGraph* graph, Alignment& alignment) {
    std::vector<Mapping>& mappings = alignment.path.mapping;
    for (size_t i = 0; i < mappings.size(); ++i) {
        Position& p = mappings[i].position;
        const handle_t h = d.dagified.get_handle(p.node_id, p.is_reverse);
        const std::pair<nid_t, bool> base = d.to_base(h);
        if (i == 0) {
            // An alignment that starts on (a copy of) an anchor's node — possibly a cut one, which is as long as the anchor's own
            // piece — gets back the bases the cut took away in front of it.
            if (!is_empty(left_anchor) && base.first == left_anchor.node_id && base.second == left_anchor.is_reverse) {
                if (d.dagified.get_length(h) == d.dagified.get_length(d.left_anchor_handle)) p.offset += left_anchor.offset;
            } else if (!is_empty(right_anchor) && base.first == right_anchor.node_id && base.second != right_anchor.is_reverse) {
                if (d.dagified.get_length(h) == d.dagified.get_length(d.right_anchor_handle))
                    p.offset += (int64_t)graph->get_length(graph->get_handle(right_anchor.node_id)) - right_anchor.offset;
            }
        }
        p.node_id = base.first; p.is_reverse = base.second;
    }
    if (!mappings.empty()) {                                                    // no empty edit, no empty mapping at the very end
        std::vector<Edit>& edits = mappings.back().edit;
        if (!edits.empty() && edits.back().from_length == 0 && edits.back().to_length == 0 && edits.back().sequence.empty()) edits.pop_back();
        if (edits.empty()) mappings.pop_back();
    }
}

uint16_t as_gap_limit(size_t max_gap_length) { return (uint16_t)std::min<size_t>(max_gap_length, 65535); }

}  // namespace

bool align_sequence_between(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length,
                            const HandleGraph* graph, const Aligner* aligner, Alignment& alignment, const std::string* alignment_name,
                            size_t max_dp_cells, const BandPaddingFunction& choose_band_padding) {
    DagifiedLocalGraph d(left_anchor, right_anchor, max_path_length, *graph);
    warn_trimmed(d.trim_tips(), left_anchor, right_anchor, d, alignment_name);
    switch (choose_route(left_anchor, right_anchor, d, alignment, max_dp_cells, alignment_name)) {
        case Route::SOFTCLIP: return false;
        case Route::BANDED:
            // global, so the alignment runs from a source to a sink; permissive banding; padding by what is being aligned (:3761-3782)
            try { aligner->align_global_banded(alignment, d.dagified, (int32_t)choose_band_padding(alignment, d.dagified), true, max_dp_cells); }
            catch (BandMatricesTooBigException& e) {
                std::cerr << std::string("warning[MinimizerMapper::align_sequence_between]: ") + e.what() + "\n";
                alignment.path.mapping.clear();                                  // "we did not compute an alignment"
            }
            break;
        case Route::PINNED:
            aligner->align_pinned(alignment, d.dagified, !is_empty(left_anchor), true, as_gap_limit(max_gap_length));
            break;
    }
    translate_back(left_anchor, right_anchor, d, graph, alignment);
    return true;
}

// ---- orientation-independent form ------------------------------------------------------------------------------------------------------
std::string reverse_complement(const std::string& seq) {
    std::string out(seq.rbegin(), seq.rend());
    for (char& c : out) switch (c) {
        case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break;
        case 'a': c = 't'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break; case 't': c = 'a'; break;
        default: break;
    }
    return out;
}

Alignment reverse_complement_alignment(const Alignment& aln, const std::function<int64_t(nid_t)>& node_length) {
    Alignment out = aln;
    out.sequence = reverse_complement(aln.sequence);
    out.quality.assign(aln.quality.rbegin(), aln.quality.rend());
    out.path.mapping.clear();
    for (size_t i = aln.path.mapping.size(); i-- > 0;) {
        const Mapping& m = aln.path.mapping[i];
        Mapping r; r.position = m.position;
        if (m.position.node_id != 0) {
            // on the other strand the offset counts the bases behind what the mapping uses
            r.position.offset = node_length(m.position.node_id) - (int64_t)mapping_from_length(m) - m.position.offset;
            r.position.is_reverse = !m.position.is_reverse;
        }
        for (size_t j = m.edit.size(); j-- > 0;) { Edit e = m.edit[j]; e.sequence = reverse_complement(e.sequence); r.edit.push_back(e); }
        r.rank = (int64_t)out.path.mapping.size() + 1;
        out.path.mapping.push_back(std::move(r));
    }
    return out;
}

namespace {
// align_sequence_between_consistently's choice (:3872-3905): does this request run on the other strand, between the swapped anchors?
bool runs_flipped(const Position& left_anchor, const Position& right_anchor, const Alignment& alignment, const Alignment& flipped) {
    auto key = [](const Position& p) { return std::make_tuple(p.node_id, p.is_reverse, p.offset); };
    if (key(left_anchor) < key(right_anchor)) return false;                    // unambiguously in order: as it is
    if (key(left_anchor) == key(right_anchor) && flipped.sequence >= alignment.sequence) return false;     // a tie the sequence does not break either
    return true;
}
Position turned(const Position& p, const HandleGraph* graph) {
    Position r;
    if (!is_empty(p)) { r.node_id = p.node_id; r.is_reverse = !p.is_reverse; r.offset = (int64_t)graph->get_length(graph->get_handle(p.node_id)) - p.offset; }
    return r;
}
void check_one_piece(const Alignment& alignment) {
    for (size_t i = 1; i < alignment.path.mapping.size(); ++i)
        if (alignment.path.mapping[i].position.offset != 0) throw std::logic_error("align_sequence_between_consistently: an offset inside the path");
}
}  // namespace

bool align_sequence_between_consistently(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length,
                                         const HandleGraph* graph, const Aligner* aligner, Alignment& alignment, const std::string* alignment_name,
                                         size_t max_dp_cells, const BandPaddingFunction& choose_band_padding) {
    auto node_length = [&](nid_t id) -> int64_t { return (int64_t)graph->get_length(graph->get_handle(id)); };
    Alignment flipped = reverse_complement_alignment(alignment, node_length);
    if (!runs_flipped(left_anchor, right_anchor, alignment, flipped))
        return align_sequence_between(left_anchor, right_anchor, max_path_length, max_gap_length, graph, aligner, alignment, alignment_name, max_dp_cells, choose_band_padding);
    // align the other strand between the swapped, turned-around anchors, then turn the answer back
    const bool result = align_sequence_between(turned(right_anchor, graph), turned(left_anchor, graph), max_path_length, max_gap_length, graph, aligner, flipped,
                                               alignment_name, max_dp_cells, choose_band_padding);
    alignment = reverse_complement_alignment(flipped, node_length);
    check_one_piece(alignment);
    return result;
}

// ---- ChainConnector ---------------------------------------------------------------------------------------------------------------------
struct ChainConnector::Request {
    Position left, right; size_t max_path_length, max_gap_length; Alignment* alignment;
    Alignment* answer = nullptr; Alignment other_strand; bool flipped = false;   // a request that runs on the other strand: `alignment` is other_strand, `answer` the caller's
    std::unique_ptr<DagifiedLocalGraph> d; Route route = Route::SOFTCLIP; size_t band_padding = 0;
};

ChainConnector::ChainConnector(const Aligner& aligner, const HandleGraph& graph, size_t max_dp_cells, BandPaddingFunction choose_band_padding)
    : aligner_(aligner), graph_(graph), max_dp_cells_(max_dp_cells), choose_band_padding_(std::move(choose_band_padding)) {}
ChainConnector::~ChainConnector() = default;

size_t ChainConnector::add(const Position& left_anchor, const Position& right_anchor, size_t max_path_length, size_t max_gap_length, Alignment& alignment,
                           bool consistently) {
    auto r = std::make_unique<Request>();
    r->left = left_anchor; r->right = right_anchor; r->max_path_length = max_path_length; r->max_gap_length = max_gap_length; r->alignment = &alignment;
    if (consistently) {
        auto node_length = [&](nid_t id) -> int64_t { return (int64_t)graph_.get_length(graph_.get_handle(id)); };
        r->other_strand = reverse_complement_alignment(alignment, node_length);
        if (runs_flipped(left_anchor, right_anchor, alignment, r->other_strand)) {
            r->flipped = true; r->answer = &alignment; r->alignment = &r->other_strand;
            r->left = turned(right_anchor, &graph_); r->right = turned(left_anchor, &graph_);
        } else r->other_strand = Alignment();
    }
    requests_.push_back(std::move(r));
    outcomes_.emplace_back();
    return requests_.size() - 1;
}

void ChainConnector::run(unsigned threads) {
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t) { return std::chrono::duration<double, std::milli>(clock::now() - t).count(); };
    const size_t first = answered_;                                             // requests added since the last run
    answered_ = requests_.size();
    if (!threads) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(requests_.size() - first, 1));
    auto on_threads = [&](const std::function<void(size_t)>& body) {
        std::atomic<size_t> next{first};
        auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < requests_.size();) body(i); };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    };
    // 1. the local graphs, on the host threads; each thread also submits its problems (the host half of a submission — topological order,
    //    packing — runs on the submitting thread, outside the batch's lock)
    auto t0 = clock::now();
    AlignmentBatch batch(aligner_);
    batch.isolate_failures = true;
    on_threads([&](size_t i) {
        Request& r = *requests_[i]; Outcome& o = outcomes_[i];
        o = Outcome{};
        try {
            r.d = std::make_unique<DagifiedLocalGraph>(r.left, r.right, r.max_path_length, graph_);
            o.trims = r.d->trim_tips();
            r.route = choose_route(r.left, r.right, *r.d, *r.alignment, max_dp_cells_, nullptr);
            if (r.route == Route::SOFTCLIP) { o.status = TOO_BIG; r.d.reset(); }
            else if (r.route == Route::BANDED) {
                r.band_padding = choose_band_padding_(*r.alignment, r.d->dagified);
                batch.align_global_banded(*r.alignment, r.d->dagified, (int32_t)r.band_padding, true, max_dp_cells_);
            } else batch.align_pinned(*r.alignment, r.d->dagified, !is_empty(r.left), true, as_gap_limit(r.max_gap_length));
        } catch (ChainAlignmentFailedError& e) { o.status = NO_GRAPH; o.message = e.what(); r.d.reset(); }
        catch (BandMatricesTooBigException& e) { r.alignment->path.mapping.clear(); o.status = TOO_BIG; o.message = e.what(); o.did_align = true; r.d.reset(); }   // (refused while it was being prepared)
        catch (std::exception& e) { o.status = FAILED; o.message = e.what(); r.d.reset(); }
    });
    last_extract_ms = ms_since(t0);
    // 2. every DP problem in one flush: one launch per kernel family
    t0 = clock::now();
    batch.flush();
    last_align_ms = ms_since(t0);
    // 3. back into the base graph
    t0 = clock::now();
    on_threads([&](size_t i) {
        Request& r = *requests_[i]; Outcome& o = outcomes_[i];
        if (!r.d) return;
        try {
            if (std::exception_ptr failed = batch.failure_of(*r.alignment)) std::rethrow_exception(failed);
            translate_back(r.left, r.right, *r.d, &graph_, *r.alignment);
            o.status = ALIGNED; o.did_align = true;
        } catch (BandMatricesTooBigException& e) {
            r.alignment->path.mapping.clear(); o.status = TOO_BIG; o.message = e.what(); o.did_align = true;      // (the direct call returns true here as well)
        } catch (NoAlignmentInBandException& e) { o.status = NO_ALIGNMENT_IN_BAND; o.message = e.what(); }
        catch (std::exception& e) { o.status = FAILED; o.message = e.what(); }
        r.d.reset();
    });
    on_threads([&](size_t i) {                                                    // answers found on the other strand are turned back
        Request& r = *requests_[i];
        if (!r.flipped) return;
        auto node_length = [&](nid_t id) -> int64_t { return (int64_t)graph_.get_length(graph_.get_handle(id)); };
        *r.answer = reverse_complement_alignment(r.other_strand, node_length);
        try { check_one_piece(*r.answer); } catch (std::exception& e) { outcomes_[i].status = FAILED; outcomes_[i].message = e.what(); }
    });
    last_translate_ms = ms_since(t0);
}

}  // namespace vgamd
