// seed_policy.hpp — which of a read's minimizers become seeds: MinimizerMapper::find_seeds' filters (reference
// src/minimizer_mapper.cpp:4109-4470) with the scores find_minimizers gives them (:3918-3941), the order sort_minimizers_by_score puts them
// in (:4074-4107) and the window downsampling of algorithms::sample_minimal (src/algorithms/sample_minimal.cpp:21-207).  Host logic beside
// the seeding stage: the engine's vgk_minimizer_seeds finds every minimizer and its hits; these rules say which hits are
// looked at.
//
// Pinned: sample_minimal, by the reference's six unit tests (src/unittest/sample_minimal.cpp:14-176; tests/golden/ref_sample_minimal.json).
// [PARITY-UNPINNED] the filter chain — the reference holds no test for find_seeds; tests/test_seed_policy.py holds it to a direct
// restatement of the cited lines.  Equal scores are ordered by key (Minimizer::operator<, src/minimizer_mapper.hpp:577); the runs that share
// the BEST score are shuffled as sort_shuffling_ties shuffles them (src/utility.hpp:771-799, :720-727: Knuth's shuffle over std::minstd_rand,
// which LazyRNG seeds from the read's sequence, src/utility.cpp:911-927) when the read's sequence is given — the single-end rule
// (src/minimizer_mapper.cpp:620-627); the paired path seeds one generator from both mates and carries it from the first mate's sort to the
// second's (:1529-1541): a caller that maps pairs keeps a ReadRng over mate 1 + mate 2 and hands it to both calls.  Inside a run
// of one key the order (std::sort's, unspecified) cannot matter: its minimizers share hits and score, and pass or fail together.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace vgamd {

// every element minimal in some window of `window_size` bases of a sequence: elements sorted by start, all of one length; should_beat(a, b):
// a displaces b.  sample(i) at least once for every such element (ties at one start are all sampled; of ties at different starts the
// earliest, the others when they come to the front).
void sample_minimal(size_t count, size_t element_length, size_t window_size, size_t sequence_length, const std::function<size_t(size_t)>& get_start,
                    const std::function<bool(size_t, size_t)>& should_beat, const std::function<void(size_t)>& sample);

struct PolicyMinimizer {                 // what find_seeds reads of MinimizerMapper::Minimizer (src/minimizer_mapper.hpp:540-600)
    uint64_t key = 0;                    // value.key
    size_t forward_offset = 0;           // first read base of the k-mer
    size_t length = 0;                   // k
    size_t hits = 0;                     // occurrences in the index
    double score = 0;                    // filled by score_minimizers
};
struct SeedPolicy {                      // MinimizerMapper's parameters of the same names (src/minimizer_mapper.hpp:140-260), giraffe's defaults
    size_t hit_cap = 10, hard_hit_cap = 500;
    double minimizer_score_fraction = 0.9;
    size_t max_unique_min = 500, num_bp_per_min = 1000;
    bool exclude_overlapping_min = false;
    size_t minimizer_coverage_flank = 250;
    size_t minimizer_downsampling_window_count = 0, minimizer_downsampling_max_window_length = (size_t)-1;
};
enum SeedFilter : uint8_t { SEED_TAKEN = 0, SEED_DOWNSAMPLED = 1, SEED_NO_HITS = 2, SEED_HARD_HIT_CAP = 3, SEED_OVERLAPPING = 4, SEED_MAX_MIN = 5, SEED_HIT_CAP = 6 };

// find_minimizers' score per minimizer (:3927-3937): 1 + ln(hard_hit_cap) - ln(hits), 1 beyond the hard cap, 0 without hits
void score_minimizers(std::vector<PolicyMinimizer>& minimizers_in_read_order, size_t hard_hit_cap);
// LazyRNG (src/utility.hpp:693-713, src/utility.cpp:907-927): std::minstd_rand, seeded at its first use from a string — the read's sequence in the
// single-end path (:620), both mates' sequences in a row in the paired path (:1529), where ONE generator serves the first mate's sort and then the second's.
class ReadRng {
public:
    explicit ReadRng(std::string seed_sequence) : seed_(std::move(seed_sequence)) {}
    uint32_t operator()();               // the next number; the generator is made at the first call
    bool started() const { return started_; }
private:
    std::string seed_; bool started_ = false; uint32_t state_ = 1;
};
// sort_minimizers_by_score (:4074-4107): runs of one key together, the runs by descending score (ties: header) -> indices in that order
std::vector<size_t> minimizers_by_score(const std::vector<PolicyMinimizer>& minimizers_in_read_order, const std::string* sequence = nullptr);
// the same drawing from a generator the caller keeps (the paired path: ReadRng rng(mate1 + mate2); the first mate's call, then the second's)
std::vector<size_t> minimizers_by_score(const std::vector<PolicyMinimizer>& minimizers_in_read_order, ReadRng& rng);
// find_seeds' selection (:4109-4440): per minimizer (read order) the filter it failed, or SEED_TAKEN — the hits of the taken ones are the seeds.
// Throws std::runtime_error where the reference crashes (a minimizer longer than the downsampling window).
// sequence (nullable): the read — seeds the shuffle of the runs tied at the top; without it they stay in key order
std::vector<uint8_t> select_minimizers(const std::vector<PolicyMinimizer>& minimizers_in_read_order, size_t read_length, const SeedPolicy& policy,
                                       const std::string* sequence = nullptr);
std::vector<uint8_t> select_minimizers(const std::vector<PolicyMinimizer>& minimizers_in_read_order, size_t read_length, const SeedPolicy& policy, ReadRng& rng);
// the filters over an order made before (minimizers_by_score)
std::vector<uint8_t> select_minimizers_in_order(const std::vector<PolicyMinimizer>& minimizers_in_read_order, size_t read_length, const SeedPolicy& policy, const std::vector<size_t>& order);

// MinimizerMapper::score_cluster (src/minimizer_mapper.cpp:4738-4781): a cluster's score is the sum of the scores of the DISTINCT minimizers its seeds come
// from (`present`), its coverage the fraction of the read's bases that those minimizers' k-mers cover.  seed_sources[i] = the minimizer (index into
// minimizers_in_read_order) seed i of the cluster came from (Seed::source).  [PARITY-UNPINNED: the reference holds no test for it; tests/test_seed_policy.py
// holds it to the cited lines restated.  Who makes the clusters — SnarlDistanceIndexClusterer — is outside the snapshot.]
struct ClusterScore { double score = 0.0, coverage = 0.0; std::vector<uint8_t> present; };
ClusterScore score_cluster(const std::vector<size_t>& seed_sources, const std::vector<PolicyMinimizer>& minimizers_in_read_order, size_t seq_length);

}  // namespace vgamd
