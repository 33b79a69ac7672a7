// mapq_cap.hpp — giraffe's cap on a read's mapping quality from the minimizers that were explored: the probability that sequencing errors
// created every one of them (MinimizerMapper::faster_cap, reference src/minimizer_mapper.cpp:2946-3090, with
// for_each_agglomeration_interval :3092-3161, get_log10_prob_of_disruption_in_interval :3163-3201, get_prob_of_disruption_in_column
// :3203-3260 and prob_for_at_least_one, src/statistics.cpp:525-560).  Host arithmetic beside the seeding stage; held to
// the reference's two unit tests (src/unittest/minimizer_mapper.cpp:154-252) by tests/test_mapq_cap.py.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace vgamd {

// what faster_cap reads of MinimizerMapper::Minimizer (src/minimizer_mapper.hpp:565-600)
struct CapMinimizer {
    uint64_t hash = 0;                  // value.hash
    size_t offset = 0;                  // value.offset: first base of the k-mer, or its last when is_reverse
    bool is_reverse = false;
    size_t agglomeration_start = 0, agglomeration_length = 0;
    int32_t length = 0;                 // k
    size_t forward_offset() const { return is_reverse ? offset - (size_t)(length - 1) : offset; }
};

// An approximate probability of at least one of n <= 32 events of probability p / 2^64 each (the table of src/statistics.cpp:525-560)
double prob_for_at_least_one(uint64_t p, size_t n);
double phred_to_prob(uint8_t phred);

// -> Phred; +infinity without base qualities.  `minimizers_explored` is sorted in place as the reference sorts it.
// Throws std::runtime_error where the reference prints an error and exits.
double faster_cap(const std::vector<CapMinimizer>& minimizers, std::vector<size_t>& minimizers_explored, const std::string& sequence, const std::string& quality_bytes);

}  // namespace vgamd
