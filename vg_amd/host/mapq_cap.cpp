// mapq_cap.cpp — see mapq_cap.hpp.
#include "mapq_cap.hpp"
#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <limits>
#include <stdexcept>

namespace vgamd {

namespace {
constexpr size_t MAX_AT_LEAST_ONE_EVENTS = 32, AT_LEAST_ONE_PRECISION = 8;      // src/statistics.hpp:176-178

// One interval [left, right) of read bases over which the minimizers with ranks [bottom, top) (in explored order) all have their
// agglomerations: the sweep of for_each_agglomeration_interval.
void for_each_interval(const std::vector<CapMinimizer>& minimizers, size_t read_length, const std::vector<size_t>& order,
                       const std::function<void(size_t, size_t, size_t, size_t)>& emit) {
    if (order.empty()) return;
    std::deque<const CapMinimizer*> open{&minimizers[order.front()]};
    size_t left = open.front()->agglomeration_start, bottom = 0;
    auto emit_before = [&](size_t right) {
        while (left < right) {
            const size_t first_end = open.front()->agglomeration_start + open.front()->agglomeration_length;
            if (first_end <= right) {
                if (first_end < left) throw std::runtime_error("faster_cap: minimizers not sorted properly");
                emit(left, first_end, bottom, bottom + open.size());
                left = open.size() == 1 ? right : first_end;           // a single open item: a gap follows it
                bottom += 1;
                open.pop_front();
            } else {
                emit(left, right, bottom, bottom + open.size());
                left = right;
            }
        }
    };
    for (size_t k = 1; k < order.size(); ++k) {
        if (open.empty()) throw std::runtime_error("faster_cap: minimizers not stacked up properly");
        emit_before(minimizers[order[k]].agglomeration_start);
        open.push_back(&minimizers[order[k]]);
    }
    emit_before(read_length);
}

// the probability that an error at read base `index` disrupts every one of the minimizers order[bottom, top)
double disruption_in_column(const std::vector<CapMinimizer>& minimizers, const std::string& quality, const std::vector<size_t>& order,
                            size_t bottom, size_t top, size_t index) {
    double p = phred_to_prob((uint8_t)quality[index]);
    for (size_t k = bottom; k < top; ++k) {
        const CapMinimizer& m = minimizers[order[k]];
        if (m.forward_offset() <= index && index < m.forward_offset() + (size_t)m.length) continue;      // inside the k-mer: the error itself disrupts it
        // in the flank: the error must create a k-mer that beats this one — at most one new k-mer per base of a k-mer, per base from the
        // agglomeration's start to here, per base from here to its end
        const size_t possible = std::min((size_t)m.length, std::min(index - m.agglomeration_start + 1, (m.agglomeration_start + m.agglomeration_length) - index));
        p *= prob_for_at_least_one(m.hash, possible);
    }
    return p;
}
}  // namespace

double phred_to_prob(uint8_t phred) { return std::pow(10.0, -(double)phred / 10.0); }

double prob_for_at_least_one(uint64_t p, size_t n) {
    static const std::vector<double> table = [] {
        const size_t values = (size_t)1 << AT_LEAST_ONE_PRECISION;
        std::vector<double> t((MAX_AT_LEAST_ONE_EVENTS + 1) * values, 0.0);
        for (size_t k = 1; k <= MAX_AT_LEAST_ONE_EVENTS; ++k)
            for (size_t q = 0; q < values; ++q) t[(k << AT_LEAST_ONE_PRECISION) + q] = 1.0 - std::pow(1.0 - (2 * q + 1) / (2.0 * values), (double)k);      // the middle of the bucket
        return t;
    }();
    if (n > MAX_AT_LEAST_ONE_EVENTS) throw std::runtime_error("prob_for_at_least_one: too many events");
    return table[(n << AT_LEAST_ONE_PRECISION) + (size_t)(p >> (64 - AT_LEAST_ONE_PRECISION))];
}

double faster_cap(const std::vector<CapMinimizer>& minimizers, std::vector<size_t>& explored, const std::string& sequence, const std::string& quality) {
    if (quality.empty()) return std::numeric_limits<double>::infinity();
    // by agglomeration end, then start (:2957-2962)
    std::sort(explored.begin(), explored.end(), [&](size_t a, size_t b) {
        const size_t ae = minimizers[a].agglomeration_start + minimizers[a].agglomeration_length, be = minimizers[b].agglomeration_start + minimizers[b].agglomeration_length;
        return ae < be || (ae == be && minimizers[a].agglomeration_start < minimizers[b].agglomeration_start);
    });
    for (size_t i : explored) if (minimizers[i].length == 0) throw std::runtime_error("faster_cap: minimizer with no sequence");
    // c[i + 1] = log10 probability that minimizers 0 .. i were all created by errors (:2996-3000)
    std::vector<double> c(explored.size() + 1, -std::numeric_limits<double>::infinity());
    c[0] = 0.0;
    for_each_interval(minimizers, sequence.size(), explored, [&](size_t left, size_t right, size_t bottom, size_t top) {
        double p_here = 0.0;                                     // a 0-length interval needs no disruption
        if (left != right) {
            double p = disruption_in_column(minimizers, quality, explored, bottom, top, left);
            for (size_t i = left + 1; i < right; ++i) { const double q = disruption_in_column(minimizers, quality, explored, bottom, top, i); p = p + q - p * q; }      // OR, assuming independence
            p_here = std::log10(p);
        }
        if (std::isinf(p_here)) throw std::runtime_error("faster_cap: minimizers seem impossible to disrupt in a region");
        const double p = c[bottom] + p_here;
        for (size_t i = bottom + 1; i < top + 1; ++i) if (c[i] < p) c[i] = p;
    });
    if (std::isinf(c.back())) throw std::runtime_error("faster_cap: minimizers seem impossible to disrupt");
    return -c.back() * 10.0;
}

}  // namespace vgamd
