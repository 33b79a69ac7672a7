// STAND-IN (vg_amd/host/vg_standin/): see mapping_quality.hpp.  Follows the reference's formulas (src/mapping_quality_calculator.cpp:26-139,
// src/statistics.hpp:110-120); excluded from size / originality claims.
#include "mapping_quality.hpp"
#include <algorithm>
#include <cmath>
#include <limits>

namespace vgamd {

namespace {
const double quality_scale_factor = 10.0 / std::log(10.0);
double add_log(double x, double y) { return x > y ? x + std::log1p(std::exp(y - x)) : y + std::log1p(std::exp(x - y)); }
double subtract_log(double x, double y) { return x + std::log1p(-std::exp(y - x)); }
}  // namespace

double MappingQualityCalculator::maximum_mapping_quality_exact(const std::vector<double>& scaled_scores, size_t* max_idx_out, const std::vector<double>* multiplicities) {
    double log_sum_exp = std::numeric_limits<double>::lowest(), to_score = std::numeric_limits<double>::lowest();
    for (int64_t i = (int64_t)scaled_scores.size() - 1; i >= 0; --i) {          // backwards: kinder to sorted scores, and ties go to the earlier item
        double score = scaled_scores[(size_t)i];
        if (max_idx_out && score >= to_score) { *max_idx_out = (size_t)i; to_score = score; }
        if (multiplicities && (*multiplicities)[(size_t)i] > 1.0) score += std::log((*multiplicities)[(size_t)i]);
        log_sum_exp = add_log(log_sum_exp, score);
    }
    if (scaled_scores.size() == 1 && (!multiplicities || (*multiplicities)[0] <= 1.0)) log_sum_exp = add_log(log_sum_exp, 0.0);      // the null alignment
    if (!max_idx_out) to_score = scaled_scores.empty() ? 0.0 : scaled_scores.front();
    const double direct_mapq = -quality_scale_factor * subtract_log(0.0, to_score - log_sum_exp);
    return std::isinf(direct_mapq) ? (double)std::numeric_limits<int32_t>::max() : direct_mapq;
}

double MappingQualityCalculator::maximum_mapping_quality_approx(const std::vector<double>& scaled_scores, size_t* max_idx_out, const std::vector<double>* multiplicities) {
    double max_score = scaled_scores.at(0); size_t max_idx = 0;
    double next_score = 0.0, next_count = 1.0;                                  // the null alignment to begin with
    auto mult = [&](size_t i) { return multiplicities ? (*multiplicities)[i] : 1.0; };
    if (multiplicities && mult(0) > 1.0) { next_score = max_score; next_count = mult(0) - 1.0; }
    for (size_t i = 1; i < scaled_scores.size(); ++i) {
        const double score = scaled_scores[i];
        if (score > max_score) {
            if (multiplicities && mult(i) > 1.0) { next_score = score; next_count = mult(i) - 1.0; }
            else if (next_score == max_score) next_count += 1.0;
            else { next_score = max_score; next_count = mult(max_idx); }
            max_score = score; max_idx = i;
        } else if (score > next_score) { next_score = score; next_count = mult(i); }
        else if (score == next_score) next_count += mult(i);
    }
    if (max_idx_out) *max_idx_out = max_idx;
    if (max_idx_out || max_idx == 0) return std::max(0.0, quality_scale_factor * (max_score - next_score - (next_count > 1.0 ? std::log(next_count) : 0.0)));
    return maximum_mapping_quality_exact(scaled_scores, nullptr, multiplicities);
}

int32_t MappingQualityCalculator::compute_max_mapping_quality(const std::vector<double>& scores, bool fast_approximation, const std::vector<double>* multiplicities) const {
    std::vector<double> scaled(scores.size());
    for (size_t i = 0; i < scores.size(); ++i) scaled[i] = log_base * scores[i];
    size_t idx;
    return (int32_t)(fast_approximation ? maximum_mapping_quality_approx(scaled, &idx, multiplicities) : maximum_mapping_quality_exact(scaled, &idx, multiplicities));
}
int32_t MappingQualityCalculator::compute_first_mapping_quality(const std::vector<double>& scores, bool fast_approximation, const std::vector<double>* multiplicities) const {
    std::vector<double> scaled(scores.size());
    for (size_t i = 0; i < scores.size(); ++i) scaled[i] = log_base * scores[i];
    return (int32_t)(fast_approximation ? first_mapping_quality_approx(scaled, multiplicities) : first_mapping_quality_exact(scaled, multiplicities));
}
double MappingQualityCalculator::mapping_quality_score_diff(double mapping_quality) const { return mapping_quality / (quality_scale_factor * log_base); }

}  // namespace vgamd
