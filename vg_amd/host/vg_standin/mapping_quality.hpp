// STAND-IN (vg_amd/host/vg_standin/): restates a slice of vg that stays vg's own in a real integration; present only so that the
// GSSWAligner mirror has the members callers read (src/aligner.hpp:142-148) and the reference's unit tests for them can be driven
// without vg.  Excluded from size / originality claims.
// mapping_quality.hpp — MappingQualityCalculator (reference: src/mapping_quality_calculator.hpp:25-128, .cpp:26-139): mapping qualities
// from a vector of scaled alignment scores; pure host arithmetic, untouched by the engine.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace vgamd {

class MappingQualityCalculator {
public:
    MappingQualityCalculator(double match, double mismatch, double log_base) : rep_match(match), rep_mismatch(mismatch), log_base(log_base) {}
    int32_t compute_max_mapping_quality(const std::vector<double>& scores, bool fast_approximation, const std::vector<double>* multiplicities = nullptr) const;
    int32_t compute_first_mapping_quality(const std::vector<double>& scores, bool fast_approximation, const std::vector<double>* multiplicities = nullptr) const;
    double mapping_quality_score_diff(double mapping_quality) const;
    static double maximum_mapping_quality_exact(const std::vector<double>& scaled_scores, size_t* max_idx_out, const std::vector<double>* multiplicities = nullptr);
    static double maximum_mapping_quality_approx(const std::vector<double>& scaled_scores, size_t* max_idx_out, const std::vector<double>* multiplicities = nullptr);
    static double first_mapping_quality_exact(const std::vector<double>& s, const std::vector<double>* m = nullptr) { return maximum_mapping_quality_exact(s, nullptr, m); }
    static double first_mapping_quality_approx(const std::vector<double>& s, const std::vector<double>* m = nullptr) { return maximum_mapping_quality_approx(s, nullptr, m); }
    double rep_match, rep_mismatch, log_base;
};

}  // namespace vgamd
