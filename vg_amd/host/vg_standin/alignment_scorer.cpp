// alignment_scorer.cpp — STAND-IN for code that stays vg's own in a real integration (see the note in gbwt_types.cpp): the
// scorer tables (MatrixAlignmentScorer, QualAdjAlignmentScorer: src/alignment_scorer.cpp:30-99, 264-314, 419-513) and
// GSSWAligner::identify_pinning_points (src/aligner.cpp:87-118).  The floating-point code must round exactly as vg's does to
// produce the same int8 tables, so it follows the reference formula by formula.  Not part of the engine; excluded from any size
// or originality claim.
#include "../aligner.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace vgamd {

std::string nonATGCNtoN(const std::string& s) {
    std::string n = s;
    for (auto& b : n) if (b != 'A' && b != 'T' && b != 'G' && b != 'C' && b != 'N') b = 'N';
    return n;
}

MatrixAlignmentScorer::MatrixAlignmentScorer(const int8_t* m4, int8_t go, int8_t ge, int8_t bonus)
    : match(m4[0]), mismatch((int8_t)-m4[1]), gap_open(go), gap_extension(ge), full_length_bonus(bonus) {
    for (size_t i = 0, j = 0; i < 25; ++i) {
        if (i % 5 == 4 || i / 5 == 4) score_matrix[i] = 0;
        else score_matrix[i] = m4[j++];
    }
}

size_t MatrixAlignmentScorer::longest_detectable_gap(size_t read_length, size_t read_pos) const {
    int64_t overhang_length = (int64_t)std::min(read_pos, read_length - read_pos);
    int64_t numer = (int64_t)match * overhang_length + full_length_bonus;
    int64_t gap_length = (numer - gap_open) / gap_extension + 1;
    return gap_length >= 0 && overhang_length > 0 ? (size_t)gap_length : 0;
}

int32_t MatrixAlignmentScorer::score_contiguous_alignment(const Alignment& aln, bool allow_left_bonus, bool allow_right_bonus) const {
    int32_t score = 0;
    bool in_deletion = false;
    const auto& maps = aln.path.mapping;
    for (size_t i = 0; i < maps.size(); ++i) for (size_t j = 0; j < maps[i].edit.size(); ++j) {
        const Edit& e = maps[i].edit[j];
        const bool at_an_end = (i == 0 && j == 0) || (i + 1 == maps.size() && j + 1 == maps[i].edit.size());
        if (edit_is_match(e)) { score += match * e.to_length; in_deletion = false; }
        else if (edit_is_sub(e)) { score -= mismatch * e.to_length; in_deletion = false; }
        else if (e.from_length > 0 && e.to_length == 0) { score -= in_deletion ? e.from_length * gap_extension : gap_open + (e.from_length - 1) * gap_extension; in_deletion = true; }
        else if (e.from_length == 0 && e.to_length == 0) { /* an empty edit changes nothing, not even whether a deletion is running */ }
        else if (edit_is_insertion(e) && !at_an_end) { score -= gap_open + (e.to_length - 1) * gap_extension; in_deletion = false; }
        else in_deletion = false;                                              // a soft clip
    }
    auto clipped = [&](bool left) {
        if (maps.empty()) return false;
        const Mapping& m = left ? maps.front() : maps.back();
        if (m.edit.empty()) return false;
        return edit_is_insertion(left ? m.edit.front() : m.edit.back());
    };
    if (allow_left_bonus && !clipped(true)) score += full_length_bonus;
    if (allow_right_bonus && !clipped(false)) score += full_length_bonus;
    return score;
}

vgk_scoring MatrixAlignmentScorer::as_vgk() const {
    vgk_scoring s{};
    for (int i = 0; i < 25; ++i) s.matrix[i] = score_matrix[i];
    s.gap_open = (uint8_t)gap_open; s.gap_extend = (uint8_t)gap_extension; s.full_length_bonus = full_length_bonus;
    return s;
}

double QualAdjAlignmentScorer::recover_log_base(const double matrix[16], double gc_content, double tol) {
    double nt_freqs[4] = {0.5 * (1 - gc_content), 0.5 * gc_content, 0.5 * gc_content, 0.5 * (1 - gc_content)};
    auto partition = [&](double lambda) {
        double p = 0.0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) p += nt_freqs[i] * nt_freqs[j] * std::exp(lambda * matrix[i * 4 + j]);
        return p;
    };
    // verify_valid_log_odds_score_matrix (:101-117)
    bool positive = false; double expected = 0.0;
    for (int i = 0; i < 16; ++i) positive = positive || matrix[i] > 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) expected += nt_freqs[i] * nt_freqs[j] * matrix[i * 4 + j];
    if (!positive || !(expected < 0.0))
        throw std::invalid_argument("error:[AlignmentScorer] Score matrix is invalid. Must have a negative expected score against random sequence.");
    double lower_bound, upper_bound, lambda = 1.0;
    double part = partition(lambda);
    if (part < 1.0) {
        lower_bound = lambda;
        while (part <= 1.0) { lower_bound = lambda; lambda *= 2.0; part = partition(lambda); }
        upper_bound = lambda;
    } else {
        upper_bound = lambda;
        while (part >= 1.0) { upper_bound = lambda; lambda /= 2.0; part = partition(lambda); }
        lower_bound = lambda;
    }
    while (upper_bound / lower_bound - 1.0 > tol) {
        lambda = 0.5 * (lower_bound + upper_bound);
        if (partition(lambda) < 1.0) lower_bound = lambda; else upper_bound = lambda;
    }
    return 0.5 * (lower_bound + upper_bound);
}

QualAdjAlignmentScorer::QualAdjAlignmentScorer(const int8_t* m4, int8_t go, int8_t ge, int8_t bonus, double gc_content)
    : MatrixAlignmentScorer(m4, go, ge, bonus) {
    constexpr uint32_t max_qual = 255;
    double dm[16];
    for (int i = 0; i < 16; ++i) dm[i] = (double)m4[i];
    log_base = recover_log_base(dm, gc_content);
    double nt_freqs[4] = {0.5 * (1 - gc_content), 0.5 * gc_content, 0.5 * gc_content, 0.5 * (1 - gc_content)};
    // qual_adjusted_matrix (:438-492)
    double align_prob[16], align_complement_prob[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) align_prob[i * 4 + j] = std::exp(log_base * m4[i * 4 + j]) * nt_freqs[i] * nt_freqs[j];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        align_complement_prob[i * 4 + j] = 0.0;
        for (int k = 0; k < 4; ++k) if (k != j) align_complement_prob[i * 4 + j] += align_prob[i * 4 + k];
    }
    int lowest_meaningful_qual = (int)std::ceil(-10.0 * std::log10(0.75));
    qual_adj_matrix.assign(25 * (max_qual + 1), 0);
    for (uint32_t q = 0; q <= max_qual; ++q) {
        double err = std::pow(10.0, -((double)q) / 10.0);
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) {
            int8_t score;
            if (i == 4 || j == 4 || (int)q < lowest_meaningful_qual) score = 0;
            else score = (int8_t)std::round(std::log(((1.0 - err) * align_prob[i * 4 + j] + (err / 3.0) * align_complement_prob[i * 4 + j])
                                                   / (nt_freqs[i] * ((1.0 - err) * nt_freqs[j] + (err / 3.0) * (1.0 - nt_freqs[j])))) / log_base);
            qual_adj_matrix[q * 25 + i * 5 + j] = score;
        }
    }
    // qual_adjusted_bonuses (:494-513)
    double p_full_len = std::exp(log_base * bonus) / (1.0 + std::exp(log_base * bonus));
    qual_adj_full_length_bonuses.assign(max_qual + 1, 0);
    ++lowest_meaningful_qual;      // the reference's "hack": Illumina's minimum quality 2 scores zero
    for (uint32_t q = (uint32_t)lowest_meaningful_qual; q <= max_qual; ++q) {
        double err = std::pow(10.0, -((double)q) / 10.0);
        double score = std::log(((1.0 - err * 4.0 / 3.0) * p_full_len + (err * 4.0 / 3.0) * (1.0 - p_full_len)) / (1.0 - p_full_len)) / log_base;
        qual_adj_full_length_bonuses[q] = (int8_t)std::round(score);
    }
}

std::unordered_set<nid_t> GSSWAligner::identify_pinning_points(const HandleGraph& graph) const {
    std::unordered_set<nid_t> return_val;
    for (const handle_t& handle : handlealgs::tail_nodes(&graph)) {
        std::vector<handle_t> stack(1, handle);
        while (!stack.empty()) {
            handle_t here = stack.back(); stack.pop_back();
            if (graph.get_length(here) > 0) return_val.insert(graph.get_id(here));
            else graph.follow_edges_v(here, true, [&](const handle_t& prev) {
                if (!return_val.count(graph.get_id(prev))) stack.push_back(prev);
            });
        }
    }
    return return_val;
}

}  // namespace vgamd
