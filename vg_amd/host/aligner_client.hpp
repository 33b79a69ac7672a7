// aligner_client.hpp — the remaining class shapes of the reference's aligner interface, thin over what aligner.hpp holds:
//   * XdropAligner / QualAdjXdropAligner (reference src/dozeu_interface.hpp:259-330, src/xdrop_aligner.cpp, src/qual_adj_xdrop_aligner.cpp):
//     dozeu behind a (score matrix, gap open, gap extension) constructor, the full-length bonus an argument of every call.  vg's Aligner
//     owns one of these and forwards its X-drop calls (src/aligner.cpp:628-682, :833-855); here the engine is reached through an
//     Aligner, so these classes keep one engine context per bonus they are asked for and forward the other way;
//   * AlignerClient (src/aligner.hpp:266-316, src/aligner.cpp:1350-1440): the pair of aligners (plain, quality-adjusted) a mapper holds,
//     re-made whenever the scores are set; parse_matrix.
#pragma once
#include <istream>
#include <map>
#include <memory>
#include <mutex>
#include "aligner.hpp"

namespace vgamd {

class XdropAligner {
public:
    XdropAligner() = default;
    // a 4 x 4 score matrix (src/dozeu_interface.hpp:264-267)
    XdropAligner(const int8_t* score_matrix, int8_t gap_open, int8_t gap_extension, std::shared_ptr<EngineApi> engine = nullptr, int device = 0);
    virtual ~XdropAligner() = default;
    // DozeuInterface::align (src/dozeu_interface.hpp:101-103): the seeded two-pass alignment over `order`
    void align(Alignment& alignment, const HandleGraph& graph, const std::vector<handle_t>& order, const std::vector<MaximalExactMatch>& mems,
               bool reverse_complemented, int8_t full_length_bonus, uint16_t max_gap_length = default_xdrop_max_gap_length);
    // DozeuInterface::align_pinned (:114-115)
    void align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, int8_t full_length_bonus, uint16_t max_gap_length = default_xdrop_max_gap_length);
protected:
    virtual std::unique_ptr<Aligner> make(int8_t full_length_bonus) const;
    const Aligner& with_bonus(int8_t full_length_bonus);
    int8_t matrix[16] = {0}; int8_t gap_open = default_gap_open, gap_extension = default_gap_extension;
    std::shared_ptr<EngineApi> engine; int device = 0; bool configured = false;
private:
    std::mutex mu; std::map<int, std::unique_ptr<Aligner>> by_bonus;
};

class QualAdjXdropAligner : public XdropAligner {
public:
    QualAdjXdropAligner() = default;
    QualAdjXdropAligner(const int8_t* score_matrix, int8_t gap_open, int8_t gap_extension, double gc_content = 0.5, std::shared_ptr<EngineApi> engine = nullptr, int device = 0);
protected:
    std::unique_ptr<Aligner> make(int8_t full_length_bonus) const override;
    double gc_content = 0.5;
};

class AlignerClient {
public:
    virtual ~AlignerClient() = default;
    virtual void set_alignment_scores(int8_t match, int8_t mismatch, int8_t gap_open, int8_t gap_extend, int8_t full_length_bonus);
    virtual void set_alignment_scores(std::istream& matrix_stream, int8_t gap_open, int8_t gap_extend, int8_t full_length_bonus);
    virtual void set_alignment_scores(const int8_t* score_matrix, int8_t gap_open, int8_t gap_extend, int8_t full_length_bonus);
    // 16 scores, ACGT x ACGT, whitespace-separated, each in [-127, 127]; throws std::runtime_error where the reference prints and throws
    static std::vector<int8_t> parse_matrix(std::istream& matrix_stream);
    bool adjust_alignments_for_base_quality = false;
    // (public here — protected in the reference, whose users are subclasses — so that tests and non-deriving callers can reach them)
    const GSSWAligner* get_aligner(bool have_qualities = true) const;
    const QualAdjAligner* get_qual_adj_aligner() const;
    const Aligner* get_regular_aligner() const;
    AlignerClient(double gc_content_estimate = 0.5, std::shared_ptr<EngineApi> engine = nullptr, int device = 0);
protected:
    double gc_content_estimate;
private:
    std::unique_ptr<QualAdjAligner> qual_adj_aligner;
    std::unique_ptr<Aligner> regular_aligner;
    std::shared_ptr<EngineApi> engine; int device;
};

}  // namespace vgamd
