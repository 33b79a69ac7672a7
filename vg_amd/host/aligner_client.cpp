// aligner_client.cpp — see aligner_client.hpp.
#include "aligner_client.hpp"
#include <cstring>

namespace vgamd {

XdropAligner::XdropAligner(const int8_t* score_matrix, int8_t go, int8_t ge, std::shared_ptr<EngineApi> eng, int dev)
    : gap_open(go), gap_extension(ge), engine(std::move(eng)), device(dev), configured(true) { std::memcpy(matrix, score_matrix, 16); }
std::unique_ptr<Aligner> XdropAligner::make(int8_t bonus) const { return std::make_unique<Aligner>(matrix, gap_open, gap_extension, bonus, 0.5, engine, device); }
const Aligner& XdropAligner::with_bonus(int8_t bonus) {
    if (!configured) throw std::runtime_error("XdropAligner: default-constructed (no scores)");
    std::lock_guard<std::mutex> lk(mu);
    std::unique_ptr<Aligner>& a = by_bonus[bonus];
    if (!a) a = make(bonus);
    return *a;
}
void XdropAligner::align(Alignment& alignment, const HandleGraph& graph, const std::vector<handle_t>& order, const std::vector<MaximalExactMatch>& mems,
                         bool reverse_complemented, int8_t full_length_bonus, uint16_t max_gap_length) {
    with_bonus(full_length_bonus).align_xdrop(alignment, graph, order, mems, reverse_complemented, max_gap_length);
}
void XdropAligner::align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, int8_t full_length_bonus, uint16_t max_gap_length) {
    with_bonus(full_length_bonus).align_pinned(alignment, g, pin_left, true, max_gap_length);
}

QualAdjXdropAligner::QualAdjXdropAligner(const int8_t* score_matrix, int8_t go, int8_t ge, double gc, std::shared_ptr<EngineApi> eng, int dev)
    : XdropAligner(score_matrix, go, ge, std::move(eng), dev), gc_content(gc) {}
std::unique_ptr<Aligner> QualAdjXdropAligner::make(int8_t bonus) const { return std::make_unique<QualAdjAligner>(matrix, gap_open, gap_extension, bonus, gc_content, engine, device); }

AlignerClient::AlignerClient(double gc, std::shared_ptr<EngineApi> eng, int dev) : gc_content_estimate(gc), engine(std::move(eng)), device(dev) {
    // the default scoring parameters (src/aligner.cpp:1350-1358); not a virtual call, as the reference notes
    AlignerClient::set_alignment_scores(default_score_matrix, default_gap_open, default_gap_extension, default_full_length_bonus);
}
const GSSWAligner* AlignerClient::get_aligner(bool have_qualities) const {
    return (have_qualities && adjust_alignments_for_base_quality) ? (const GSSWAligner*)get_qual_adj_aligner() : (const GSSWAligner*)get_regular_aligner();
}
const QualAdjAligner* AlignerClient::get_qual_adj_aligner() const { if (!qual_adj_aligner) throw std::runtime_error("AlignerClient: no aligner"); return qual_adj_aligner.get(); }
const Aligner* AlignerClient::get_regular_aligner() const { if (!regular_aligner) throw std::runtime_error("AlignerClient: no aligner"); return regular_aligner.get(); }
std::vector<int8_t> AlignerClient::parse_matrix(std::istream& in) {
    std::vector<int8_t> m(16);
    for (size_t i = 0; i < 16; ++i) {
        if (!in.good()) throw std::runtime_error("error: vg Aligner::parse_matrix requires a 4x4 whitespace separated integer matrix");
        int score = 0;
        in >> score;
        if (in.fail()) throw std::runtime_error("error: vg Aligner::parse_matrix requires a 4x4 whitespace separated integer matrix");
        if (score > 127 || score < -127) throw std::runtime_error("error: vg Aligner::parse_matrix requires values in the range [-127,127]");
        m[i] = (int8_t)score;
    }
    return m;
}
void AlignerClient::set_alignment_scores(int8_t match, int8_t mismatch, int8_t go, int8_t ge, int8_t bonus) {
    int8_t m[16];
    for (size_t i = 0; i < 16; ++i) m[i] = i % 5 == 0 ? match : (int8_t)-mismatch;      // matches on the diagonal
    this->set_alignment_scores(m, go, ge, bonus);
}
void AlignerClient::set_alignment_scores(const int8_t* m, int8_t go, int8_t ge, int8_t bonus) {
    qual_adj_aligner = std::make_unique<QualAdjAligner>(m, go, ge, bonus, gc_content_estimate, engine, device);
    regular_aligner = std::make_unique<Aligner>(m, go, ge, bonus, gc_content_estimate, engine, device);
}
void AlignerClient::set_alignment_scores(std::istream& in, int8_t go, int8_t ge, int8_t bonus) {
    const std::vector<int8_t> m = parse_matrix(in);
    this->set_alignment_scores(m.data(), go, ge, bonus);
}

}  // namespace vgamd
