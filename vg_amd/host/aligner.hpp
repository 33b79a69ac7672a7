// aligner.hpp — host-side mirror of vg's aligner interface
// (reference: src/aligner.hpp:32-213, src/alignment_scorer.hpp:18-29) whose
// method bodies hand the DP to the MI355X engine through the C ABI in
// include/vgk.h instead of calling gssw / dozeu / BandedGlobalAligner on the CPU.
// Names, argument meaning and error behaviour follow the reference so that the
// parity tests read like src/unittest/{aligner,pinned_alignment}.cpp.
#pragma once
#include <limits>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "vg_standin/alignment.hpp"
#include "engine.hpp"
#include "vg_standin/handle_graph.hpp"
#include "vg_standin/mapping_quality.hpp"

namespace vgamd {

// default scoring parameters (reference: src/alignment_scorer.hpp:18-29)
static constexpr int8_t default_match = 1;
static constexpr int8_t default_mismatch = 4;
static constexpr int8_t default_score_matrix[16] = {
     default_match,    -default_mismatch, -default_mismatch, -default_mismatch,
    -default_mismatch,  default_match,    -default_mismatch, -default_mismatch,
    -default_mismatch, -default_mismatch,  default_match,    -default_mismatch,
    -default_mismatch, -default_mismatch, -default_mismatch,  default_match };
static constexpr int8_t default_gap_open = 6;
static constexpr int8_t default_gap_extension = 1;
static constexpr int8_t default_full_length_bonus = 5;
static constexpr uint16_t default_xdrop_max_gap_length = 40;

// MatrixAlignmentScorer (reference: src/alignment_scorer.cpp:284-314): 4x4 -> 5x5
// with an all-zero N row/column; plus the closed-form helpers callers read.
struct MatrixAlignmentScorer {
    int8_t score_matrix[25];
    int8_t match, mismatch, gap_open, gap_extension, full_length_bonus;
    MatrixAlignmentScorer(const int8_t* score_matrix_4x4, int8_t go, int8_t ge, int8_t bonus);
    virtual ~MatrixAlignmentScorer() = default;      // (an aligner owns its scorer through this type: a QualAdjAlignmentScorer's tables must go with it)
    // reference: src/alignment_scorer.cpp:264-271
    size_t longest_detectable_gap(size_t read_length, size_t read_pos) const;
    // re-score an alignment from its edits: matches, substitutions, gaps (a deletion that runs on across a node boundary opens once), the
    // full-length bonus at either end that is not soft-clipped (src/alignment_scorer.cpp:158-238)
    int32_t score_contiguous_alignment(const Alignment& aln, bool allow_left_bonus = true, bool allow_right_bonus = true) const;
    double log_base = 0.0;              // recovered from the matrix and the GC content when the aligner is built (src/alignment_scorer.cpp:30-99)
    double get_log_base() const { return log_base; }
    vgk_scoring as_vgk() const;
};

// MaximalExactMatch as the aligner sees it (reference: src/mem.hpp:25-65; fields read at
// src/dozeu_interface.cpp:91-110): a read interval and its graph hits (gcsa::Node = id, offset, strand).
struct MaximalExactMatch {
    size_t begin = 0, end = 0;                      // [begin, end) in the read
    struct Hit { nid_t id; size_t offset; bool is_reverse; };
    std::vector<Hit> nodes;
    size_t length() const { return end - begin; }
};

// QualAdjAlignmentScorer (reference: src/alignment_scorer.cpp:419-513): 5x5xQ table indexed
// 25*qual + 5*nt[ref] + nt[read] and the quality-adjusted full-length bonuses, from the recovered log base.
struct QualAdjAlignmentScorer : MatrixAlignmentScorer {
    std::vector<int8_t> qual_adj_matrix;               // [256][25]
    std::vector<int8_t> qual_adj_full_length_bonuses;  // [256]
    QualAdjAlignmentScorer(const int8_t* score_matrix_4x4, int8_t go, int8_t ge, int8_t bonus, double gc_content);
    static double recover_log_base(const double matrix[16], double gc_content, double tol = 1e-12);   // :30-99
};

// thrown by align_global_banded (reference: src/banded_global_aligner.hpp:31-43)
class NoAlignmentInBandException : public std::exception {
public:
    const char* what() const noexcept override { return "error:[BandedGlobalAligner] cannot align to graph within band, consider permissive banding"; }
};
class BandMatricesTooBigException : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// DeletionAligner (reference: src/deletion_aligner.cpp:18-25, :57-113, :201-218): the global alignment of an EMPTY read,
// i.e. the shortest source-to-sink walk scored as one deletion.  Pure graph bookkeeping, so it stays on the host.
class DeletionAligner {
public:
    DeletionAligner(int8_t gap_open, int8_t gap_extension) : gap_open(gap_open), gap_extension(gap_extension) {}
    void align(Alignment& aln, const HandleGraph& graph) const;
    void align_multi(Alignment& aln, std::vector<Alignment>& alt_alns, const HandleGraph& graph, int32_t max_alt_alns) const;
private:
    int8_t gap_open, gap_extension;
};

class BaseAligner {
public:
    virtual ~BaseAligner() = default;
    virtual void align(Alignment& alignment, const HandleGraph& g, bool traceback_aln) const = 0;
};

class GSSWAligner : public BaseAligner {
protected:
    GSSWAligner(std::unique_ptr<MatrixAlignmentScorer> owned_scorer, std::shared_ptr<EngineApi> engine, int device,
                const QualAdjAlignmentScorer* qual_adj = nullptr);
    bool qual_adjusted = false;        // quality-adjusted context: every problem carries Alignment.quality
    ~GSSWAligner() override;

    // what create_gssw_graph hands to gssw (src/aligner.cpp:30-85): nodes in
    // lazier_topological_order, sequences cleaned by nonATGCNtoN, forward edges as CSR
    struct PackedGraph {
        std::vector<handle_t> order;
        std::vector<uint32_t> node_len, pred_off, pred_idx;
        std::string seq;
        vgk_graph view() const;
    };
    PackedGraph create_packed_graph(const HandleGraph& g) const;
    PackedGraph create_packed_graph(const HandleGraph& g, const std::vector<handle_t>& topological_order, bool raw_sequence = false) const;
    std::unordered_set<nid_t> identify_pinning_points(const HandleGraph& graph) const;   // src/aligner.cpp:87-118

    // gssw_mapping_to_alignment (src/aligner.cpp:120-241) over the engine's op list
    void ops_to_alignment(const PackedGraph& pg, const HandleGraph& seq_source, const vgk_result& res,
                          const vgk_op* ops, Alignment& alignment) const;
    // BABuilder's Path (src/banded_global_aligner.cpp:102-205) over the banded engine's op list
    static void banded_ops_to_alignment(const std::vector<handle_t>& order, const HandleGraph& g, const vgk_result& res, const vgk_op* ops,
                                        Alignment& alignment);

public:
    GSSWAligner(const GSSWAligner&) = delete;
    GSSWAligner& operator=(const GSSWAligner&) = delete;

    virtual void align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, bool xdrop = false,
                              uint16_t xdrop_max_gap_length = default_xdrop_max_gap_length) const = 0;
    virtual void align_pinned_multi(Alignment& alignment, std::vector<Alignment>& alt_alignments, const HandleGraph& g,
                                    bool pin_left, int32_t max_alt_alns) const = 0;

    std::unique_ptr<MatrixAlignmentScorer> scorer;
    std::unique_ptr<MappingQualityCalculator> mapq_calc;     // (src/aligner.hpp:148) host arithmetic on score vectors; untouched by the engine
    // X-drop calls (align_pinned(..., xdrop = true), align_xdrop) with dozeu's band restated (vgk_xdrop_band_align) instead of the
    // exact extension that keeps every cell (the default; the two agree whenever the band contains the optimal path).  PARITY-UNPINNED.
    mutable bool xdrop_band = false;

    // the engine binding, for the extenders that are constructed around an Aligner (src/gbwt_extender.hpp:156)
    const EngineApi& engine_api() const { return *engine; }
    vgk_ctx* engine_context() const { return ctx; }

protected:
    std::shared_ptr<EngineApi> engine;
    vgk_ctx* ctx = nullptr;
};

class Aligner : public GSSWAligner {
public:
    Aligner(const int8_t* score_matrix = default_score_matrix,
            int8_t gap_open = default_gap_open,
            int8_t gap_extension = default_gap_extension,
            int8_t full_length_bonus = default_full_length_bonus,
            double gc_content = 0.5,
            std::shared_ptr<EngineApi> engine = nullptr,   // nullptr = the HIP product library
            int device = 0);
protected:
    Aligner(std::unique_ptr<MatrixAlignmentScorer> owned_scorer, std::shared_ptr<EngineApi> engine, int device,
            const QualAdjAlignmentScorer* qual_adj);
public:

    // local alignment, bonus at both ends (src/aligner.cpp:566-569)
    void align(Alignment& alignment, const HandleGraph& g, bool traceback_aln) const override;
    // local alignment over a caller-supplied order that may mix strands (src/aligner.cpp:571-626)
    void align(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& topological_order) const;
    void align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, bool xdrop = false,
                      uint16_t xdrop_max_gap_length = default_xdrop_max_gap_length) const override;
    void align_pinned_multi(Alignment& alignment, std::vector<Alignment>& alt_alignments, const HandleGraph& g,
                            bool pin_left, int32_t max_alt_alns) const override;
    // banded global alignment from any source to any sink (src/aligner.cpp:699-760, :1189-1248); throws
    // NoAlignmentInBandException / BandMatricesTooBigException like the reference
    void align_global_banded(Alignment& alignment, const HandleGraph& g, int32_t band_padding = 0, bool permissive_banding = true,
                             uint64_t max_cells = std::numeric_limits<uint64_t>::max()) const;
    // the k best global alignments, best first; the best also in `alignment` (src/aligner.cpp:763-831)
    void align_global_banded_multi(Alignment& alignment, std::vector<Alignment>& alt_alignments, const HandleGraph& g, int32_t max_alt_alns,
                                   int32_t band_padding = 0, bool permissive_banding = true,
                                   uint64_t max_cells = std::numeric_limits<uint64_t>::max()) const;
    // two-pass seeded X-drop alignment (src/aligner.cpp:833-855 -> DozeuInterface::align, src/dozeu_interface.cpp:608-685)
    void align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<MaximalExactMatch>& mems,
                     bool reverse_complemented, uint16_t max_gap_length = default_xdrop_max_gap_length) const;
    void align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order,
                     const std::vector<MaximalExactMatch>& mems, bool reverse_complemented,
                     uint16_t max_gap_length = default_xdrop_max_gap_length) const;

    // The same for many reads at once (what MinimizerMapper::attempt_rescue asks for read after read, src/minimizer_mapper.cpp:3385, :3426):
    // every request's first pass — the scan for a head, or the extension from the seed towards it — goes to the engine in ONE call, every
    // second pass (the traced extension from the head) in another; each alignment comes out as align_xdrop(..., order, ...) leaves it.
    struct XdropRequest {
        Alignment* alignment = nullptr; const HandleGraph* graph = nullptr; std::vector<handle_t> order;      // order empty: lazier_topological_order(graph)
        std::vector<MaximalExactMatch> mems; bool reverse_complemented = false; uint16_t max_gap_length = default_xdrop_max_gap_length;
    };
    void align_xdrop_many(std::vector<XdropRequest>& requests) const;

    // the two halves of an alignment call, for AlignmentBatch
    struct Job;
    std::unique_ptr<Job> prepare_job(Alignment& alignment, const HandleGraph& g, bool pinned, bool pin_left, bool traceback_aln) const;
    void finish_job(Job& job, vgk_result res, std::vector<vgk_op> ops, std::vector<Alignment>* multi_alignments, int32_t max_alt_alns) const;
    std::unique_ptr<Job> prepare_banded_job(Alignment& alignment, const HandleGraph& g, int32_t band_padding, bool permissive_banding, uint64_t max_cells) const;
    void finish_banded_job(Job& job, const vgk_result& res, const vgk_op* ops) const;
    // align_pinned(..., xdrop = true) split the same way: DozeuPinningOverlay + DozeuInterface::align_pinned up to the engine call
    // (src/aligner.cpp:628-682, src/dozeu_interface.cpp:724-766), then calculate_and_save_alignment + the id translation (:338-572)
    std::unique_ptr<Job> prepare_xdrop_job(Alignment& alignment, const HandleGraph& g, bool pin_left, uint16_t xdrop_max_gap_length) const;
    void finish_xdrop_job(Job& job, vgk_result res, std::vector<vgk_op> ops) const;

private:
    // one pinned X-drop extension from an interior graph position (node index in `order`, offset in that node)
    // towards the right (right_to_left = false) or the left; dozeu's extend over do_poa (src/dozeu_interface.cpp:210-307)
    struct Extension {
        int32_t score = 0;
        size_t end_node = 0, end_ref_offset = 0, end_query = 0;   // position of the maximum, in `order` / read coordinates
        std::vector<Mapping> mappings;                              // traceback (only when requested), already in read order
        size_t matches = 0;
    };
    void xdrop_align(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order,
                     const std::vector<MaximalExactMatch>& mems, bool reverse_complemented, uint16_t max_gap_length) const;
    void xdrop_align_many(std::vector<XdropRequest>& requests) const;
    Extension xdrop_extend(const HandleGraph& g, const std::vector<handle_t>& order, size_t node_index, size_t ref_offset,
                           const std::string& read, const std::string& quality, size_t query_offset, bool right_to_left,
                           bool traceback, uint16_t max_gap_length) const;
    // xdrop_extend in two halves around the engine call (a job must stay where it is between them: its problem points into it)
    struct ExtensionJob {
        Extension ext;                                  // the answer when nothing runs, and the start position
        size_t node_index = 0, ref_offset = 0, query_offset = 0; bool right_to_left = false, traceback = false, runs = false;
        std::string query, qqual; PackedGraph pg; std::vector<size_t> kept; vgk_gssw_problem prob{};
    };
    void xdrop_extend_prepare(const HandleGraph& g, const std::vector<handle_t>& order, size_t node_index, size_t ref_offset,
                              const std::string& read, const std::string& quality, size_t query_offset, bool right_to_left,
                              bool traceback, uint16_t max_gap_length, ExtensionJob& job) const;
    Extension xdrop_extend_finish(const HandleGraph& g, const std::vector<handle_t>& order, const std::string& read, ExtensionJob& job,
                                  const vgk_result& res, std::vector<vgk_op>& ops) const;
    // the scan for a head when there is no seed (scan_seed_position), likewise
    struct ScanJob { std::string tail, tail_q; PackedGraph pg; std::vector<handle_t> run_order; size_t scan_len = 0; vgk_gssw_problem prob{}; };
    void xdrop_scan_prepare(const Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order, bool direction, ScanJob& job) const;
    // what follows the traced extension from the head (full-length insertion, the unseen read part, identity)
    void xdrop_finish(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order, Extension& down, size_t head_node, size_t head_ref,
                      size_t head_query, bool direction) const;
    void align_internal(Alignment& alignment, std::vector<Alignment>* multi_alignments, const HandleGraph& g,
                        bool pinned, bool pin_left, int32_t max_alt_alns, bool traceback_aln) const;
};

// QualAdjAligner (reference: src/aligner.hpp:218-258, src/aligner.cpp:859-1348): the same calls with scores and
// bonuses adjusted by base quality.  Reads must carry raw phred qualities (Alignment.quality).
class QualAdjAligner : public Aligner {
public:
    QualAdjAligner(const int8_t* score_matrix = default_score_matrix,
                   int8_t gap_open = default_gap_open,
                   int8_t gap_extension = default_gap_extension,
                   int8_t full_length_bonus = default_full_length_bonus,
                   double gc_content = 0.5,
                   std::shared_ptr<EngineApi> engine = nullptr,
                   int device = 0);
private:
    QualAdjAligner(QualAdjAlignmentScorer* owned, std::shared_ptr<EngineApi> engine, int device);
};

// Deferred submission: giraffe / map call the aligner once per read from many OpenMP threads
// (src/subcommand/giraffe_main.cpp:2416-2471), the engine wants thousands of problems per launch.  An AlignmentBatch takes the same
// calls from any number of threads, keeps the prepared problems, and a flush runs them in one engine call per kernel family and
// fills every Alignment exactly as the direct call would have.  The Alignment objects and the graphs must stay alive until the
// flush that runs them has returned.
//   * submission is thread-safe (the host-side preparation of a problem runs on the submitting thread, outside the lock);
//   * `max_pending` > 0: the submission that makes the batch that large flushes it (on the submitting thread);
//   * several devices: one Aligner (= one engine context) per device, successive flushes go to them in turn, and flushes on
//     different devices run concurrently when different threads trigger them (vg is ONE process with many threads: this is the
//     multi-GPU path inside it; bench.py's one-process-per-GPU sharding is the other);
//   * flush() returns when everything submitted before it — by any thread — has been answered.
class AlignmentBatch {
public:
    explicit AlignmentBatch(const Aligner& aligner, size_t max_pending = 0);
    AlignmentBatch(const std::vector<const Aligner*>& per_device, size_t max_pending = 0);      // same scoring on every one
    ~AlignmentBatch();
    void align(Alignment& alignment, const HandleGraph& g, bool traceback_aln);
    void align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, bool xdrop = false,
                      uint16_t xdrop_max_gap_length = default_xdrop_max_gap_length);
    void align_global_banded(Alignment& alignment, const HandleGraph& g, int32_t band_padding = 0, bool permissive_banding = true,
                             uint64_t max_cells = std::numeric_limits<uint64_t>::max());
    // Aligner::align_xdrop (the seeded two-pass X-drop of the rescue call sites, src/minimizer_mapper.cpp:3385, :3426), deferred: answered by
    // the next flush() through Aligner::align_xdrop_many (a size-triggered flush leaves these waiting: their passes depend on each other)
    void align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<MaximalExactMatch>& mems, bool reverse_complemented,
                     uint16_t max_gap_length = default_xdrop_max_gap_length);
    size_t size() const;
    size_t flushes() const { return n_flushes; }
    void flush();
    // Failures of single problems (the exceptions the direct calls throw: no alignment in the band, matrices too big, an engine limit)
    // normally end the flush that meets one.  Isolated, each is kept for its own caller — failure_of(alignment) after the flush — and
    // every other problem of the batch is still answered.  Set before submitting.
    bool isolate_failures = false;
    std::exception_ptr failure_of(const Alignment& alignment) const;
private:
    void submit(std::unique_ptr<Aligner::Job> job);
    void run(std::vector<std::unique_ptr<Aligner::Job>>& jobs, size_t device, std::vector<Aligner::XdropRequest>* xdrops = nullptr);
    std::vector<const Aligner*> aligners;
    size_t max_pending;
    mutable std::mutex mu; std::condition_variable idle;
    std::vector<std::unique_ptr<Aligner::Job>> jobs;
    std::vector<Aligner::XdropRequest> xdrop_requests;
    std::vector<std::unique_ptr<std::mutex>> device_mu;
    size_t next_device = 0, in_flight = 0, n_flushes = 0;
    std::exception_ptr failure;
    std::unordered_map<const Alignment*, std::exception_ptr> job_failures;
};

// nonATGCNtoN (reference: src/utility.cpp:323-332)
std::string nonATGCNtoN(const std::string& s);

}  // namespace vgamd
