#include "aligner.hpp"
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <functional>
#include <atomic>
#include <set>
#include <cmath>
#include <sstream>
#include <stdexcept>

namespace vgamd {

// (nonATGCNtoN, the scorer tables and identify_pinning_points: vg_standin/alignment_scorer.cpp)

GSSWAligner::GSSWAligner(std::unique_ptr<MatrixAlignmentScorer> owned_scorer, std::shared_ptr<EngineApi> eng, int device,
                         const QualAdjAlignmentScorer* qual_adj)
    : scorer(std::move(owned_scorer)), engine(eng ? eng : load_engine()) {
    vgk_scoring s = scorer->as_vgk();
    int rc;
    if (qual_adj) {
        qual_adjusted = true;
        vgk_qual_adj qa{qual_adj->qual_adj_matrix.data(), qual_adj->qual_adj_full_length_bonuses.data()};
        rc = engine->create_qual_adj(device, &s, &qa, &ctx);
    } else rc = engine->create(device, &s, &ctx);
    if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: cannot create engine context: ") + engine->strerror(rc));
}

GSSWAligner::~GSSWAligner() { if (ctx) engine->destroy(ctx); }

vgk_graph GSSWAligner::PackedGraph::view() const {
    vgk_graph v{};
    v.n_nodes = (uint32_t)order.size();
    v.node_len = node_len.data(); v.seq = seq.data(); v.pred_off = pred_off.data(); v.pred_idx = pred_idx.data();
    return v;
}

GSSWAligner::PackedGraph GSSWAligner::create_packed_graph(const HandleGraph& g) const {
    return create_packed_graph(g, handlealgs::lazier_topological_order(&g));
}

GSSWAligner::PackedGraph GSSWAligner::create_packed_graph(const HandleGraph& g, const std::vector<handle_t>& order, bool raw_sequence) const {
    PackedGraph pg;
    pg.order = order;
    std::unordered_map<handle_t, uint32_t, handle_hash> index;
    for (uint32_t i = 0; i < order.size(); ++i) index[order[i]] = i;
    pg.pred_off.push_back(0);
    for (uint32_t i = 0; i < order.size(); ++i) {
        std::string s = raw_sequence ? g.get_sequence(order[i]) : nonATGCNtoN(g.get_sequence(order[i]));
        pg.node_len.push_back((uint32_t)s.size());
        pg.seq += s;
        bool bad = false;
        g.follow_edges_v(order[i], true, [&](const handle_t& prev) {
            auto it = index.find(prev);
            if (it == index.end()) return;            // edge leaves the ordered subgraph (src/aligner.cpp:596-599)
            if (it->second >= i) { bad = true; return; }
            pg.pred_idx.push_back(it->second);
        });
        if (bad) throw std::runtime_error("vgamd: graph handed to the aligner is not a DAG in the given order "
                                          "(reference dies here too: src/aligner.cpp:61-78)");
        pg.pred_off.push_back((uint32_t)pg.pred_idx.size());
    }
    return pg;
}

void GSSWAligner::ops_to_alignment(const PackedGraph& pg, const HandleGraph& seq_source, const vgk_result& res,
                                   const vgk_op* ops, Alignment& alignment) const {
    alignment.clear_path();
    alignment.score = res.score;
    alignment.query_position = 0;
    const std::string& to_seq = alignment.sequence;
    int to_pos = 0;
    int from_pos = res.first_offset;
    uint32_t i = 0; bool first_node = true;
    while (i < res.n_ops) {
        uint32_t node = ops[i].node;
        uint32_t j = i; while (j < res.n_ops && ops[j].node == node) ++j;
        const handle_t h = pg.order[node];
        const std::string node_seq = nonATGCNtoN(seq_source.get_sequence(h));
        alignment.path.mapping.emplace_back();
        Mapping& mapping = alignment.path.mapping.back();
        if (!first_node) from_pos = 0;
        first_node = false;
        mapping.position.node_id = seq_source.get_id(h);
        mapping.position.offset = from_pos;
        mapping.rank = (int64_t)alignment.path.mapping.size();
        for (uint32_t k = i; k < j; ++k) {
            int32_t length = ops[k].len;
            switch (ops[k].op) {
                case VGK_OP_M: {
                    int hpos = from_pos, last_start = from_pos, q = to_pos;
                    for (; hpos < from_pos + length; ++hpos, ++q) {
                        if (node_seq[hpos] != to_seq[q]) {
                            if (hpos - last_start > 0) { Edit e; e.from_length = e.to_length = hpos - last_start; mapping.edit.push_back(e); }
                            Edit e; e.from_length = e.to_length = 1; e.sequence = to_seq.substr(q, 1); mapping.edit.push_back(e);
                            last_start = hpos + 1;
                        }
                    }
                    if (hpos - last_start > 0) { Edit e; e.from_length = e.to_length = hpos - last_start; mapping.edit.push_back(e); }
                    to_pos += length; from_pos += length;
                } break;
                case VGK_OP_D: { Edit e; e.from_length = length; e.to_length = 0; mapping.edit.push_back(e); from_pos += length; } break;
                case VGK_OP_I:
                case VGK_OP_S: { Edit e; e.from_length = 0; e.to_length = length; e.sequence = to_seq.substr(to_pos, length);
                                 mapping.edit.push_back(e); to_pos += length; } break;
                default: throw std::runtime_error("vgamd: unsupported cigar op from engine");
            }
        }
        i = j;
    }
    alignment.identity = identity(alignment.path);
}

Aligner::Aligner(const int8_t* score_matrix, int8_t gap_open, int8_t gap_extension, int8_t full_length_bonus,
                 double gc_content, std::shared_ptr<EngineApi> eng, int device)
    : GSSWAligner(std::make_unique<MatrixAlignmentScorer>(score_matrix, gap_open, gap_extension, full_length_bonus), eng, device) {
    double dm[16];
    for (int i = 0; i < 16; ++i) dm[i] = score_matrix[i];
    scorer->log_base = QualAdjAlignmentScorer::recover_log_base(dm, gc_content);
    mapq_calc = std::make_unique<MappingQualityCalculator>((double)scorer->match, (double)scorer->mismatch, scorer->log_base);
}

Aligner::Aligner(std::unique_ptr<MatrixAlignmentScorer> owned_scorer, std::shared_ptr<EngineApi> eng, int device,
                 const QualAdjAlignmentScorer* qual_adj)
    : GSSWAligner(std::move(owned_scorer), eng, device, qual_adj) {}

static std::unique_ptr<QualAdjAlignmentScorer> make_qual_scorer(const int8_t* m, int8_t go, int8_t ge, int8_t b, double gc) {
    return std::make_unique<QualAdjAlignmentScorer>(m, go, ge, b, gc);
}

QualAdjAligner::QualAdjAligner(const int8_t* score_matrix, int8_t gap_open, int8_t gap_extension, int8_t full_length_bonus,
                               double gc_content, std::shared_ptr<EngineApi> eng, int device)
    : QualAdjAligner(make_qual_scorer(score_matrix, gap_open, gap_extension, full_length_bonus, gc_content).release(), eng, device) {}

QualAdjAligner::QualAdjAligner(QualAdjAlignmentScorer* owned, std::shared_ptr<EngineApi> eng, int device)
    : Aligner(std::unique_ptr<MatrixAlignmentScorer>(owned), eng, device, owned) {
    mapq_calc = std::make_unique<MappingQualityCalculator>((double)scorer->match, (double)scorer->mismatch, scorer->log_base);
}

// the reads of a quality-adjusted aligner must carry one quality per base (the reference asserts, src/banded_global_aligner.cpp:1981-1987)
static const uint8_t* quality_of(bool qual_adjusted, const std::string& quality, size_t read_len) {
    if (!qual_adjusted) return nullptr;
    if (quality.size() != read_len) throw std::invalid_argument("error:[QualAdjAligner] quality-adjusted alignment needs one base quality per read base");
    return reinterpret_cast<const uint8_t*>(quality.data());
}

// unreverse_graph_mapping (src/aligner.cpp:255-300) on the flat op list
static void unreverse_ops(std::vector<vgk_op>& ops, vgk_result& res, const std::vector<uint32_t>& node_len) {
    std::reverse(ops.begin(), ops.end());      // reverses node order and the elements inside each node at once
    if (ops.empty()) { res.first_offset = 0; return; }
    uint32_t first = ops[0].node, aligned = 0, n_nodes = 1;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (i && ops[i].node != ops[i - 1].node) ++n_nodes;
        if (ops[i].node == first && n_nodes == 1 && (ops[i].op == VGK_OP_M || ops[i].op == VGK_OP_D)) aligned += ops[i].len;
    }
    res.first_offset = (int32_t)node_len[first] - (int32_t)aligned - (n_nodes == 1 ? res.first_offset : 0);
}

// align_internal (src/aligner.cpp:344-564) in two halves, so that many calls can share one engine launch (AlignmentBatch):
// prepare_job builds the oriented / null-masked view, the packed graph and the engine problem; finish_job turns the
// engine's result into the Alignment exactly as the reference does after gssw returns.
struct Aligner::Job {
    Alignment* alignment = nullptr; const HandleGraph* g = nullptr;
    bool pinned = false, pin_left = false, traceback = true;
    std::unique_ptr<ReverseGraph> reversed_graph; std::unique_ptr<NullMaskingGraph> null_masked_graph;
    const HandleGraph* oriented_graph = nullptr; const HandleGraph* align_graph = nullptr;
    std::string reversed_sequence, reversed_quality;
    PackedGraph pg; std::vector<uint8_t> pin_mask;
    bool has_problem = false; vgk_gssw_problem prob{};
    // banded global jobs (align_global_banded) use the same carrier
    bool banded = false; vgk_banded_problem bprob{}; uint64_t max_cells = 0;
    // pinned X-drop jobs (align_pinned(..., xdrop = true)): the overlay the pass runs on, the read as the engine sees it, the head position
    bool xdrop = false, answered = false;
    std::unique_ptr<DozeuPinningOverlay> overlay; std::string run_seq, run_qual;
    handle_t head{}; bool have_head = false;
};

std::unique_ptr<Aligner::Job> Aligner::prepare_job(Alignment& alignment, const HandleGraph& g, bool pinned, bool pin_left, bool traceback_aln) const {
    // input contract (the reference prints and exit(1)s: src/aligner.cpp:348-363; we throw)
    if (pin_left && !pinned) throw std::invalid_argument("error:[Aligner] cannot choose pinned end in non-pinned alignment");
    auto job = std::make_unique<Job>();
    Job& j = *job;
    j.alignment = &alignment; j.g = &g; j.pinned = pinned; j.pin_left = pin_left; j.traceback = traceback_aln;
    j.oriented_graph = &g;
    const std::string* align_sequence = &alignment.sequence;
    const std::string* align_quality = &alignment.quality;
    if (pin_left) {
        j.reversed_graph = std::make_unique<ReverseGraph>(&g, false);
        j.oriented_graph = j.reversed_graph.get();
        j.reversed_sequence.assign(alignment.sequence.rbegin(), alignment.sequence.rend());
        align_sequence = &j.reversed_sequence;
        j.reversed_quality.assign(alignment.quality.rbegin(), alignment.quality.rend());
        align_quality = &j.reversed_quality;
    }
    std::unordered_set<nid_t> pinning_ids;
    j.align_graph = j.oriented_graph;
    if (pinned) {
        pinning_ids = identify_pinning_points(*j.oriented_graph);
        j.null_masked_graph = std::make_unique<NullMaskingGraph>(j.oriented_graph);
        j.align_graph = j.null_masked_graph.get();
    }
    j.pg = create_packed_graph(*j.align_graph);
    if (!j.pg.order.empty() && !align_sequence->empty()) {
        vgk_gssw_problem& prob = j.prob;
        prob.read = align_sequence->data(); prob.read_len = (uint32_t)align_sequence->size();
        prob.qual = quality_of(qual_adjusted, *align_quality, align_sequence->size());
        prob.flags = (pinned ? VGK_GSSW_PINNED : VGK_GSSW_LOCAL) | (traceback_aln ? VGK_GSSW_TRACEBACK : 0);
        prob.graph = j.pg.view();
        if (pinned) {
            j.pin_mask.resize(j.pg.order.size());
            for (size_t i = 0; i < j.pg.order.size(); ++i) j.pin_mask[i] = pinning_ids.count(j.align_graph->get_id(j.pg.order[i])) ? 1 : 0;
            prob.pinning = j.pin_mask.data();
        }
        j.has_problem = true;
    }
    return job;
}

void Aligner::finish_job(Job& j, vgk_result res, std::vector<vgk_op> ops, std::vector<Alignment>* multi_alignments, int32_t max_alt_alns) const {
    Alignment& alignment = *j.alignment; const HandleGraph& g = *j.g;
    const bool did_dp = j.has_problem;
    if (did_dp && res.status != VGK_OK) throw std::runtime_error(std::string("vgamd: gssw problem failed: ") + engine->strerror(res.status));
    if (j.traceback) {
        if (j.pinned) {
            if (did_dp && res.score > 0) {
                if (j.pin_left) unreverse_ops(ops, res, j.pg.node_len);
                // after un-reversal the cigar refers to the forward sequences of g
                ops_to_alignment(j.pg, g, res, ops.data(), alignment);
                if (multi_alignments) multi_alignments->emplace_back(alignment);   // the alternates follow in align_internal
            } else if (g.get_node_count() > 0) {
                // no positive-score traceback: synthesise soft clips at the id-sorted tail nodes.
                // The reference writes every alternate into `alignment` (src/aligner.cpp:505-520); reproduced as is.
                auto pinning_points = handlealgs::tail_nodes(j.oriented_graph);
                std::sort(pinning_points.begin(), pinning_points.end(), [&](const handle_t& a, const handle_t& b) {
                    return j.oriented_graph->get_id(a) < j.oriented_graph->get_id(b); });
                for (size_t i = 0; i < (size_t)max_alt_alns && i < pinning_points.size(); i++) {
                    if (multi_alignments) multi_alignments->emplace_back();
                    handle_t& pinning_point = pinning_points[i];
                    alignment.path.mapping.emplace_back();
                    Mapping& mapping = alignment.path.mapping.back();
                    mapping.rank = 1;
                    mapping.position.node_id = j.oriented_graph->get_id(pinning_point);
                    mapping.position.offset = j.pin_left ? 0 : (int64_t)j.oriented_graph->get_length(pinning_point);
                    Edit e; e.to_length = (int32_t)alignment.sequence.length(); e.sequence = alignment.sequence;
                    mapping.edit.push_back(e);
                    if (i == 0 && multi_alignments) multi_alignments->back() = alignment;
                }
            }
        } else {
            ops_to_alignment(j.pg, g, res, ops.data(), alignment);
        }
    } else {
        alignment.score = res.score;
        alignment.path.mapping.emplace_back();
        Position& p = alignment.path.mapping.back().position;
        if (res.end_node >= 0) { p.node_id = j.align_graph->get_id(j.pg.order[res.end_node]); p.offset = res.end_offset; }
    }
}

void Aligner::align_internal(Alignment& alignment, std::vector<Alignment>* multi_alignments, const HandleGraph& g,
                             bool pinned, bool pin_left, int32_t max_alt_alns, bool traceback_aln) const {
    if (multi_alignments && !pinned) throw std::invalid_argument("error:[Aligner] multiple traceback is not implemented in local alignment, only pinned and global");
    if (!multi_alignments && max_alt_alns != 1) throw std::invalid_argument("error:[Aligner] cannot specify maximum number of alignments in single alignment");
    if (max_alt_alns <= 0) throw std::invalid_argument("error:[Aligner] cannot do less than 1 alignment");
    auto job = prepare_job(alignment, g, pinned, pin_left, traceback_aln);
    vgk_result res{};
    std::vector<vgk_op> ops;
    if (job->has_problem && multi_alignments && max_alt_alns > 1) {
        // k-best pinned tracebacks (src/aligner.cpp:423-435, :455-480): the engine returns them best first, all with score > 0
        const size_t per = job->prob.read_len + job->pg.seq.size() + job->pg.order.size() + 4;
        std::vector<vgk_result> all((size_t)max_alt_alns); std::vector<vgk_op> all_ops(per * (size_t)max_alt_alns + 1);
        uint32_t count = 0; size_t written = 0;
        int rc = engine->gssw_align_multi(ctx, &job->prob, 1, (uint32_t)max_alt_alns, all.data(), &count, all_ops.data(), all_ops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: gssw engine failed: ") + engine->strerror(rc));
        if (all[0].status != VGK_OK) throw std::runtime_error(std::string("vgamd: gssw problem failed: ") + engine->strerror(all[0].status));
        if (count > 0) { res = all[0]; ops.assign(all_ops.begin() + res.ops_begin, all_ops.begin() + res.ops_begin + res.n_ops); res.ops_begin = 0; }
        finish_job(*job, res, std::move(ops), multi_alignments, max_alt_alns);
        for (uint32_t k = 1; k < count; ++k) {
            vgk_result r = all[k];
            std::vector<vgk_op> o(all_ops.begin() + r.ops_begin, all_ops.begin() + r.ops_begin + r.n_ops); r.ops_begin = 0;
            if (pin_left) unreverse_ops(o, r, job->pg.node_len);
            multi_alignments->emplace_back();
            Alignment& next = multi_alignments->back();
            next.sequence = alignment.sequence; next.quality = alignment.quality;
            ops_to_alignment(job->pg, g, r, o.data(), next);
        }
        return;
    }
    if (job->has_problem) {
        ops.resize(job->prob.read_len + job->pg.seq.size() + job->pg.order.size() + 4);
        size_t written = 0;
        int rc = engine->gssw_align(ctx, &job->prob, 1, &res, ops.data(), ops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: gssw engine failed: ") + engine->strerror(rc));
        ops.resize(res.n_ops);
    }
    finish_job(*job, res, std::move(ops), multi_alignments, max_alt_alns);
}

// ---- AlignmentBatch: many Aligner calls, one engine launch per kernel family (the deferred-submission shim) -------
AlignmentBatch::AlignmentBatch(const Aligner& aligner, size_t max_pending) : AlignmentBatch(std::vector<const Aligner*>{&aligner}, max_pending) {}
AlignmentBatch::AlignmentBatch(const std::vector<const Aligner*>& per_device, size_t max_pending) : aligners(per_device), max_pending(max_pending) {
    if (aligners.empty()) throw std::invalid_argument("AlignmentBatch: no aligner");
    for (size_t i = 0; i < aligners.size(); ++i) device_mu.emplace_back(new std::mutex());
}
AlignmentBatch::~AlignmentBatch() = default;
std::exception_ptr AlignmentBatch::failure_of(const Alignment& alignment) const {
    std::lock_guard<std::mutex> lk(mu);
    auto found = job_failures.find(&alignment);
    return found == job_failures.end() ? nullptr : found->second;
}
void AlignmentBatch::submit(std::unique_ptr<Aligner::Job> job) {
    std::vector<std::unique_ptr<Aligner::Job>> full; std::vector<Aligner::XdropRequest> xfull; size_t device = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        jobs.push_back(std::move(job));
        // (the deferred seeded X-drops count like every other pending call, and go with the batch they filled)
        if (max_pending && jobs.size() + xdrop_requests.size() >= max_pending) { full.swap(jobs); xfull.swap(xdrop_requests); device = next_device++ % aligners.size(); ++in_flight; ++n_flushes; }
    }
    if (!full.empty() || !xfull.empty()) run(full, device, &xfull);            // the submission that filled the batch runs it
}
void AlignmentBatch::align(Alignment& alignment, const HandleGraph& g, bool traceback_aln) { submit(aligners[0]->prepare_job(alignment, g, false, false, traceback_aln)); }
void AlignmentBatch::align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, bool xdrop, uint16_t xdrop_max_gap_length) {
    submit(xdrop ? aligners[0]->prepare_xdrop_job(alignment, g, pin_left, xdrop_max_gap_length) : aligners[0]->prepare_job(alignment, g, true, pin_left, true));
}
void AlignmentBatch::align_global_banded(Alignment& alignment, const HandleGraph& g, int32_t band_padding, bool permissive_banding, uint64_t max_cells) {
    submit(aligners[0]->prepare_banded_job(alignment, g, band_padding, permissive_banding, max_cells));
}
void AlignmentBatch::align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<MaximalExactMatch>& mems, bool reverse_complemented, uint16_t max_gap_length) {
    Aligner::XdropRequest rq;
    rq.alignment = &alignment; rq.graph = &g; rq.mems = mems; rq.reverse_complemented = reverse_complemented; rq.max_gap_length = max_gap_length;
    std::vector<std::unique_ptr<Aligner::Job>> full; std::vector<Aligner::XdropRequest> xfull; size_t device = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        xdrop_requests.push_back(std::move(rq));
        if (max_pending && jobs.size() + xdrop_requests.size() >= max_pending) { full.swap(jobs); xfull.swap(xdrop_requests); device = next_device++ % aligners.size(); ++in_flight; ++n_flushes; }
    }
    if (!full.empty() || !xfull.empty()) run(full, device, &xfull);
}
size_t AlignmentBatch::size() const { std::lock_guard<std::mutex> lk(mu); return jobs.size() + xdrop_requests.size(); }
void AlignmentBatch::flush() {
    std::vector<std::unique_ptr<Aligner::Job>> mine; std::vector<Aligner::XdropRequest> xmine; size_t device = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!jobs.empty() || !xdrop_requests.empty()) { mine.swap(jobs); xmine.swap(xdrop_requests); device = next_device++ % aligners.size(); ++in_flight; ++n_flushes; }
    }
    if (!mine.empty() || !xmine.empty()) run(mine, device, &xmine);
    std::unique_lock<std::mutex> lk(mu);             // ... and whatever other threads' flushes still have in the engine
    idle.wait(lk, [this] { return in_flight == 0; });
    if (failure) { std::exception_ptr f = failure; failure = nullptr; std::rethrow_exception(f); }
}
void AlignmentBatch::run(std::vector<std::unique_ptr<Aligner::Job>>& run, size_t device, std::vector<Aligner::XdropRequest>* xdrops) {
    struct Done {                                    // in_flight goes down however this ends
        AlignmentBatch* b; std::exception_ptr err;
        ~Done() { std::lock_guard<std::mutex> lk(b->mu); if (err && !b->failure) b->failure = err; if (--b->in_flight == 0) b->idle.notify_all(); }
    } done{this, nullptr};
    try {
    const Aligner& aligner = *aligners[device];
    std::lock_guard<std::mutex> on_device(*device_mu[device]);      // one flush at a time per device; other devices run beside it
    {   // a reused batch: what an Alignment failed with in an earlier run says nothing about this one
        std::lock_guard<std::mutex> lk(mu);
        if (!job_failures.empty()) {
            for (const auto& j : run) job_failures.erase(j->alignment);
            if (xdrops) for (const Aligner::XdropRequest& rq : *xdrops) job_failures.erase(rq.alignment);
        }
    }
    // gssw family
    std::vector<vgk_gssw_problem> probs; std::vector<size_t> owner;
    size_t cap = 0;
    for (size_t i = 0; i < run.size(); ++i) if (!run[i]->banded && run[i]->has_problem) {
        probs.push_back(run[i]->prob); owner.push_back(i);
        cap += run[i]->prob.read_len + run[i]->pg.seq.size() + run[i]->pg.order.size() + 4;
    }
    std::vector<vgk_result> res(probs.size()); std::vector<vgk_op> ops(cap + 1);
    if (!probs.empty()) {
        size_t written = 0;
        int rc = aligner.engine_api().gssw_align(aligner.engine_context(), probs.data(), (uint32_t)probs.size(), res.data(), ops.data(), ops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: gssw engine failed: ") + aligner.engine_api().strerror(rc));
    }
    // banded family
    std::vector<vgk_banded_problem> bprobs; std::vector<size_t> bowner;
    size_t bcap = 0;
    for (size_t i = 0; i < run.size(); ++i) if (run[i]->banded && run[i]->has_problem) {
        bprobs.push_back(run[i]->bprob); bowner.push_back(i);
        bcap += run[i]->bprob.read_len + run[i]->pg.seq.size() + 2 * run[i]->pg.order.size() + 8;
    }
    std::vector<vgk_result> bres(bprobs.size()); std::vector<vgk_op> bops(bcap + 1);
    if (!bprobs.empty()) {
        size_t written = 0;
        int rc = aligner.engine_api().banded_align(aligner.engine_context(), bprobs.data(), (uint32_t)bprobs.size(), bres.data(), bops.data(), bops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: banded engine failed: ") + aligner.engine_api().strerror(rc));
    }
    // apply: problems that had nothing to run (empty read / empty graph) still go through the result code
    std::vector<const vgk_result*> rof(run.size(), nullptr);
    for (size_t k = 0; k < owner.size(); ++k) rof[owner[k]] = &res[k];
    for (size_t k = 0; k < bowner.size(); ++k) rof[bowner[k]] = &bres[k];
    for (size_t i = 0; i < run.size(); ++i) {
        Aligner::Job& j = *run[i];
        try {
            if (j.banded) { aligner.finish_banded_job(j, rof[i] ? *rof[i] : vgk_result{}, rof[i] ? bops.data() + rof[i]->ops_begin : nullptr); continue; }
            vgk_result r{}; std::vector<vgk_op> o;
            if (rof[i]) { r = *rof[i]; o.assign(ops.begin() + r.ops_begin, ops.begin() + r.ops_begin + r.n_ops); r.ops_begin = 0; }
            if (j.xdrop) { aligner.finish_xdrop_job(j, r, std::move(o)); continue; }
            aligner.finish_job(j, r, std::move(o), nullptr, 1);
        } catch (...) {
            // what the direct call would have thrown at its caller (NoAlignmentInBandException, BandMatricesTooBigException, ...): with
            // isolated failures it is kept for that caller (failure_of) and the other problems of the batch are still answered
            if (!isolate_failures) throw;
            std::lock_guard<std::mutex> lk(mu);
            job_failures[j.alignment] = std::current_exception();
        }
    }
    if (xdrops && !xdrops->empty()) {                // the seeded two-pass X-drops: two or three engine calls for all of them
        try { aligner.align_xdrop_many(*xdrops); }
        catch (...) {
            // one request's failure fails the joint call: with isolated failures every request is answered on its own and only the
            // failing ones keep their exception (failure_of), like the other calls of the batch
            if (!isolate_failures) throw;
            for (Aligner::XdropRequest& rq : *xdrops) {
                std::vector<Aligner::XdropRequest> one; one.push_back(rq);
                try { rq.alignment->clear_path(); aligner.align_xdrop_many(one); }
                catch (...) { std::lock_guard<std::mutex> lk(mu); job_failures[rq.alignment] = std::current_exception(); }
            }
        }
    }
    } catch (...) { done.err = std::current_exception(); }
}

void Aligner::align(Alignment& alignment, const HandleGraph& g, bool traceback_aln) const {
    align_internal(alignment, nullptr, g, false, false, 1, traceback_aln);
}

void Aligner::align(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& topological_order) const {
    PackedGraph pg = create_packed_graph(g, topological_order);
    vgk_gssw_problem prob{};
    prob.read = alignment.sequence.data(); prob.read_len = (uint32_t)alignment.sequence.size();
    prob.qual = quality_of(qual_adjusted, alignment.quality, alignment.sequence.size());
    prob.flags = VGK_GSSW_LOCAL | VGK_GSSW_TRACEBACK;
    prob.graph = pg.view();
    vgk_result res{};
    std::vector<vgk_op> ops(prob.read_len + pg.seq.size() + pg.order.size() + 4);
    size_t written = 0;
    int rc = engine->gssw_align(ctx, &prob, 1, &res, ops.data(), ops.size(), &written);
    if (rc != VGK_OK || res.status != VGK_OK)
        throw std::runtime_error(std::string("vgamd: gssw engine failed: ") + engine->strerror(rc ? rc : res.status));
    ops_to_alignment(pg, g, res, ops.data(), alignment);
    // node ids were order indices inside the engine; ops_to_alignment already wrote
    // g.get_id(handle); add the strand (src/aligner.cpp:615-621)
    size_t gi = 0; uint32_t i = 0;
    while (i < res.n_ops) {
        uint32_t node = ops[i].node; while (i < res.n_ops && ops[i].node == node) ++i;
        alignment.path.mapping[gi++].position.is_reverse = g.get_is_reverse(topological_order[node]);
    }
}

void Aligner::align_pinned(Alignment& alignment, const HandleGraph& g, bool pin_left, bool xdrop,
                           uint16_t xdrop_max_gap_length) const {
    if (!xdrop) { align_internal(alignment, nullptr, g, true, pin_left, 1, true); return; }
    auto job = prepare_xdrop_job(alignment, g, pin_left, xdrop_max_gap_length);
    vgk_result res{}; std::vector<vgk_op> ops;
    if (job->has_problem) {
        ops.resize(job->prob.read_len + job->pg.seq.size() + job->pg.order.size() + 4);
        size_t written = 0;
        int rc = xdrop_band ? engine->xdrop_band_align(ctx, &job->prob, 1, &res, ops.data(), ops.size(), &written, nullptr)
                            : engine->gssw_align(ctx, &job->prob, 1, &res, ops.data(), ops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: xdrop engine failed: ") + engine->strerror(rc));
        ops.resize(res.n_ops);
    }
    finish_xdrop_job(*job, res, std::move(ops));
}

// Aligner::align_pinned's xdrop branch (src/aligner.cpp:628-682) + DozeuInterface::align_pinned (src/dozeu_interface.cpp:724-766) up
// to the engine call: everything that needs only the host.  The job keeps the overlay, the (possibly reversed) read and the
// packed graph alive until finish_xdrop_job.
std::unique_ptr<Aligner::Job> Aligner::prepare_xdrop_job(Alignment& alignment, const HandleGraph& g, bool pin_left, uint16_t xdrop_max_gap_length) const {
    auto job = std::make_unique<Job>();
    Job& j = *job;
    j.alignment = &alignment; j.g = &g; j.pinned = true; j.pin_left = pin_left; j.traceback = true; j.xdrop = true;
    // dozeu declines to produce an alignment when the gap is set to 0 (src/aligner.cpp:637-638)
    xdrop_max_gap_length = std::max<uint16_t>(xdrop_max_gap_length, 1);
    // wrap the graph so that empty pinning points are handled correctly (src/aligner.cpp:640-641)
    j.overlay = std::make_unique<DozeuPinningOverlay>(&g, !pin_left);
    const DozeuPinningOverlay& overlay = *j.overlay;
    if (overlay.get_node_count() == 0 && g.get_node_count() != 0) {
        // only empty pinning nodes: infer the soft clip from the pinning point (src/aligner.cpp:643-668)
        g.for_each_handle([&](const handle_t& handle) {
            bool can_pin = g.follow_edges(handle, pin_left, [&](const handle_t&) { return false; });
            if (can_pin) {
                alignment.path.mapping.emplace_back();
                Mapping& mapping = alignment.path.mapping.back();
                mapping.position.node_id = g.get_id(handle); mapping.position.is_reverse = false;
                mapping.position.offset = pin_left ? 0 : (int64_t)g.get_length(handle);
                mapping.rank = 1;
                Edit e; e.from_length = 0; e.to_length = (int32_t)alignment.sequence.size(); e.sequence = alignment.sequence;
                mapping.edit.push_back(e);
                alignment.score = 0;
                return false;
            }
            return true;
        });
        j.answered = true;
        return job;
    }
    std::vector<handle_t> order = handlealgs::lazier_topological_order(&overlay);     // lazy_topological_order in the reference (:728)
    if (order.empty()) { j.answered = true; return job; }
    // The engine extends left to right from every source node.  A right pin is the same problem on the
    // reversed graph and read (dozeu walks the node strings backwards with a reverse-packed query, :178-185, :282-283).
    const HandleGraph* run_graph = &overlay;
    j.run_seq = alignment.sequence; j.run_qual = alignment.quality;
    if (!pin_left) {
        j.reversed_graph = std::make_unique<ReverseGraph>(&overlay, false);
        run_graph = j.reversed_graph.get();
        std::reverse(j.run_seq.begin(), j.run_seq.end()); std::reverse(j.run_qual.begin(), j.run_qual.end());
    }
    // dozeu sees raw get_sequence(); the packed graph is only a transport for (order, lengths, edges, bases)
    std::vector<handle_t> run_order = handlealgs::lazier_topological_order(run_graph);
    PackedGraph& pg = j.pg;
    pg.order = run_order;
    std::unordered_map<handle_t, uint32_t, handle_hash> index;
    for (uint32_t i = 0; i < run_order.size(); ++i) index[run_order[i]] = i;
    pg.pred_off.push_back(0);
    for (uint32_t i = 0; i < run_order.size(); ++i) {
        std::string sq = run_graph->get_sequence(run_order[i]);
        pg.node_len.push_back((uint32_t)sq.size()); pg.seq += sq;
        run_graph->follow_edges_v(run_order[i], true, [&](const handle_t& prev) { auto it = index.find(prev); if (it != index.end()) pg.pred_idx.push_back(it->second); });
        pg.pred_off.push_back((uint32_t)pg.pred_idx.size());
    }
    // head position = first tip in the pin direction in the (forward) order (:738-755); used when nothing aligns
    j.head = order.front();
    for (const handle_t& h : order) {
        if (overlay.follow_edges(h, pin_left, [](const handle_t&) { return false; })) { j.head = h; j.have_head = true; break; }
    }
    if (!j.run_seq.empty()) {
        vgk_gssw_problem& prob = j.prob;
        prob.read = j.run_seq.data(); prob.read_len = (uint32_t)j.run_seq.size();
        prob.qual = quality_of(qual_adjusted, j.run_qual, j.run_seq.size());
        prob.flags = VGK_XDROP_PINNED | VGK_GSSW_TRACEBACK;
        prob.graph = pg.view(); prob.max_gap_length = xdrop_max_gap_length;
        j.has_problem = true;
    }
    return job;
}

// DozeuInterface::calculate_and_save_alignment (src/dozeu_interface.cpp:338-572) + the id translation after duplications
// (src/aligner.cpp:673-680)
void Aligner::finish_xdrop_job(Job& j, vgk_result res, std::vector<vgk_op> ops) const {
    if (j.answered) return;
    Alignment& alignment = *j.alignment;
    const DozeuPinningOverlay& g = *j.overlay;          // the graph the pass ran on
    const bool pin_left = j.pin_left;
    const PackedGraph& pg = j.pg;
    if (j.has_problem && res.status != VGK_OK) throw std::runtime_error(std::string("vgamd: xdrop engine failed: ") + engine->strerror(res.status));
    alignment.clear_path();
    alignment.score = res.score;
    if (res.score == 0 || ops.empty()) {
        // no alignment scoring anything other than 0: full-length insertion at the head (:344-359)
        if (j.have_head) {
            alignment.path.mapping.emplace_back();
            Mapping& m = alignment.path.mapping.back();
            m.position.node_id = g.get_id(j.head); m.position.is_reverse = g.get_is_reverse(j.head);
            m.position.offset = pin_left ? 0 : (int64_t)g.get_length(j.head);
            m.rank = 1;
            Edit e; e.from_length = 0; e.to_length = (int32_t)alignment.sequence.size(); e.sequence = alignment.sequence;
            m.edit.push_back(e);
        }
    } else {
        if (!pin_left) unreverse_ops(ops, res, pg.node_len);
        // dozeu path -> vg Path (calculate_and_save_alignment): matches merged, every mismatching base its own edit,
        // the unaligned read end an insertion (merged into a leading insertion, separate when trailing)
        const std::string& query = alignment.sequence;
        size_t to_pos = 0, matches = 0;
        int from_pos = res.first_offset;
        uint32_t i = 0; bool first_node = true;
        while (i < ops.size()) {
            uint32_t node = ops[i].node, j = i;
            while (j < ops.size() && ops[j].node == node) ++j;
            const handle_t h = pg.order[node];
            const std::string node_seq = g.get_sequence(h);
            alignment.path.mapping.emplace_back();
            Mapping& mapping = alignment.path.mapping.back();
            if (!first_node) from_pos = 0;
            first_node = false;
            mapping.position.node_id = g.get_id(h); mapping.position.is_reverse = g.get_is_reverse(h);
            mapping.position.offset = from_pos; mapping.rank = (int64_t)alignment.path.mapping.size();
            for (uint32_t k = i; k < j; ++k) {
                const int32_t len = ops[k].len;
                switch (ops[k].op) {
                    case VGK_OP_M: {
                        int run = 0;
                        for (int t = 0; t < len; ++t) {
                            if (node_seq[from_pos + t] == query[to_pos + t]) { ++run; ++matches; }
                            else {
                                if (run) { Edit e; e.from_length = e.to_length = run; mapping.edit.push_back(e); run = 0; }
                                Edit e; e.from_length = e.to_length = 1; e.sequence = query.substr(to_pos + t, 1); mapping.edit.push_back(e);
                            }
                        }
                        if (run) { Edit e; e.from_length = e.to_length = run; mapping.edit.push_back(e); }
                        from_pos += len; to_pos += len;
                    } break;
                    case VGK_OP_D: { Edit e; e.from_length = len; e.to_length = 0; mapping.edit.push_back(e); from_pos += len; } break;
                    case VGK_OP_I:
                    case VGK_OP_S: {
                        // a leading clip merges with an adjacent path insertion (state machine of :498-507); a trailing one is pushed on its own (:520-526)
                        const bool merge_prev = !mapping.edit.empty() && edit_is_insertion(mapping.edit.back()) &&
                                                !(ops[k].op == VGK_OP_S && k + 1 == ops.size());
                        if (merge_prev) { Edit& e = mapping.edit.back(); e.to_length += len; e.sequence += query.substr(to_pos, len); }
                        else { Edit e; e.from_length = 0; e.to_length = len; e.sequence = query.substr(to_pos, len); mapping.edit.push_back(e); }
                        to_pos += len;
                    } break;
                    default: throw std::runtime_error("vgamd: unsupported cigar op from engine");
                }
            }
            i = j;
        }
        alignment.identity = query.empty() ? 0.0 : (double)matches / (double)query.size();
        alignment.query_position = 0;
    }
    if (g.performed_duplications()) {
        // the overlay is not a strict subset of the underlying graph: translate duplicate ids back (src/aligner.cpp:673-680)
        for (Mapping& m : alignment.path.mapping) {
            handle_t under = g.get_underlying_handle(g.get_handle(m.position.node_id));
            m.position.node_id = j.g->get_id(under); m.position.is_reverse = j.g->get_is_reverse(under);
        }
    }
}

void Aligner::align_pinned_multi(Alignment& alignment, std::vector<Alignment>& alt_alignments, const HandleGraph& g,
                                 bool pin_left, int32_t max_alt_alns) const {
    if (!alt_alignments.empty())
        throw std::invalid_argument("error:[Aligner::align_pinned_multi] output vector must be empty for pinned multi-aligning");
    align_internal(alignment, &alt_alignments, g, true, pin_left, max_alt_alns, true);
}

static void json_escape(std::ostringstream& o, const std::string& s) {
    o << '"';
    for (char c : s) { if (c == '"' || c == '\\') o << '\\'; o << c; }
    o << '"';
}

// ---- banded global alignment (src/aligner.cpp:699-760, :1189-1248) -------------------------------------------------
void DeletionAligner::align(Alignment& aln, const HandleGraph& graph) const {
    if (!aln.sequence.empty()) throw std::invalid_argument("error: DeletionAligner can only be used for alignments of empty strings");
    aln.clear_path();
    std::vector<handle_t> order = handlealgs::lazier_topological_order(&graph);
    if (order.empty()) { aln.score = 0; return; }
    std::unordered_map<handle_t, size_t, handle_hash> index_of;
    for (size_t i = 0; i < order.size(); ++i) index_of[order[i]] = i;
    // shortest distance to the left side of every node, and the sinks with the length of the walk through them (:57-113)
    const size_t inf = std::numeric_limits<size_t>::max();
    std::vector<size_t> dists(order.size(), inf);
    size_t best_sink = inf, best_dist = inf;
    for (size_t i = 0; i < order.size(); ++i) {
        if (dists[i] == inf) dists[i] = 0;
        size_t thru = dists[i] + graph.get_length(order[i]);
        bool is_sink = true;
        graph.follow_edges_v(order[i], false, [&](const handle_t& next) { size_t j = index_of.at(next); dists[j] = std::min(dists[j], thru); is_sink = false; });
        if (is_sink && thru < best_dist) { best_dist = thru; best_sink = i; }     // the min-heap keeps the lowest (dist, sink) (:117-132, :195-205)
    }
    // walk back along the first predecessor that realises each distance (:140-163)
    std::vector<handle_t> trace;
    for (size_t at = best_sink; at != inf;) {
        trace.push_back(order[at]);
        size_t next = inf;
        graph.follow_edges_v(order[at], true, [&](const handle_t& prev) {
            size_t idx = index_of.at(prev);
            if (next == inf && dists[idx] + graph.get_length(prev) == dists[at]) next = idx;
        });
        at = next;
    }
    int64_t total = 0;
    for (auto it = trace.rbegin(); it != trace.rend(); ++it) {
        aln.path.mapping.emplace_back();
        Mapping& m = aln.path.mapping.back();
        m.position.node_id = graph.get_id(*it); m.position.is_reverse = graph.get_is_reverse(*it);
        Edit e; e.from_length = (int32_t)graph.get_length(*it); e.to_length = 0; m.edit.push_back(e);
        total += e.from_length;
    }
    aln.score = total ? -gap_open - (int32_t)(total - 1) * gap_extension : 0;
}

// The k shortest source-to-sink walks as pure deletions (src/deletion_aligner.cpp:27-41, :115-199): a min-heap of (distance,
// deflections) as in the reference; walks with equal distance come out in the order of their deflection lists.
void DeletionAligner::align_multi(Alignment& aln, std::vector<Alignment>& alt_alns, const HandleGraph& graph, int32_t max_alt_alns) const {
    if (!aln.sequence.empty()) throw std::invalid_argument("error: DeletionAligner can only be used for alignments of empty strings");
    std::vector<handle_t> order = handlealgs::lazier_topological_order(&graph);
    std::unordered_map<handle_t, size_t, handle_hash> index_of;
    for (size_t i = 0; i < order.size(); ++i) index_of[order[i]] = i;
    const size_t inf = std::numeric_limits<size_t>::max();
    std::vector<size_t> dists(order.size(), inf);
    std::vector<std::pair<size_t, size_t>> sinks;
    for (size_t i = 0; i < order.size(); ++i) {
        if (dists[i] == inf) dists[i] = 0;
        const size_t thru = dists[i] + graph.get_length(order[i]);
        bool is_sink = true;
        graph.follow_edges_v(order[i], false, [&](const handle_t& next) { size_t j = index_of.at(next); dists[j] = std::min(dists[j], thru); is_sink = false; });
        if (is_sink) sinks.emplace_back(i, thru);
    }
    typedef std::vector<std::pair<size_t, size_t>> deflections_t;
    std::multiset<std::pair<size_t, deflections_t>> heap;           // min and max at either end, like structures::MinMaxHeap
    std::vector<std::vector<handle_t>> traces;
    auto propose = [&](size_t from, size_t to, size_t dist, const deflections_t& curr) {
        if (heap.size() + traces.size() < (size_t)max_alt_alns || (!heap.empty() && std::prev(heap.end())->first > dist)) {
            deflections_t d = curr; d.emplace_back(from, to);
            heap.emplace(dist, std::move(d));
            if (heap.size() + traces.size() > (size_t)max_alt_alns) heap.erase(std::prev(heap.end()));
        }
    };
    for (auto& sink : sinks) propose(order.size(), sink.first, sink.second, deflections_t());
    while (!heap.empty()) {
        const size_t trace_dist = heap.begin()->first; const deflections_t deflections = heap.begin()->second;
        heap.erase(heap.begin());
        traces.emplace_back();
        size_t deflxn = 0;
        auto get_next = [&](size_t at) -> size_t {
            if (deflxn < deflections.size() && at == deflections[deflxn].first) return deflections[deflxn++].second;
            size_t next = inf; const size_t dist_here = dists[at];
            graph.follow_edges_v(order[at], true, [&](const handle_t& prev) {
                const size_t idx = index_of.at(prev), dist_thru = dists[idx] + graph.get_length(prev);
                if (next == inf && dist_thru == dist_here) next = idx;
                else if (deflxn == deflections.size()) propose(at, idx, trace_dist - dist_here + dist_thru, deflections);
            });
            return next;
        };
        for (size_t tracer = get_next(order.size()); tracer != inf; tracer = get_next(tracer)) traces.back().push_back(order[tracer]);
    }
    if (order.empty() && max_alt_alns > 0) traces.emplace_back();
    for (const auto& trace : traces) {
        alt_alns.emplace_back();
        Alignment& a = alt_alns.back();
        a.sequence = aln.sequence; a.quality = aln.quality;
        int64_t total = 0;
        for (auto it = trace.rbegin(); it != trace.rend(); ++it) {
            a.path.mapping.emplace_back();
            Mapping& m = a.path.mapping.back();
            m.position.node_id = graph.get_id(*it); m.position.is_reverse = graph.get_is_reverse(*it);
            Edit e; e.from_length = (int32_t)graph.get_length(*it); e.to_length = 0; m.edit.push_back(e);
            total += e.from_length;
        }
        a.score = total ? -gap_open - (int32_t)(total - 1) * gap_extension : 0;
    }
    if (alt_alns.empty()) return;
    aln.path = alt_alns.front().path; aln.score = alt_alns.front().score;
}

std::unique_ptr<Aligner::Job> Aligner::prepare_banded_job(Alignment& alignment, const HandleGraph& g, int32_t band_padding, bool permissive_banding,
                                                          uint64_t max_cells) const {
    auto job = std::make_unique<Job>();
    Job& j = *job;
    j.alignment = &alignment; j.g = &g; j.banded = true; j.max_cells = max_cells;
    if (alignment.sequence.empty()) return job;                      // DeletionAligner's case: nothing for the engine (:703-706)
    // BandedGlobalAligner's constructor takes the graph in lazier_topological_order and the raw node sequences (:1976, :255)
    j.pg = create_packed_graph(g, handlealgs::lazier_topological_order(&g), /*raw_sequence=*/true);
    vgk_banded_problem& p = j.bprob;
    p.read = alignment.sequence.c_str(); p.read_len = (uint32_t)alignment.sequence.size();
    p.qual = quality_of(qual_adjusted, alignment.quality, alignment.sequence.size());
    p.flags = permissive_banding ? VGK_BANDED_PERMISSIVE : 0u;
    p.graph = j.pg.view(); p.band_padding = band_padding;
    p.max_cells = max_cells == std::numeric_limits<uint64_t>::max() ? 0 : max_cells;
    j.has_problem = true;
    return job;
}

void Aligner::finish_banded_job(Job& j, const vgk_result& res, const vgk_op* ops) const {
    Alignment& alignment = *j.alignment;
    if (!j.has_problem) { DeletionAligner(scorer->gap_open, scorer->gap_extension).align(alignment, *j.g); return; }
    if (res.status == VGK_ENOBAND) throw NoAlignmentInBandException();
    if (res.status == VGK_ETOOBIG) throw BandMatricesTooBigException("error:[BandedGlobalAligner] band matrices exceed the limit of " + std::to_string(j.max_cells) + " cells");
    if (res.status != VGK_OK) throw std::runtime_error(std::string("vgamd: banded global alignment failed: ") + engine->strerror(res.status));
    banded_ops_to_alignment(j.pg.order, *j.g, res, ops, alignment);
}

void Aligner::align_global_banded(Alignment& alignment, const HandleGraph& g, int32_t band_padding, bool permissive_banding,
                                  uint64_t max_cells) const {
    auto job = prepare_banded_job(alignment, g, band_padding, permissive_banding, max_cells);
    vgk_result res{};
    std::vector<vgk_op> ops;
    if (job->has_problem) {
        ops.resize(alignment.sequence.size() + job->pg.seq.size() + 2 * job->pg.order.size() + 8);
        size_t n_ops = 0;
        int rc = engine->banded_align(ctx, &job->bprob, 1, &res, ops.data(), ops.size(), &n_ops);
        if (rc != VGK_OK && res.status == VGK_OK) throw std::runtime_error(std::string("vgamd: banded global alignment failed: ") + engine->strerror(rc));
    }
    finish_banded_job(*job, res, ops.data() + res.ops_begin);
}

void Aligner::align_global_banded_multi(Alignment& alignment, std::vector<Alignment>& alt_alignments, const HandleGraph& g, int32_t max_alt_alns,
                                        int32_t band_padding, bool permissive_banding, uint64_t max_cells) const {
    if (!alt_alignments.empty()) throw std::invalid_argument("error:[Aligner] alternate alignment vector must be empty before aligning");   // (:691-694)
    if (max_alt_alns <= 0) throw std::invalid_argument("error:[Aligner] cannot do multi-alignment with max_alt_alns <= 0");
    if (alignment.sequence.empty()) {                       // (:767-771)
        DeletionAligner(scorer->gap_open, scorer->gap_extension).align_multi(alignment, alt_alignments, g, max_alt_alns);
        return;
    }
    PackedGraph pg = create_packed_graph(g, handlealgs::lazier_topological_order(&g), /*raw_sequence=*/true);
    vgk_banded_problem p{};
    p.read = alignment.sequence.c_str(); p.read_len = (uint32_t)alignment.sequence.size();
    p.qual = quality_of(qual_adjusted, alignment.quality, alignment.sequence.size());
    p.flags = permissive_banding ? VGK_BANDED_PERMISSIVE : 0u;
    p.graph = pg.view(); p.band_padding = band_padding;
    p.max_cells = max_cells == std::numeric_limits<uint64_t>::max() ? 0 : max_cells;
    std::vector<vgk_result> res((size_t)max_alt_alns);
    std::vector<vgk_op> ops((alignment.sequence.size() + pg.seq.size() + 2 * pg.order.size() + 8) * (size_t)max_alt_alns);
    uint32_t n_alns = 0; size_t n_ops = 0;
    int rc = engine->banded_align_multi(ctx, &p, 1, (uint32_t)max_alt_alns, res.data(), &n_alns, ops.data(), ops.size(), &n_ops);
    if (res[0].status == VGK_ENOBAND) throw NoAlignmentInBandException();
    if (res[0].status == VGK_ETOOBIG) throw BandMatricesTooBigException("error:[BandedGlobalAligner] band matrices exceed the limit of " + std::to_string(max_cells) + " cells");
    if (rc != VGK_OK || n_alns == 0) throw std::runtime_error(std::string("vgamd: banded global multi-alignment failed: ") + engine->strerror(rc ? rc : res[0].status));
    // the optimal alignment goes into both the main object and the first position of the vector (:2414-2419)
    for (uint32_t k = 0; k < n_alns; ++k) {
        alt_alignments.emplace_back();
        Alignment& a = alt_alignments.back();
        a.sequence = alignment.sequence; a.quality = alignment.quality;
        banded_ops_to_alignment(pg.order, g, res[k], ops.data() + res[k].ops_begin, a);
    }
    alignment.path = alt_alignments.front().path; alignment.score = alt_alignments.front().score; alignment.identity = alt_alignments.front().identity;
}

// BABuilder's edits (src/banded_global_aligner.cpp:102-205): one mapping per node at offset 0, matches split from
// mismatches by comparing the raw sequences, an empty edit on an empty node
void GSSWAligner::banded_ops_to_alignment(const std::vector<handle_t>& order, const HandleGraph& g, const vgk_result& res, const vgk_op* ops,
                                          Alignment& alignment) {
    alignment.clear_path();
    alignment.score = res.score;
    const std::string& read = alignment.sequence;
    size_t to_pos = 0;
    for (uint32_t i = 0; i < res.n_ops;) {
        uint32_t node = ops[i].node, j = i;
        while (j < res.n_ops && ops[j].node == node) ++j;
        const handle_t h = order[node];
        const std::string node_seq = g.get_sequence(h);
        alignment.path.mapping.emplace_back();
        Mapping& mapping = alignment.path.mapping.back();
        mapping.position.node_id = g.get_id(h); mapping.position.offset = 0;
        mapping.rank = (int64_t)alignment.path.mapping.size();
        size_t from_pos = 0;
        for (uint32_t k = i; k < j; ++k) {
            const size_t len = ops[k].len;
            if (len == 0) { mapping.edit.emplace_back(); continue; }
            switch (ops[k].op) {
                case VGK_OP_M:
                    for (size_t a = 0; a < len;) {
                        const bool matching = node_seq[from_pos + a] == read[to_pos + a];
                        size_t b = a + 1;
                        while (b < len && (node_seq[from_pos + b] == read[to_pos + b]) == matching) ++b;
                        Edit e; e.from_length = e.to_length = (int32_t)(b - a);
                        if (!matching) e.sequence = read.substr(to_pos + a, b - a);
                        mapping.edit.push_back(e);
                        a = b;
                    }
                    from_pos += len; to_pos += len; break;
                case VGK_OP_I: { Edit e; e.from_length = 0; e.to_length = (int32_t)len; e.sequence = read.substr(to_pos, len); mapping.edit.push_back(e); to_pos += len; } break;
                case VGK_OP_D: { Edit e; e.from_length = (int32_t)len; e.to_length = 0; mapping.edit.push_back(e); from_pos += len; } break;
                default: throw std::runtime_error("vgamd: unsupported op from the banded engine");
            }
        }
        i = j;
    }
    alignment.identity = identity(alignment.path);
}

std::string alignment_to_json(const Alignment& a) {
    std::ostringstream o;
    o << "{\"score\":" << a.score << ",\"identity\":" << a.identity << ",\"query_position\":" << a.query_position << ",\"sequence\":";
    json_escape(o, a.sequence);
    o << ",\"path\":{\"mapping\":[";
    for (size_t i = 0; i < a.path.mapping.size(); ++i) {
        const Mapping& m = a.path.mapping[i];
        if (i) o << ',';
        o << "{\"position\":{\"node_id\":" << m.position.node_id << ",\"offset\":" << m.position.offset
          << ",\"is_reverse\":" << (m.position.is_reverse ? "true" : "false") << "},\"rank\":" << m.rank << ",\"edit\":[";
        for (size_t j = 0; j < m.edit.size(); ++j) {
            if (j) o << ',';
            o << "{\"from_length\":" << m.edit[j].from_length << ",\"to_length\":" << m.edit[j].to_length << ",\"sequence\":";
            json_escape(o, m.edit[j].sequence);
            o << '}';
        }
        o << "]}";
    }
    o << "]}}";
    return o.str();
}

}  // namespace vgamd

// ---------------------------------------------------------------------------------------------------------
// Seeded two-pass X-drop alignment (DozeuInterface::align, src/dozeu_interface.cpp:608-685): a first pinned
// extension from the seed finds the "head" (position of the maximum), a second pinned extension from the
// head in the opposite direction is traced back.  Both passes run on the engine's VGK_XDROP_PINNED mode over
// a sub-DAG cut at the start position (dozeu is handed `seq + ref_offset`, src/dozeu_interface.cpp:236-243).
// ---------------------------------------------------------------------------------------------------------
namespace vgamd {

void Aligner::xdrop_extend_prepare(const HandleGraph& g, const std::vector<handle_t>& order, size_t node_index, size_t ref_offset,
                                   const std::string& read, const std::string& quality, size_t query_offset, bool right_to_left,
                                   bool traceback, uint16_t max_gap_length, ExtensionJob& job) const {
    job = ExtensionJob();
    Extension& ext = job.ext;
    ext.end_node = node_index; ext.end_ref_offset = ref_offset; ext.end_query = query_offset;
    job.node_index = node_index; job.ref_offset = ref_offset; job.query_offset = query_offset; job.right_to_left = right_to_left; job.traceback = traceback;
    std::string& query = job.query; std::string& qqual = job.qqual; PackedGraph& pg = job.pg; std::vector<size_t>& kept = job.kept;
    std::unordered_map<handle_t, size_t, handle_hash> index_of;
    for (size_t i = 0; i < order.size(); ++i) index_of[order[i]] = i;
    // the part of the start node that lies in the extension direction, and the read part to consume
    std::string start_seq = g.get_sequence(order[node_index]);
    const bool have_q = qual_adjusted && quality.size() == read.size();
    if (!right_to_left) { start_seq = start_seq.substr(ref_offset); query = read.substr(query_offset); if (have_q) qqual = quality.substr(query_offset); }
    else { start_seq = start_seq.substr(0, ref_offset); std::reverse(start_seq.begin(), start_seq.end());
           query = read.substr(0, query_offset); std::reverse(query.begin(), query.end());
           if (have_q) { qqual = quality.substr(0, query_offset); std::reverse(qqual.begin(), qqual.end()); } }
    if (query.empty()) return;
    // nodes reachable from the start, in extension order (the caller's order, reversed for a leftward pass)
    std::vector<char> reach(order.size(), 0);
    reach[node_index] = 1;
    std::vector<size_t> sub;                       // sub index -> index in `order`
    const int64_t inc = right_to_left ? -1 : 1;
    for (int64_t i = (int64_t)node_index; i >= 0 && i < (int64_t)order.size(); i += inc) {
        if (i != (int64_t)node_index) {
            bool r = false;
            g.follow_edges_v(order[i], !right_to_left, [&](const handle_t& nb) { auto it = index_of.find(nb); if (it != index_of.end() && reach[it->second]) r = true; });
            reach[i] = r;
        }
        if (reach[i]) sub.push_back((size_t)i);
    }
    const bool skip_start = start_seq.empty();     // pinned exactly at the node's end: its neighbours start from the root
    std::unordered_map<size_t, uint32_t> sub_of;
    for (size_t i : sub) { if (i == node_index && skip_start) continue; sub_of[i] = (uint32_t)kept.size(); kept.push_back(i); }
    if (kept.empty()) return;
    pg.pred_off.push_back(0);
    for (size_t i : kept) {
        std::string s = (i == node_index) ? start_seq : g.get_sequence(order[i]);
        if (right_to_left && i != node_index) std::reverse(s.begin(), s.end());
        if (s.empty()) throw std::runtime_error("vgamd: empty node inside an X-drop extension (mask it with DozeuPinningOverlay first)");
        pg.order.push_back(order[i]); pg.node_len.push_back((uint32_t)s.size()); pg.seq += s;
        if (i != node_index) g.follow_edges_v(order[i], !right_to_left, [&](const handle_t& nb) {
            auto it = index_of.find(nb);
            if (it == index_of.end() || !reach[it->second]) return;
            auto st = sub_of.find(it->second);
            if (st != sub_of.end()) pg.pred_idx.push_back(st->second);      // (a skipped start node leaves its neighbours as sources)
        });
        pg.pred_off.push_back((uint32_t)pg.pred_idx.size());
    }
    vgk_gssw_problem& prob = job.prob;
    prob.read = query.data(); prob.read_len = (uint32_t)query.size();
    prob.qual = quality_of(qual_adjusted, qqual, query.size());
    prob.flags = VGK_XDROP_PINNED | (traceback ? VGK_GSSW_TRACEBACK : 0);
    prob.graph = pg.view(); prob.max_gap_length = std::max<uint16_t>(max_gap_length, 1);
    job.runs = true;
}

Aligner::Extension Aligner::xdrop_extend_finish(const HandleGraph& g, const std::vector<handle_t>& order, const std::string& read, ExtensionJob& job,
                                                const vgk_result& res_in, std::vector<vgk_op>& ops) const {
    vgk_result res = res_in;                       // (a leftward pass flips its own copy back)
    Extension ext = job.ext;
    const size_t node_index = job.node_index, ref_offset = job.ref_offset, query_offset = job.query_offset; const bool right_to_left = job.right_to_left, traceback = job.traceback;
    const PackedGraph& pg = job.pg; const std::vector<size_t>& kept = job.kept;
    if (!job.runs) return ext;
    ops.resize(res.n_ops);
    ext.score = res.score;
    if (res.score <= 0) return ext;
    // position of the maximum, back in the caller's coordinates (exclusive ends / remaining prefix lengths)
    const size_t en = kept[(size_t)res.end_node];
    const size_t used = (size_t)res.end_offset + 1;               // bases of that node consumed in extension direction
    ext.end_node = en;
    if (!right_to_left) { ext.end_ref_offset = (en == node_index ? ref_offset : 0) + used; ext.end_query = query_offset + (size_t)res.end_read + 1; }
    else { ext.end_ref_offset = (en == node_index ? ref_offset : g.get_length(order[en])) - used; ext.end_query = query_offset - ((size_t)res.end_read + 1); }
    if (!traceback) return ext;
    // ops -> mappings in read order.  A leftward pass ran on reversed strings: flip it back (as for a right pin).
    if (right_to_left) unreverse_ops(ops, res, pg.node_len);
    size_t to_pos = right_to_left ? 0 : query_offset;
    int from_pos = right_to_left ? res.first_offset : 0;
    uint32_t i = 0; bool first_node = true;
    while (i < ops.size()) {
        uint32_t node = ops[i].node, j = i;
        while (j < ops.size() && ops[j].node == node) ++j;
        const size_t oi = kept[node];
        const std::string node_seq = g.get_sequence(order[oi]);
        // where this mapping starts on the original node
        int node_from;
        if (!right_to_left) node_from = (first_node && oi == node_index) ? (int)ref_offset : 0;
        else node_from = first_node ? from_pos : 0;
        first_node = false;
        ext.mappings.emplace_back();
        Mapping& mapping = ext.mappings.back();
        mapping.position.node_id = g.get_id(order[oi]); mapping.position.is_reverse = g.get_is_reverse(order[oi]);
        mapping.position.offset = node_from;
        int fp = node_from;
        for (uint32_t k = i; k < j; ++k) {
            const int32_t len = ops[k].len;
            switch (ops[k].op) {
                case VGK_OP_M: {
                    int run = 0;
                    for (int t = 0; t < len; ++t) {
                        if (node_seq[fp + t] == read[to_pos + t]) { ++run; ++ext.matches; }
                        else {
                            if (run) { Edit e; e.from_length = e.to_length = run; mapping.edit.push_back(e); run = 0; }
                            Edit e; e.from_length = e.to_length = 1; e.sequence = read.substr(to_pos + t, 1); mapping.edit.push_back(e);
                        }
                    }
                    if (run) { Edit e; e.from_length = e.to_length = run; mapping.edit.push_back(e); }
                    fp += len; to_pos += len;
                } break;
                case VGK_OP_D: { Edit e; e.from_length = len; e.to_length = 0; mapping.edit.push_back(e); fp += len; } break;
                case VGK_OP_I:
                case VGK_OP_S: {
                    const bool merge_prev = !mapping.edit.empty() && edit_is_insertion(mapping.edit.back());
                    if (merge_prev) { Edit& e = mapping.edit.back(); e.to_length += len; e.sequence += read.substr(to_pos, len); }
                    else { Edit e; e.from_length = 0; e.to_length = len; e.sequence = read.substr(to_pos, len); mapping.edit.push_back(e); }
                    to_pos += len;
                } break;
                default: throw std::runtime_error("vgamd: unsupported cigar op from engine");
            }
        }
        i = j;
    }
    return ext;
}


Aligner::Extension Aligner::xdrop_extend(const HandleGraph& g, const std::vector<handle_t>& order, size_t node_index,
                                         size_t ref_offset, const std::string& read, const std::string& quality, size_t query_offset,
                                         bool right_to_left, bool traceback, uint16_t max_gap_length) const {
    ExtensionJob job;
    xdrop_extend_prepare(g, order, node_index, ref_offset, read, quality, query_offset, right_to_left, traceback, max_gap_length, job);
    if (!job.runs) return job.ext;
    vgk_result res{}; std::vector<vgk_op> ops(job.prob.read_len + job.pg.seq.size() + job.kept.size() + 4);
    size_t written = 0;
    int rc = xdrop_band ? engine->xdrop_band_align(ctx, &job.prob, 1, &res, ops.data(), ops.size(), &written, nullptr)
                        : engine->gssw_align(ctx, &job.prob, 1, &res, ops.data(), ops.size(), &written);
    if (rc != VGK_OK || res.status != VGK_OK)
        throw std::runtime_error(std::string("vgamd: xdrop engine failed: ") + engine->strerror(rc ? rc : res.status));
    return xdrop_extend_finish(g, order, read, job, res, ops);
}

void Aligner::align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<MaximalExactMatch>& mems,
                          bool reverse_complemented, uint16_t max_gap_length) const {
    align_xdrop(alignment, g, handlealgs::lazier_topological_order(&g), mems, reverse_complemented, max_gap_length);
}

void Aligner::align_xdrop(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order,
                          const std::vector<MaximalExactMatch>& mems, bool reverse_complemented, uint16_t max_gap_length) const {
    xdrop_align(alignment, g, order, mems, reverse_complemented, max_gap_length);
    if (!alignment.has_path() && mems.empty()) {
        // dozeu couldn't find an alignment, probably because its seeding heuristic failed: fall back on gssw
        // (bonus at both ends there, once in dozeu — known inconsistency, src/aligner.cpp:848-854)
        align(alignment, g, order);
    }
}

// scan_seed_position (src/dozeu_interface.cpp:143-208): locate the best match of the 15-base read end nearest the far side.
// dz_scan's exact rules are in the missing dozeu source [PARITY-UNPINNED]; the engine's local mode on
// the same 15 bases finds the same maximum position whenever the best hit is an end-to-end match.
void Aligner::xdrop_scan_prepare(const Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order, bool direction, ScanJob& job) const {
    job = ScanJob();
    const std::string& read = alignment.sequence;
    const size_t qlen = read.size(); job.scan_len = std::min<size_t>(qlen, 15);
    std::string& tail = job.tail; std::string& tail_q = job.tail_q; PackedGraph& pg = job.pg;
    tail = direction ? read.substr(0, job.scan_len) : read.substr(qlen - job.scan_len);
    if (qual_adjusted && alignment.quality.size() == qlen) tail_q = direction ? alignment.quality.substr(0, job.scan_len) : alignment.quality.substr(qlen - job.scan_len);
    pg.order = order; pg.pred_off.push_back(0);
    const HandleGraph* sg = &g; ReverseGraph rg(&g, false);
    std::vector<handle_t>& run_order = job.run_order; run_order = order;
    if (direction) { sg = &rg; std::reverse(run_order.begin(), run_order.end()); std::reverse(tail.begin(), tail.end()); std::reverse(tail_q.begin(), tail_q.end()); pg.order = run_order; }
    std::unordered_map<handle_t, uint32_t, handle_hash> ridx;
    for (uint32_t i = 0; i < run_order.size(); ++i) ridx[run_order[i]] = i;
    for (uint32_t i = 0; i < run_order.size(); ++i) {
        std::string s = sg->get_sequence(run_order[i]);
        pg.node_len.push_back((uint32_t)s.size()); pg.seq += s;
        sg->follow_edges_v(run_order[i], true, [&](const handle_t& p) { auto it = ridx.find(p); if (it != ridx.end() && it->second < i) pg.pred_idx.push_back(it->second); });
        pg.pred_off.push_back((uint32_t)pg.pred_idx.size());
    }
    vgk_gssw_problem& prob = job.prob;
    prob.read = tail.data(); prob.read_len = (uint32_t)tail.size(); prob.flags = VGK_GSSW_LOCAL; prob.graph = pg.view();
    prob.qual = quality_of(qual_adjusted, tail_q, tail.size());
}

// what follows the traced extension from the head (align_downward's tail, src/dozeu_interface.cpp:687-722)
void Aligner::xdrop_finish(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order, Extension& down, size_t head_node, size_t head_ref,
                           size_t head_query, bool direction) const {
    const std::string& read = alignment.sequence;
    alignment.score = down.score;
    alignment.query_position = 0;
    if (down.score <= 0 || down.mappings.empty()) {
        // full-length insertion at the head position (:344-359)
        alignment.score = 0;
        alignment.path.mapping.emplace_back();
        Mapping& m = alignment.path.mapping.back();
        m.position.node_id = g.get_id(order[head_node]); m.position.is_reverse = g.get_is_reverse(order[head_node]);
        m.position.offset = (int64_t)head_ref; m.rank = 1;
        Edit e; e.from_length = 0; e.to_length = (int32_t)read.size(); e.sequence = read; m.edit.push_back(e);
        return;
    }
    alignment.path.mapping = std::move(down.mappings);
    for (size_t i = 0; i < alignment.path.mapping.size(); ++i) alignment.path.mapping[i].rank = (int64_t)i + 1;
    // the read part on the far side of the head was never shown to dozeu: it becomes an insertion (:498-526)
    if (!direction) {            // aligned read[0, head_query); the rest trails
        if (head_query < read.size()) {
            Edit e; e.from_length = 0; e.to_length = (int32_t)(read.size() - head_query); e.sequence = read.substr(head_query);
            alignment.path.mapping.back().edit.push_back(e);
        }
    } else if (head_query > 0) { // aligned read[head_query, end); the part before it leads
        Mapping& m = alignment.path.mapping.front();
        if (!m.edit.empty() && edit_is_insertion(m.edit.front())) {
            m.edit.front().to_length += (int32_t)head_query; m.edit.front().sequence = read.substr(0, head_query) + m.edit.front().sequence;
        } else { Edit e; e.from_length = 0; e.to_length = (int32_t)head_query; e.sequence = read.substr(0, head_query); m.edit.insert(m.edit.begin(), e); }
    }
    alignment.identity = (double)down.matches / (double)read.size();
}

void Aligner::xdrop_align(Alignment& alignment, const HandleGraph& g, const std::vector<handle_t>& order,
                          const std::vector<MaximalExactMatch>& mems, bool reverse_complemented, uint16_t max_gap_length) const {
    std::vector<XdropRequest> one(1);
    one[0].alignment = &alignment; one[0].graph = &g; one[0].order = order; one[0].mems = mems; one[0].reverse_complemented = reverse_complemented; one[0].max_gap_length = max_gap_length;
    xdrop_align_many(one);
}

// DozeuInterface::align (src/dozeu_interface.cpp:608-685) for many requests: the passes of all of them side by side.
void Aligner::xdrop_align_many(std::vector<XdropRequest>& requests) const {
    struct State { bool live = false, have_head = false, direction = false; size_t head_node = 0, head_ref = 0, head_query = 0;
                   std::unordered_map<handle_t, size_t, handle_hash> index_of; std::unique_ptr<ScanJob> scan; std::unique_ptr<ExtensionJob> up, down; };
    const size_t n = requests.size();
    std::vector<State> st(n);
    auto lap_t0 = std::chrono::steady_clock::now(); const bool lap_on = std::getenv("VGAMD_TIMING") != nullptr;
    auto lap = [&](const char* what) { if (!lap_on) return; const auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[align_xdrop_many] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - lap_t0).count()); lap_t0 = t; };
    // one engine call over the problems of a pass; results and ops per problem come back through `take`
    auto run = [&](std::vector<vgk_gssw_problem>& probs, bool band, std::vector<vgk_result>& res, std::vector<vgk_op>& ops, const char* what) {
        res.assign(probs.size(), vgk_result{});
        size_t cap = 16; for (const vgk_gssw_problem& p : probs) { size_t bases = 0; for (uint32_t v = 0; v < p.graph.n_nodes; ++v) bases += p.graph.node_len[v]; cap += p.read_len + bases + p.graph.n_nodes + 4; }
        ops.assign(cap, vgk_op{});
        size_t written = 0;
        if (probs.empty()) return;
        const int rc = band ? engine->xdrop_band_align(ctx, probs.data(), (uint32_t)probs.size(), res.data(), ops.data(), ops.size(), &written, nullptr)
                            : engine->gssw_align(ctx, probs.data(), (uint32_t)probs.size(), res.data(), ops.data(), ops.size(), &written);
        if (rc != VGK_OK) throw std::runtime_error(std::string("vgamd: ") + what + " failed: " + engine->strerror(rc));
        for (const vgk_result& r : res) if (r.status != VGK_OK) throw std::runtime_error(std::string("vgamd: ") + what + " failed: " + engine->strerror(r.status));
    };
    auto ops_of = [](const vgk_result& r, const std::vector<vgk_op>& all) { return std::vector<vgk_op>(all.begin() + r.ops_begin, all.begin() + r.ops_begin + r.n_ops); };
    // The per-request halves around the engine calls are independent of each other: with more than a handful of requests they run on host
    // threads (a rescue batch is tens of thousands of mates; serially these loops were 0.5 s of a 0.7 s step)
    auto each = [&](size_t count, const std::function<void(size_t)>& body) {
        unsigned threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = std::getenv("VGAMD_HOST_THREADS")) threads = (unsigned)std::max(1, std::atoi(e));
        if (count < 64 || threads < 2) { for (size_t k = 0; k < count; ++k) body(k); return; }
        threads = (unsigned)std::min<size_t>(threads, (count + 31) / 32);
        std::atomic<size_t> next{0}; std::exception_ptr err; std::mutex err_mu;
        auto work = [&]() {
            try { for (size_t i; (i = next.fetch_add(32)) < count;) for (size_t k = i; k < std::min(count, i + 32); ++k) body(k); }
            catch (...) { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); next.store(count); }
        };
        std::vector<std::thread> ts;
        for (unsigned t = 1; t < threads; ++t) ts.emplace_back(work);
        work();
        for (auto& t : ts) t.join();
        if (err) std::rethrow_exception(err);
    };
    // ---- first pass: the head position (:629-673) — a scan of the read's last bases, or the extension from the seed towards it
    std::vector<vgk_gssw_problem> scans, ups; std::vector<size_t> scan_of, up_of;
    each(n, [&](size_t k) {
        XdropRequest& rq = requests[k]; State& s = st[k];
        Alignment& alignment = *rq.alignment; const HandleGraph& g = *rq.graph;
        if (rq.order.empty()) rq.order = handlealgs::lazier_topological_order(&g);
        alignment.clear_path();
        if (rq.order.empty() || alignment.sequence.empty()) return;
        s.live = true; s.direction = rq.reverse_complemented;
        for (size_t i = 0; i < rq.order.size(); ++i) s.index_of[rq.order[i]] = i;
        if (rq.mems.empty()) {
            s.scan = std::make_unique<ScanJob>();
            xdrop_scan_prepare(alignment, g, rq.order, s.direction, *s.scan);
        } else {
            // calculate_seed_position (:75-114)
            const MaximalExactMatch& seed = s.direction ? rq.mems.back() : rq.mems.front();
            const MaximalExactMatch::Hit& hit = s.direction ? seed.nodes.front() : seed.nodes.back();
            const size_t sn = s.index_of.at(g.get_handle(hit.id, hit.is_reverse));
            const size_t sref = s.direction ? g.get_length(rq.order[sn]) - hit.offset : hit.offset;
            const size_t squery = s.direction ? alignment.sequence.size() - seed.begin : seed.begin;
            // "upward" extension from the seed; its maximum is the head (:654-672)
            s.up = std::make_unique<ExtensionJob>();
            xdrop_extend_prepare(g, rq.order, sn, sref, alignment.sequence, alignment.quality, squery, s.direction, false, rq.max_gap_length, *s.up);
            if (!s.up->runs) { s.head_node = s.up->ext.end_node; s.head_ref = s.up->ext.end_ref_offset; s.head_query = s.up->ext.end_query; s.have_head = true; }
        }
    });
    for (size_t k = 0; k < n; ++k) {                                     // the passes' problem lists, in request order
        State& s = st[k];
        if (!s.live) continue;
        if (s.scan) { scans.push_back(s.scan->prob); scan_of.push_back(k); }
        else if (s.up && s.up->runs) { ups.push_back(s.up->prob); up_of.push_back(k); }
    }
    lap("first pass prepared");
    std::vector<vgk_result> res; std::vector<vgk_op> ops;
    run(scans, false, res, ops, "scan");
    lap("scan engine call");
    for (size_t q = 0; q < scan_of.size(); ++q) {
        const size_t k = scan_of[q]; State& s = st[k]; const XdropRequest& rq = requests[k]; const vgk_result& r = res[q];
        if (r.score <= 0) continue;      // scan failed: the path stays empty, the caller falls back to gssw (src/aligner.cpp:848-854)
        const handle_t h = s.scan->run_order[(size_t)r.end_node];
        const size_t qlen = rq.alignment->sequence.size(), scan_len = s.scan->scan_len;
        s.have_head = true; s.head_node = s.index_of.at(h);
        const size_t used = (size_t)r.end_offset + 1, qused = (size_t)r.end_read + 1;
        if (!s.direction) { s.head_ref = used; s.head_query = (qlen - scan_len) + qused; }
        else { s.head_ref = rq.graph->get_length(h) - used; s.head_query = scan_len - qused; }
    }
    run(ups, xdrop_band, res, ops, "xdrop engine");
    lap("seed-side engine call");
    each(up_of.size(), [&](size_t q) {
        const size_t k = up_of[q]; State& s = st[k]; const XdropRequest& rq = requests[k];
        std::vector<vgk_op> mine = ops_of(res[q], ops);
        const Extension up = xdrop_extend_finish(*rq.graph, rq.order, rq.alignment->sequence, *s.up, res[q], mine);
        s.head_node = up.end_node; s.head_ref = up.end_ref_offset; s.head_query = up.end_query; s.have_head = true;
    });
    // ---- second pass: the traced extension from the head the other way (align_downward, :687-722)
    std::vector<vgk_gssw_problem> downs; std::vector<size_t> down_of;
    each(n, [&](size_t k) {
        State& s = st[k]; const XdropRequest& rq = requests[k];
        if (!s.live || !s.have_head) return;
        s.down = std::make_unique<ExtensionJob>();
        xdrop_extend_prepare(*rq.graph, rq.order, s.head_node, s.head_ref, rq.alignment->sequence, rq.alignment->quality, s.head_query, !s.direction, true, rq.max_gap_length, *s.down);
    });
    for (size_t k = 0; k < n; ++k) if (st[k].down && st[k].down->runs) { downs.push_back(st[k].down->prob); down_of.push_back(k); }
    lap("second pass prepared");
    run(downs, xdrop_band, res, ops, "xdrop engine");
    lap("traced engine call");
    std::vector<char> answered(n, 0);
    each(down_of.size(), [&](size_t q) {
        const size_t k = down_of[q]; State& s = st[k]; const XdropRequest& rq = requests[k];
        std::vector<vgk_op> mine = ops_of(res[q], ops);
        Extension down = xdrop_extend_finish(*rq.graph, rq.order, rq.alignment->sequence, *s.down, res[q], mine);
        xdrop_finish(*rq.alignment, *rq.graph, rq.order, down, s.head_node, s.head_ref, s.head_query, s.direction);
        answered[k] = 1;
    });
    each(n, [&](size_t k) {
        State& s = st[k]; const XdropRequest& rq = requests[k];
        if (!s.live || !s.have_head || answered[k]) return;
        Extension down = s.down->ext;                                   // nothing ran: nothing of the read or the graph lies that way
        xdrop_finish(*rq.alignment, *rq.graph, rq.order, down, s.head_node, s.head_ref, s.head_query, s.direction);
    });
    lap("finished");
    // the jobs' arrays were allocated on the threads above; handing them back one request after another on this thread alone was a third of
    // a rescue batch's time
    each(n, [&](size_t k) { st[k] = State{}; });
    lap("released");
}

void Aligner::align_xdrop_many(std::vector<XdropRequest>& requests) const {
    xdrop_align_many(requests);
    for (XdropRequest& rq : requests)
        if (!rq.alignment->has_path() && rq.mems.empty())        // dozeu's seeding heuristic failed: gssw instead (src/aligner.cpp:848-854)
            align(*rq.alignment, *rq.graph, rq.order);
}

}  // namespace vgamd
