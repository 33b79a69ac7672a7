// tail_stage.cpp — see tail_stage.hpp.
#include "tail_stage.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

namespace vgamd {
namespace {

template <class F> void for_chunks(size_t n, F f, size_t chunk = 4096) {                       // f(lo, hi, chunk) on a few threads, chunks in index order
    unsigned T = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = std::getenv("VGAMD_HOST_THREADS")) T = (unsigned)std::max(1, std::atoi(e));
    const size_t chunks = (n + chunk - 1) / chunk;
    if (chunks <= 1 || T <= 1) { for (size_t c = 0; c < chunks; ++c) f(c * chunk, std::min(n, (c + 1) * chunk), c); return; }
    std::atomic<size_t> next{0};
    auto body = [&]() { for (;;) { const size_t c = next.fetch_add(1); if (c >= chunks) break; f(c * chunk, std::min(n, (c + 1) * chunk), c); } };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < std::min<size_t>(T, chunks); ++t) ts.emplace_back(body);
    body();
    for (auto& t : ts) t.join();
}

inline int64_t longest_detectable_gap(int match, int gap_open, int gap_extend, int bonus, int64_t read_length, int64_t read_pos) {
    const int64_t overhang = std::min(read_pos, read_length - read_pos);    // (src/alignment_scorer.cpp:264-271)
    const int64_t numer = match * overhang + bonus;
    const int64_t gap = (numer - gap_open) / gap_extend + 1;
    return (gap >= 0 && overhang > 0) ? gap : 0;
}
inline char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; } }

struct Tail { uint32_t ext, read; uint32_t begin, end; uint8_t left; int64_t gap; };
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point& t0) { const auto t = Clock::now(); const double ms = std::chrono::duration<double, std::milli>(t - t0).count(); t0 = t; return ms; }

}  // namespace

int run_tail_stage(const EngineApi& api, vgk_ctx* ctx, const vgk_haplo* index, const TailStageInput& in, TailStageOutput& out) {
    auto t0 = Clock::now();
    const uint32_t n = in.n_reads;
    // (the sets come in problem order, as vgk_gapless_extend hands them back: the last read's end is the number of extensions)
    const uint64_t n_ext = n ? (uint64_t)in.res[n - 1].ext_begin + in.res[n - 1].n_ext : 0;
    std::vector<uint32_t> read_of(n_ext);
    for_chunks(n, [&](size_t lo, size_t hi, size_t) { for (size_t i = lo; i < hi; ++i) for (uint32_t k = 0; k < in.res[i].n_ext; ++k) read_of[in.res[i].ext_begin + k] = (uint32_t)i; });
    // which tails exist: right tails of every extension first, then left tails (two passes over the extensions, each in index order)
    const size_t chunks = (n_ext + 4095) / 4096;
    std::vector<std::vector<Tail>> part_r(chunks), part_l(chunks);
    std::vector<std::vector<vgk_tail_problem>> prob_r(chunks), prob_l(chunks);
    for_chunks(n_ext, [&](size_t lo, size_t hi, size_t c) {
        for (size_t e = lo; e < hi; ++e) {
            const vgk_extension& x = in.ext[e]; const uint32_t r = read_of[e];
            if (in.res[r].status != VGK_OK || in.res[r].full_length || !x.path_len) continue;      // full-length sets are scored as they are (:5440)
            const int64_t L = (int64_t)(in.read_off[r + 1] - in.read_off[r]);
            if (!x.right_full) {                                            // look right from the end, forward state (:5768-5775)
                uint64_t before_last = 0;
                for (uint32_t k = 0; k + 1 < x.path_len; ++k) before_last += in.oriented_len[in.nodes[x.path_begin + k]];
                const int64_t tail = L - x.read_end;
                const int64_t gap = longest_detectable_gap(in.match, in.gap_open, in.gap_extend, in.bonus, L, tail);
                vgk_tail_problem p; p.node = x.state[0]; p.lo = (int32_t)x.state[1]; p.hi = (int32_t)x.state[2];
                p.offset = (uint32_t)(x.offset + (x.read_end - x.read_begin) - before_last); p.walk_distance = (uint32_t)(tail + gap);
                prob_r[c].push_back(p); part_r[c].push_back(Tail{(uint32_t)e, r, x.read_end, (uint32_t)L, 0, gap});
            }
            if (!x.left_full) {                                             // look the other way from the start, backward state (:5756-5766)
                const uint32_t first = in.nodes[x.path_begin] ^ 1u;
                const int64_t tail = x.read_begin;
                const int64_t gap = longest_detectable_gap(in.match, in.gap_open, in.gap_extend, in.bonus, L, tail);
                vgk_tail_problem p; p.node = x.state[3]; p.lo = (int32_t)x.state[4]; p.hi = (int32_t)x.state[5];
                p.offset = in.oriented_len[first] - x.offset; p.walk_distance = (uint32_t)(tail + gap);
                prob_l[c].push_back(p); part_l[c].push_back(Tail{(uint32_t)e, r, 0, x.read_begin, 1, gap});
            }
        }
    });
    std::vector<size_t> at_r(chunks + 1, 0), at_l(chunks + 1, 0);
    for (size_t c = 0; c < chunks; ++c) { at_r[c + 1] = at_r[c] + part_r[c].size(); at_l[c + 1] = at_l[c] + part_l[c].size(); }
    const size_t nt = at_r[chunks] + at_l[chunks];
    std::vector<Tail> tails(nt); std::vector<vgk_tail_problem> problems(nt);
    for_chunks(chunks, [&](size_t lo, size_t hi, size_t) {
        for (size_t c = lo; c < hi; ++c) {
            std::copy(part_r[c].begin(), part_r[c].end(), tails.begin() + (long)at_r[c]); std::copy(prob_r[c].begin(), prob_r[c].end(), problems.begin() + (long)at_r[c]);
            std::copy(part_l[c].begin(), part_l[c].end(), tails.begin() + (long)(at_r[chunks] + at_l[c])); std::copy(prob_l[c].begin(), prob_l[c].end(), problems.begin() + (long)(at_r[chunks] + at_l[c]));
        }
    }, 8);
    out.n_tails = nt; out.tail_score.assign(nt, 0);
    out.ext_total.resize(n_ext);
    for_chunks(n_ext, [&](size_t lo, size_t hi, size_t) { for (size_t e = lo; e < hi; ++e) out.ext_total[e] = in.ext[e].score; });
    out.ms[0] = ms_since(t0);
    if (nt) {
        std::vector<vgk_tail_result> tres(nt);
        vgk_forest* forest = nullptr;
        int rc = api.tail_forest(ctx, index, problems.data(), (uint32_t)nt, tres.data(), &forest);
        if (rc) return rc;
        out.ms[1] = ms_since(t0);
        out.tree_nodes = api.forest_size(forest);
        // one window per tree; a tail whose walk skipped the root has several (:5838): split at the roots
        bool forests = false;
        for (size_t i = 0; i < nt; ++i) { if (tres[i].status != VGK_OK) ++out.failed; forests |= tres[i].n_trees > 1; }
        std::vector<int32_t> parent;
        if (forests) { parent.resize(out.tree_nodes); rc = api.forest_fetch(forest, parent.data(), nullptr, nullptr); if (rc) { api.forest_destroy(forest); return rc; } }
        std::vector<uint32_t> owner; std::vector<vgk_window_problem> wins;
        std::vector<uint64_t> seq_off(nt + 1, 0);
        for (size_t i = 0; i < nt; ++i) seq_off[i + 1] = seq_off[i] + (tails[i].end - tails[i].begin);
        std::vector<char> seq(seq_off[nt] + 1);
        for_chunks(nt, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; ++i) {
                const Tail& t = tails[i]; const char* rd = in.reads + in.read_off[t.read]; char* dst = seq.data() + seq_off[i];
                const uint32_t len = t.end - t.begin;
                if (!t.left) std::memcpy(dst, rd + t.begin, len);
                else for (uint32_t k = 0; k < len; ++k) dst[k] = comp(rd[t.end - 1 - k]);              // reverse_complement(sequence) (:5660)
            }
        });
        for (size_t i = 0; i < nt; ++i) {
            const vgk_tail_result& r = tres[i];
            if (r.status != VGK_OK || !r.n_nodes) continue;
            uint32_t a = r.first_node; const uint32_t end = r.first_node + r.n_nodes;
            while (a < end) {
                uint32_t b = end;
                if (r.n_trees > 1) { b = a + 1; while (b < end && parent[b] >= 0) ++b; }
                vgk_window_problem w; w.read_off = seq_off[i]; w.read_len = tails[i].end - tails[i].begin; w.flags = VGK_XDROP_PINNED | VGK_GSSW_TRACEBACK;
                w.first_node = a; w.n_nodes = b - a; w.max_gap_length = (uint32_t)tails[i].gap; w.reserved = 0;
                wins.push_back(w); owner.push_back((uint32_t)i);
                a = b;
            }
        }
        out.n_trees = wins.size();
        out.ms[2] = ms_since(t0);
        if (!wins.empty()) {
            vgk_batch* b = nullptr;
            rc = api.gssw_pack_windows(ctx, api.forest_graph(forest), seq.data(), seq_off[nt], wins.data(), (uint32_t)wins.size(), in.ops_per_problem, &b);
            if (rc) { api.forest_destroy(forest); return rc; }
            out.ms[3] = ms_since(t0);
            rc = api.gssw_run(b);
            std::vector<vgk_result> wr(wins.size());
            std::vector<vgk_op> ops((size_t)wins.size() * in.ops_per_problem + 1);
            size_t written = 0;
            if (!rc) rc = api.gssw_fetch(b, wr.data(), ops.data(), ops.size(), &written);
            out.ms[4] = ms_since(t0);
            api.batch_free(b);
            if (rc) { api.forest_destroy(forest); return rc; }
            for (size_t w = 0; w < wins.size(); ++w) {                  // the best tree of a tail; nothing aligned = the soft clip, 0 (:5632-5648)
                if (wr[w].status != VGK_OK) { ++out.failed; continue; }
                out.tail_score[owner[w]] = std::max(out.tail_score[owner[w]], wr[w].score);
            }
        }
        api.forest_destroy(forest);
        for (size_t i = 0; i < nt; ++i) out.ext_total[tails[i].ext] += out.tail_score[i];
    }
    out.read_score.resize(n);
    for_chunks(n, [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; ++i) {
            int32_t best = 0;
            for (uint32_t k = 0; k < in.res[i].n_ext; ++k) best = std::max(best, out.ext_total[in.res[i].ext_begin + k]);
            out.read_score[i] = best;
        }
    });
    out.ms[5] = ms_since(t0);
    return VGK_OK;
}

}  // namespace vgamd
