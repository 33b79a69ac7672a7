// wfa_api.cpp — vgk_wfa_extend: the host half of haplotype-consistent wavefront alignment (wfa_device.hpp).
//
// Per call: validate, mask the sequences (ReadMasker, reference src/gbwt_extender.cpp:160-170) — reverse-complemented for
// PREFIX problems, whose target becomes the start position on the other strand (:2248-2255) —, evaluate the error model's
// score cap and distance band per sequence length (:1631-1634, :2077; gbwt_extender.hpp:371-373), upload, launch one
// thread per problem over zero-initialised per-thread slabs, download, and hand paths / edits back in problem order.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "haplo.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {

struct WfaHost { PinnedBuf<char> seqs; PinnedBuf<uint32_t> src_off; PinnedBuf<WProb> probs; PinnedBuf<vgk_wfa_result> dres; PinnedBuf<uint32_t> dpaths, dedits; uint64_t zeroed_bytes = 0; void* zeroed_ptr = nullptr;
                 void* slab_ptr = nullptr; uint64_t slab_bytes = 0, slab_shape = 0;
                 GIndex walk_index{}; GMerge walk_merge{}; };      // the index the wavefront kernel walks in the current call: the caller's, or its merged-run form

const vgk_wfa_error_model kDefaultModel = { { 0.03, 1, 6 }, { 0.05, 1, 10 }, { 0.1, 1, 20 }, { 0.1, 10, 200 } };   // gbwt_extender.hpp:386-395

int32_t evaluate(const vgk_wfa_event& e, uint32_t length) {                                       // gbwt_extender.hpp:371-373
    return std::min(e.max, (int32_t)(e.per_base * length) + e.min);
}
char complement(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'X'; }
}

// The wavefront form's slabs: per resident wavefront the table, its log, the path pool and the backtrace's edit runs, carved out of one
// allocation per launch size; the table part must be all-zero before a launch (the kernel leaves it so).
struct WaveSlabs { uint32_t waves, n_slots, max_points, path_cap, mask_width; };
int carve(vgk_ctx* ctx, WfaHost& H, int slot, const WaveSlabs& z, WwParams& W) {
    Backend* be = ctx->be.get();
    const uint64_t w = z.waves;
    const uint64_t b_slots = 8ull * z.n_slots * w, b_logs = 4ull * z.max_points * w, b_paths = sizeof(WwPath) * (uint64_t)z.path_cap * w, b_runs = 4ull * W_EDITS * w;
    const uint64_t b_masks = 4ull * WW_MASK_ROWS * 3u * z.mask_width * w;           // the item filter's node sets (rows are cleared by the kernel before use)
    const uint64_t want = b_slots + b_logs + b_paths + b_runs + b_masks + 64;
    char* base = (char*)ctx->ensure_scratch(slot, want);
    if (!base) return VGK_ENOMEM;
    if (H.slab_ptr != base || H.slab_bytes < want || H.slab_shape != ((uint64_t)z.n_slots << 32 | (uint64_t)z.mask_width << 20 | z.waves)) {
        if (be->zero(base, ctx->scratch[slot].bytes)) return VGK_ENODEV;
        H.slab_ptr = base; H.slab_bytes = ctx->scratch[slot].bytes; H.slab_shape = (uint64_t)z.n_slots << 32 | (uint64_t)z.mask_width << 20 | z.waves;
    }
    W.slots = (unsigned long long*)base; W.n_slots = z.n_slots;
    W.paths = (WwPath*)(base + b_slots); W.path_cap = z.path_cap;
    W.logs = (uint32_t*)(base + b_slots + b_paths); W.max_points = z.max_points;
    W.edit_runs = (uint32_t*)(base + b_slots + b_paths + b_logs);
    W.node_masks = z.mask_width ? (uint32_t*)(base + b_slots + b_paths + b_logs + b_runs) : nullptr; W.mask_width = z.mask_width;
    return VGK_OK;
}
// one launch: every problem starts with the small tables in LDS; a wavefront runs what outgrows them again at once with its large slab
int launch_wave_form(vgk_ctx* ctx) {
    Backend* be = ctx->be.get();
    WwParams& A = ctx->wfa_wave_last[0];
    int rc;
    be->reset_wfa_ms();
    if ((rc = be->zero(A.n_declined, 16))) return rc;
    if ((rc = be->run_wfa_wave(A, ctx->wfa_wave_waves[0]))) return rc;
    ctx->wfa_wave_ms[0] = be->last_ms(6); ctx->wfa_wave_ms[1] = 0;
    unsigned long long taken_over = 0;
    if ((rc = be->download(&taken_over, A.n_todo_dev ? A.n_todo_dev : A.n_declined, sizeof taken_over))) return rc;
    ctx->wfa_wave_retried = taken_over;                                           // hybrid: what the thread kernel handed over; else: what outgrew the small tables
    if (A.stats) {
        constexpr size_t SW = WW_STAT_WORDS;
        std::vector<uint32_t> st(SW * (size_t)A.base.n);
        if (!be->download(st.data(), A.stats, sizeof(uint32_t) * st.size())) {
            std::vector<uint32_t> idx;
            for (uint32_t i = 0; i < A.base.n; ++i) if (st[SW * i + 2]) idx.push_back(i);
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return st[SW * a + 2] > st[SW * b + 2]; });
            unsigned long long chunks = 0, points = 0, steps = 0;
            for (uint32_t i : idx) { chunks += st[SW * i + 2]; points += st[SW * i]; steps += st[SW * i + 1]; }
            unsigned long long items = 0;
            for (uint32_t i : idx) items += st[SW * i + 3] >> 8;
            std::fprintf(stderr, "[wfa wave] %zu problems: %llu chunks, %llu points, %llu steps, %llu filtered items in all; heaviest (points, steps, chunks, trie nodes, items; us in extend, next, between, after; after the loop: backtrace, originals, room + output; before the loop):", idx.size(), chunks, points, steps, items);
            for (size_t k = 0; k < idx.size() && k < 8; ++k) std::fprintf(stderr, " (%u,%u,%u,%u,%u; %u,%u,%u,%u; %u,%u,%u,%u)", st[SW * idx[k]], st[SW * idx[k] + 1], st[SW * idx[k] + 2], st[SW * idx[k] + 3] & 255u, st[SW * idx[k] + 3] >> 8, st[SW * idx[k] + 4], st[SW * idx[k] + 5], st[SW * idx[k] + 6], st[SW * idx[k] + 7], st[SW * idx[k] + 8], st[SW * idx[k] + 9], st[SW * idx[k] + 10], st[SW * idx[k] + 11]);
            { unsigned long long us[4] = {0, 0, 0, 0}, nodes = 0;
              unsigned long long after[4] = {0, 0, 0, 0};
              for (uint32_t i : idx) { for (int k = 0; k < 4; ++k) { us[k] += st[SW * i + 4 + k]; after[k] += st[SW * i + 8 + k]; } nodes += st[SW * i + 3] & 255u; }
              std::fprintf(stderr, " | after the loop: %llu us backtrace, %llu the path's originals, %llu room + output; before the loop %llu us", after[0], after[1], after[2], after[3]);
              std::fprintf(stderr, " | all problems together: %llu us in extend, %llu in next, %llu at the start and between the two, %llu after the loop (backtrace, output); %llu trie nodes; the launch: %.2f ms x %u wavefronts = %.0f us of wavefront time",
                           us[0], us[1], us[2], us[3], nodes, be->last_ms(6), ctx->wfa_wave_waves[0], 1e3 * be->last_ms(6) * ctx->wfa_wave_waves[0]); }
            for (size_t q : {idx.size() / 100, idx.size() / 10, idx.size() / 2}) if (q < idx.size()) std::fprintf(stderr, " | rank %zu: (%u,%u,%u,%u,%u)", q, st[SW * idx[q]], st[SW * idx[q] + 1], st[SW * idx[q] + 2], st[SW * idx[q] + 3] & 255u, st[SW * idx[q] + 3] >> 8);
            std::fprintf(stderr, "\n");
        }
    }
    ctx->wfa_ms = be->last_ms(6);
    return VGK_OK;
}
// after_threads: the problems are the thread kernel's hand-over list (its length is on the device only)
// per_cu_default: wavefronts per CU of the launch (12: what a CU's LDS and registers hold of this kernel alone; fewer beside the thread kernel)
int prepare_wave_form(vgk_ctx* ctx, WfaHost& H, const WfaParams& P, bool after_threads, uint32_t per_cu_default) {
    Backend* be = ctx->be.get();
    const uint32_t cus = (uint32_t)std::max(1, be->compute_units());
    uint32_t per_cu = per_cu_default;                                            // 13 KB of LDS and 168 VGPRs per wavefront
    if (const char* e = std::getenv("VGAMD_WFA_WAVES_PER_CU")) per_cu = (uint32_t)std::max(1, std::atoi(e));
    // the small size keeps its tables in LDS (256 points cover all but a percent or two of giraffe's links; the median is a dozen); the
    // large size: what a link with a 60-base insertion under the default error model stores, several times over
    // (the item filter of the large size — wfa_wave_device.hpp, ww_chunk_of — keeps node sets for diagonals -256 .. 255; VGAMD_WFA_NO_FILTER
    // switches it off for comparisons)
    // The large size holds 262 144 points per link (5 MB of HBM per resident wavefront, 15.7 GB per context: the device has 288).  With 16 384 a
    // long-read batch of 209 000 links declined 400 — connects with a 25-60-base insertion, which the reference's WFA (no tables) answers — and the DP
    // route they then take is not bound to haplotypes: 3 of 8 000 chain scores came out higher than the reference's.  With 262 144: none declined for
    // its points, 16 000 / 16 000 identical; the launch is 28.7 instead of 18.6 ms (those links now run to their end).  65 536: three links still outgrow it.
    uint32_t large_points = 262144u;
    if (const char* e = std::getenv("VGAMD_WFA_LARGE_POINTS")) { large_points = 1024u; while (large_points < (uint32_t)std::min(1 << 22, std::max(1024, std::atoi(e)))) large_points <<= 1; }   // (experiments: what the large size can store, a power of two)
    WaveSlabs z{after_threads ? cus * per_cu : std::min<uint32_t>(P.n, cus * per_cu), 2u * large_points, large_points, 2048u, std::getenv("VGAMD_WFA_NO_FILTER") ? 0u : 512u};
    WwParams A{};
    A.base = P;
    A.index = H.walk_index; A.merge = H.walk_merge;
    // a caller's point budget below the tables' own sizes ends a problem as before (vgk_wfa_set_point_budgets); 0 = none
    A.base.max_points = ctx->wfa_point_budget ? ctx->wfa_point_budget : 0xffffffffu;
    A.base.max_points_tail = ctx->wfa_point_budget_tail ? ctx->wfa_point_budget_tail : 0xffffffffu;
    if (const char* e = std::getenv("VGAMD_WFA_SMALL_POINTS")) A.small_points = (uint32_t)std::max(16, std::atoi(e));     // (tests: more problems for the large size)
    int rc;
    if ((rc = carve(ctx, H, 62, z, A))) return rc;
    char* extra = (char*)ctx->ensure_scratch(63, 64);
    if (!extra) return VGK_ENOMEM;
    A.todo = P.order; A.n_todo = P.n; A.n_todo_dev = nullptr;
    if (after_threads) { A.todo = P.handed_over; A.n_todo = P.n; A.n_todo_dev = P.n_handed_over; A.base.handed_over = nullptr; }
    A.n_declined = (unsigned long long*)extra;
    A.stats = nullptr;
    if (std::getenv("VGAMD_WFA_STATS")) {                                        // per-problem statistics, printed by the call (a debugging aid)
        A.stats = (uint32_t*)ctx->ensure_scratch(64, sizeof(uint32_t) * WW_STAT_WORDS * ((size_t)P.n + 1));
        if (!A.stats || be->zero(A.stats, sizeof(uint32_t) * WW_STAT_WORDS * ((size_t)P.n + 1))) return VGK_ENOMEM;
    }
    ctx->wfa_wave_last[0] = A; ctx->wfa_wave_waves[0] = z.waves;
    return VGK_OK;
}
int run_wave_form(vgk_ctx* ctx, WfaHost& H, const WfaParams& P, bool after_threads) {
    int rc;
    if ((rc = prepare_wave_form(ctx, H, P, after_threads, 12))) return rc;
    if ((rc = launch_wave_form(ctx))) return rc;
    ctx->wfa_wave_last_valid = true; ctx->wfa_last_valid = false;
    return VGK_OK;
}
// The hybrid with both kernels AT ONCE (Backend::run_wfa_hybrid): the thread kernel at half its occupancy, the wavefront kernel in the
// other half of every CU, answering what the threads hand over while they are still at it.  Everything a launch consumes is reset here:
// the hand-out counters, the list (unwritten entries read 0xffffffff), the count of finished producers.
int launch_hybrid_at_once(vgk_ctx* ctx) {
    Backend* be = ctx->be.get();
    WfaParams& P = ctx->wfa_last; WwParams& A = ctx->wfa_wave_last[0];
    int rc;
    if ((rc = be->zero(P.counters, 64)) || (rc = be->zero(P.n_handed_over, 16)) || (rc = be->zero(P.producers_done, 16))) return rc;
    if ((rc = be->fill(P.handed_over, 0xff, sizeof(uint32_t) * ((size_t)P.n + 8)))) return rc;
    if ((rc = be->zero(A.n_declined, 16))) return rc;
    // (the two kernels share counters[0..1] — paths and edits handed out — and use counters[2] / counters[3] as their own hand-out counters)
    if ((rc = be->run_wfa_hybrid(P, ctx->wfa_last_threads, A, ctx->wfa_wave_waves[0]))) return rc;
    ctx->wfa_ms = be->last_ms(6); ctx->wfa_wave_ms[0] = ctx->wfa_ms; ctx->wfa_wave_ms[1] = 0;
    unsigned long long taken_over = 0;
    if ((rc = be->download(&taken_over, P.n_handed_over, sizeof taken_over))) return rc;
    ctx->wfa_wave_retried = taken_over;
    return VGK_OK;
}

}  // namespace

extern "C" {

int vgk_wfa_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_wfa_error_model* model, const vgk_wfa_problem* problems, uint32_t n,
                   vgk_wfa_result* results, uint32_t* paths, size_t path_cap, uint32_t* edits, size_t edit_cap, size_t written[2]) try {
    if (!ctx || !index || !vgk_tables_usable(index->ctx, ctx) || (!problems && n) || (!results && n)) return VGK_EINVAL;
    if (written) written[0] = written[1] = 0;
    if (!n) return VGK_OK;
    const vgk_wfa_error_model& em = model ? *model : kDefaultModel;
    for (const vgk_wfa_event* e : { &em.mismatches, &em.gaps, &em.gap_length })                  // the constructor's asserts (:1262-1270)
        if (e->per_base < 0 || e->min < 0 || e->max < e->min) return VGK_EINVAL;
    const int32_t match = ctx->sc.matrix[0], mism = -ctx->sc.matrix[1], go = ctx->sc.gap_open, ge = ctx->sc.gap_extend;
    if (match < 0 || mism <= 0 || go < ge || ge <= 0) return VGK_EUNSUPPORTED;                    // (:1256-1259)
    // the index with its unary runs merged is what the wavefront kernel walks: built here, once, by the first call that wants it (haplo.hpp)
    if (index->pending_merge && !std::getenv("VGAMD_WFA_NO_MERGE")) { if (int rcm = vgk_haplo_ensure_merged(const_cast<vgk_haplo*>(index))) return rcm; }
    std::lock_guard<std::mutex> lock(ctx->mu);
    // VGAMD_WFA_TIMES=1: where the call's host time goes (stderr, one line per call)
    const bool times = std::getenv("VGAMD_WFA_TIMES") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t_begin = now(); auto t_mark = t_begin; double t_phase[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](int k) { const auto t = now(); t_phase[k] += std::chrono::duration<double, std::milli>(t - t_mark).count(); t_mark = t; };
    ctx->wfa_out.valid = false;
    ctx->wfa_last_valid = false; ctx->wfa_wave_last_valid = false;            // (set again only by a call that got through: vgk_wfa_rerun must never relaunch over released buffers)
    // vgk_wfa_set_cost_hints: taken by THIS call whatever becomes of it (a call that fails below must not leave them to a later one)
    std::vector<uint32_t> hint_store; hint_store.swap(ctx->wfa_cost_hints);
    Backend* be = ctx->be.get();
    WfaParams P{};
    P.index = index->dev; P.n = n;
    P.match = match; P.bonus = ctx->sc.full_length_bonus;
    P.mismatch = 2 * (match + mism); P.gap_open = 2 * (go - ge); P.gap_extend = 2 * ge + match;   // (:1616-1618)

    if (!ctx->wfa_host) ctx->wfa_host = std::make_shared<WfaHost>();
    WfaHost& H = *static_cast<WfaHost*>(ctx->wfa_host.get());
    // the wavefront kernel walks the index with its unary runs merged when there is one (VGAMD_WFA_NO_MERGE=1: the original, for comparisons)
    { const bool runs = index->merged && index->merge.on && !std::getenv("VGAMD_WFA_NO_MERGE");
      H.walk_index = runs ? index->merged->dev : index->dev; H.walk_merge = runs ? index->merge : GMerge{}; }
    WProb* probs = H.probs.get(be, n);
    if (!probs) return VGK_ENOMEM;
    uint64_t n_seq = 0;
    // validation and descriptors in slices on the host threads; the sequence offsets from the slices' prefix sums afterwards
    const uint32_t slices = std::min<uint32_t>(64u, (n + 8191u) / 8192u);
    auto lo_of = [&](uint32_t c) { return (uint32_t)((uint64_t)n * c / slices); };
    std::vector<uint64_t> slice_seq((size_t)slices + 1, 0); std::vector<int> slice_bad(slices, 0);
    std::vector<uintptr_t> slice_lo(slices, ~(uintptr_t)0), slice_hi(slices, 0);    // the stretch of memory the slice's sequences lie in
    parallel_tasks(slices, [&](uint32_t c) {
    uint64_t n_seq = 0;                                                              // (of this slice)
    for (uint32_t i = lo_of(c); i < lo_of(c + 1); ++i) {
        const vgk_wfa_problem& p = problems[i];
        if (p.seq_len && !p.seq) { slice_bad[c] = 1; return; }
        WProb& w = probs[i];
        w.seq_off = (uint32_t)n_seq + 8; w.seq_len = p.seq_len; w.mode = p.mode;
        w.from_node = p.from_node; w.from_off = p.from_offset; w.to_node = p.to_node; w.to_off = p.to_offset;
        w.status = VGK_OK; w.score_bound = 0; w.distance_band = 0;
        if (p.mode > (uint32_t)VGK_WFA_PREFIX) w.status = VGK_EINVAL;
        else if (p.seq_len > 60000u) w.status = VGK_ETOOLONG;
        else if (p.mode == (uint32_t)VGK_WFA_PREFIX) {                                            // (:2253-2255)
            if (p.to_node >= index->n_oriented || p.to_offset >= index->len[p.to_node]) w.status = VGK_EINVAL;
            else { w.from_node = p.to_node ^ 1u; w.from_off = (index->len[p.to_node] - 1) - p.to_offset; }
            w.to_node = VGK_WFA_NO_NODE; w.to_off = 0;
        } else {
            if (p.from_node < index->n_oriented && p.from_offset >= index->len[p.from_node]) w.status = VGK_EINVAL;
            if (p.mode == (uint32_t)VGK_WFA_SUFFIX) { w.to_node = VGK_WFA_NO_NODE; w.to_off = 0; }
            else if (p.to_node < index->n_oriented && p.to_offset >= index->len[p.to_node]) w.status = VGK_EINVAL;
        }
        if (w.status == VGK_OK) {
            const int64_t bound = (int64_t)evaluate(em.mismatches, p.seq_len) * P.mismatch + (int64_t)evaluate(em.gaps, p.seq_len) * P.gap_open
                                + (int64_t)evaluate(em.gap_length, p.seq_len) * P.gap_extend;
            if (bound + P.gap_open + P.gap_extend + P.mismatch >= W_SCORES) w.status = VGK_ETOOBIG;
            else { w.score_bound = (int32_t)bound; w.distance_band = evaluate(em.distance, p.seq_len); }
        }
        if (w.status != VGK_OK) w.seq_len = 0;
        n_seq += w.seq_len;
        if (w.seq_len) { slice_lo[c] = std::min(slice_lo[c], (uintptr_t)p.seq); slice_hi[c] = std::max(slice_hi[c], (uintptr_t)p.seq + w.seq_len); }
    }
    slice_seq[c + 1] = n_seq;
    });
    for (uint32_t c = 0; c < slices; ++c) { if (slice_bad[c]) return VGK_EINVAL; slice_seq[c + 1] += slice_seq[c]; }
    n_seq = slice_seq[slices];
    if (n_seq > 0xfffffff0ull) return VGK_ETOOBIG;
    parallel_tasks(slices, [&](uint32_t c) { const uint32_t add = (uint32_t)slice_seq[c]; if (add) for (uint32_t i = lo_of(c); i < lo_of(c + 1); ++i) probs[i].seq_off += add; });
    // The sequences as the kernels read them: ReadMasker's bytes (:160-170), PREFIX problems reverse-complemented, behind each other with 8 bytes
    // of padding at either end.  When the caller's sequences lie in ONE stretch of memory (a read's pieces cut out of the read, a batch's reads
    // in one arena: the rule, not the exception) that stretch goes up as it is — straight from the caller's pages when they are page-locked
    // (vgk_host_register) — and a kernel masks, flips and lays out (wfa_mask_one): 108 MB per 8 000 long reads that no host thread touches.
    // Otherwise (pointers all over the heap) the host threads gather and mask as before.
    uintptr_t span_lo = ~(uintptr_t)0, span_hi = 0;
    for (uint32_t c = 0; c < slices; ++c) { span_lo = std::min(span_lo, slice_lo[c]); span_hi = std::max(span_hi, slice_hi[c]); }
    bool one_stretch = n_seq && span_hi > span_lo && (uint64_t)(span_hi - span_lo) <= std::max<uint64_t>(2 * n_seq, n_seq + (1u << 20)) && (uint64_t)(span_hi - span_lo) < 0xfffffff0ull
                       && !std::getenv("VGAMD_WFA_HOST_MASK");
    // ... and every byte of the stretch must be the caller's to read: the sequences in address order as they are in problem order, with less than a page
    // between one's end and the next one's start — every page such a gap touches then also holds a byte of a sequence, so it is mapped (a read's links
    // with its anchors' 29 bases between them; a batch's arena).  Anything else takes the host path.
    if (one_stretch) {
        uintptr_t prev_end = 0;
        for (uint32_t i = 0; i < n && one_stretch; ++i) {
            if (!probs[i].seq_len) continue;
            const uintptr_t a = (uintptr_t)problems[i].seq;
            if (prev_end && (a < prev_end || a - prev_end >= 4096)) one_stretch = false;
            prev_end = a + probs[i].seq_len;
        }
    }
    char* seqs = nullptr;
    if (!one_stretch) {
        seqs = H.seqs.get(be, n_seq + 16);
        if (!seqs) return VGK_ENOMEM;
        std::memset(seqs, 0, 8); std::memset(seqs + 8 + n_seq, 0, 8);
        parallel_for(n, [&](uint32_t i, unsigned) {
            const vgk_wfa_problem& p = problems[i]; const WProb& w = probs[i];
            char* r = seqs + w.seq_off;
            if (w.mode == (uint32_t)VGK_WFA_PREFIX) for (uint32_t k = 0; k < w.seq_len; ++k) r[k] = complement(p.seq[w.seq_len - 1 - k]);
            else for (uint32_t k = 0; k < w.seq_len; ++k) { const char c = p.seq[k]; r[k] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'X'; }
        });
    }

    lap(0);                                                                      // descriptors + masked sequences (host threads)
    int next_slot = 33;
    auto dev = [&](const void* src, size_t bytes) -> void* {
        void* d = ctx->ensure_scratch(next_slot++, std::max<size_t>(bytes, 16)); if (!d) return nullptr;
        if (src && bytes && be->upload(d, src, bytes)) return nullptr;
        return d;
    };
    P.probs = (const WProb*)dev(probs, sizeof(WProb) * n);
    if (!one_stretch) P.seqs = (const char*)dev(seqs, n_seq + 16);
    else {
        uint32_t* src_off = H.src_off.get(be, n);
        if (!src_off) return VGK_ENOMEM;
        parallel_for(n, [&](uint32_t i, unsigned) { src_off[i] = probs[i].seq_len ? (uint32_t)((uintptr_t)problems[i].seq - span_lo) : 0u; });
        char* d_seqs = (char*)ctx->ensure_scratch(next_slot++, n_seq + 16);
        const char* d_raw = (const char*)ctx->ensure_scratch(66, (uint64_t)(span_hi - span_lo) + 16);
        const uint32_t* d_src = (const uint32_t*)ctx->ensure_scratch(67, sizeof(uint32_t) * (uint64_t)n);
        if (!d_seqs || !d_raw || !d_src || !P.probs) return VGK_ENOMEM;
        if (be->upload((void*)d_raw, (const void*)span_lo, (size_t)(span_hi - span_lo)) || be->upload((void*)d_src, src_off, sizeof(uint32_t) * (size_t)n)) return VGK_ENODEV;
        if (be->zero(d_seqs, 8) || be->zero(d_seqs + 8 + n_seq, 8)) return VGK_ENODEV;
        if (int rcm = be->run_wfa_mask(P.probs, d_src, d_raw, d_seqs, n)) return rcm;
        if (be->sync()) return VGK_ENODEV;                                       // (the caller's pages were the copy's source: it is over before the call returns them)
        P.seqs = d_seqs;
    }
    // Hand-out order.  Threads take problems one at a time from a counter, so the order decides two things: problems that sit next to
    // each other in the graph run at the same time and find each other's records and bases in the L2 (mode 1), and long problems —
    // the expensive ones: the cost grows with the errors a sequence can hold — start first instead of leaving a tail of a few late
    // stragglers (mode 2; 3 = length classes of 32 bases, longest first, by node inside a class).  Results do not depend on it.
    int mode = 3;
    if (const char* e = std::getenv("VGAMD_WFA_ORDER")) mode = std::atoi(e);
    // vgk_wfa_set_cost_hints: what the caller expects a problem to cost beyond its length (giraffe knows the graph distance between two
    // anchors: a connect whose sequence is 40 bases longer than that holds a 40-base insertion and will fill the wavefront tables);
    // used once, for this call's order only (taken off the context at the top of the call)
    const uint32_t* hints = hint_store.size() == n ? hint_store.data() : nullptr;
    const uint32_t n_graph_nodes = index->n_oriented / 2 + 1;
    if (mode == 3 && n_graph_nodes <= (1u << 21)) {
        // the default order as ONE stable radix sort on the device: key = length class (longest first, 11 bits) | node (21 bits); the
        // keys are filled on the host threads, the sorted indices never come back (the two host counting sorts cost 5-8 ms per 500 k)
        uint32_t* d_sort = (uint32_t*)ctx->ensure_scratch(next_slot++, sizeof(uint32_t) * 4 * (size_t)n);       // key, index, sorted key, order
        if (!d_sort) return VGK_ENOMEM;
        std::vector<uint32_t> keys(2 * (size_t)n);
        parallel_for(n, [&](uint32_t i, unsigned) {
            const WProb& w = probs[i];
            const uint32_t v = (w.mode == (uint32_t)VGK_WFA_PREFIX ? w.to_node : w.from_node) / 2, node = v < n_graph_nodes ? v : n_graph_nodes - 1;
            const uint32_t c = (w.seq_len + (hints ? hints[i] : 0u)) / 32;       // (a caller's cost hint counts as that many more bases)
            keys[i] = ((2047u - (c < 2047u ? c : 2047u)) << 21) | node; keys[(size_t)n + i] = i;
        });
        if (be->upload(d_sort, keys.data(), sizeof(uint32_t) * 2 * (size_t)n)) return VGK_ENODEV;
        if (int rc0 = be->sort_pairs_u32(d_sort, d_sort + 2 * (size_t)n, d_sort + n, d_sort + 3 * (size_t)n, n, 32)) return rc0;
        if (be->sync()) return VGK_ENODEV;                                       // (keys goes out of scope)
        P.order = d_sort + 3 * (size_t)n;
    } else {
    std::vector<uint32_t> order(n);
    {
        // two stable counting-sort passes (node, then length class): O(n), a few ms per million problems
        const uint32_t n_nodes = index->n_oriented / 2 + 1;
        auto node_of = [&](uint32_t i) { const WProb& w = probs[i]; const uint32_t v = (w.mode == (uint32_t)VGK_WFA_PREFIX ? w.to_node : w.from_node) / 2; return v < n_nodes ? v : n_nodes - 1; };
        auto class_of = [&](uint32_t i) -> uint32_t {          // longest first: class 0 = the longest sequences
            const uint32_t len = probs[i].seq_len + (hints ? hints[i] : 0u), c = mode == 2 ? len : mode == 3 ? len / 32 : 0;
            return 0xffffu - (c < 0xffffu ? c : 0xffffu);
        };
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        if (mode == 1 || mode == 3) {
            std::vector<uint32_t> start(n_nodes + 1, 0), tmp(n);
            for (uint32_t i = 0; i < n; ++i) ++start[node_of(i) + 1];
            for (uint32_t v = 0; v < n_nodes; ++v) start[v + 1] += start[v];
            for (uint32_t i = 0; i < n; ++i) tmp[start[node_of(i)]++] = i;
            order.swap(tmp);
        }
        if (mode == 2 || mode == 3) {
            std::vector<uint32_t> start(0x10001, 0), tmp(n);
            for (uint32_t k = 0; k < n; ++k) ++start[class_of(order[k]) + 1];
            for (uint32_t c = 0; c < 0x10000; ++c) start[c + 1] += start[c];
            for (uint32_t k = 0; k < n; ++k) tmp[start[class_of(order[k])]++] = order[k];
            order.swap(tmp);
        }
    }
    P.order = (const uint32_t*)dev(order.data(), sizeof(uint32_t) * n);
    if (P.order && be->sync()) return VGK_ENODEV;                                // (order goes out of scope)
    }
    lap(1);                                                                      // uploads + the hand-out order
    // dense outputs; the kernel checks them
    const uint64_t cap_p = std::max<uint64_t>(path_cap, (uint64_t)n * 8 + n_seq / 4 + 1024) + 1, cap_e = std::max<uint64_t>(edit_cap, (uint64_t)n * 4 + 1024) + 1;
    P.caps[0] = cap_p; P.caps[1] = cap_e;
    P.max_points = ctx->wfa_point_budget && ctx->wfa_point_budget < (uint32_t)W_POINTS ? ctx->wfa_point_budget : (uint32_t)W_POINTS;
    P.max_points_tail = ctx->wfa_point_budget_tail && ctx->wfa_point_budget_tail < (uint32_t)W_POINTS ? ctx->wfa_point_budget_tail : (uint32_t)W_POINTS;
    // Two kernels answer the same problems with the same results: one WAVEFRONT per problem, the lanes being the diagonals
    // (wfa_wave_device.hpp; the default), and one THREAD per problem (wfa_device.hpp; VGAMD_WFA_KERNEL=thread).
    // and the default is both (hybrid): the thread kernel, which works on 64 problems per instruction, for the easy majority — it gives a
    // problem up at 16 stored points (the measured optimum: 9.4 ms per 500 000 bench problems; 8 / 32 / 128 points: 10.8 / 10.1 / 11.5 ms) — and the wavefront kernel for what it hands over.
    bool wave_form = ctx->wfa_form == VGK_WFA_FORM_WAVE, hybrid = ctx->wfa_form == VGK_WFA_FORM_HYBRID;       // vgk_wfa_set_form; the environment overrides (tests)
    if (const char* e = std::getenv("VGAMD_WFA_KERNEL")) { wave_form = std::strcmp(e, "wave") == 0; hybrid = std::strcmp(e, "hybrid") == 0; }
    uint64_t per_cu = 1024;         // 16 wavefronts per CU: the kernel is built for at most 128 VGPRs (__launch_bounds__(64, 4))
    if (const char* e = std::getenv("VGAMD_WFA_THREADS_PER_CU")) per_cu = (uint64_t)std::max(64, std::atoi(e));
    const uint32_t threads = wave_form ? 1u : (uint32_t)std::min<uint64_t>(n, (uint64_t)std::max(1, be->compute_units()) * per_cu);
    P.hand_over_points = 0; P.handed_over = nullptr; P.n_handed_over = nullptr;
    if (wave_form) P.scratch = reinterpret_cast<WScratch*>(ctx->ensure_scratch(32, 64));
    else
    { const uint64_t want = sizeof(WScratch) * (uint64_t)threads;
      P.scratch = (WScratch*)ctx->ensure_scratch(32, want);
      // the kernel leaves every slab's table all-zero; a fresh (or regrown) allocation is zeroed once
      if (P.scratch && (H.zeroed_ptr != (void*)P.scratch || H.zeroed_bytes < want)) {
          const uint64_t have = ctx->scratch[32].bytes;
          if (be->zero(P.scratch, have)) return VGK_ENODEV;
          H.zeroed_ptr = (void*)P.scratch; H.zeroed_bytes = have;
      } }
    P.results = (vgk_wfa_result*)dev(nullptr, sizeof(vgk_wfa_result) * n);
    P.paths = (uint32_t*)dev(nullptr, sizeof(uint32_t) * cap_p);
    P.edits = (uint32_t*)dev(nullptr, sizeof(uint32_t) * cap_e);
    P.counters = (unsigned long long*)dev(nullptr, 64);
    if (!P.probs || !P.seqs || !P.order || !P.scratch || !P.results || !P.paths || !P.edits || !P.counters) return VGK_ENOMEM;
    int rc;
    if ((rc = be->zero(P.counters, 64))) return rc;
    if (wave_form) {
        if ((rc = run_wave_form(ctx, H, P, false))) return rc;
    } else if (hybrid) {
        uint32_t hand_over = 16;
        if (const char* e = std::getenv("VGAMD_WFA_HAND_OVER_POINTS")) hand_over = (uint32_t)std::max(8, std::atoi(e));
        char* extra = (char*)ctx->ensure_scratch(61, sizeof(uint32_t) * ((size_t)n + 8) + 16);
        if (!extra) return VGK_ENOMEM;
        P.hand_over_points = hand_over; P.n_handed_over = (unsigned long long*)extra; P.handed_over = (uint32_t*)(extra + 16);
        if (be->wfa_concurrent() && std::getenv("VGAMD_WFA_AT_ONCE")) {
            // Both kernels at once: half the CU each (8 wavefronts of threads, 4 of the wavefront kernel).  Built, exact (every -m gpu WFA test
            // and the bench's 500 000 / 500 000), and NOT the default: the thread kernel fills the register file at 16 wavefronts per CU
            // (128 VGPRs each), so the wavefront kernel (168) only fits beside it at half that occupancy — and the thread kernel, bound by
            // memory latency, takes as much longer as it loses wavefronts: 9.46 ms per 500 000 problems either way (profiles/r04).  It pays
            // only with a wavefront kernel of 128 VGPRs beside 12 wavefronts of threads per CU.
            uint64_t t_per_cu = 512;
            if (const char* e = std::getenv("VGAMD_WFA_THREADS_PER_CU")) t_per_cu = (uint64_t)std::max(64, std::atoi(e));
            const uint32_t t_threads = (uint32_t)std::min<uint64_t>(n, (uint64_t)std::max(1, be->compute_units()) * t_per_cu);
            P.producers_done = (uint32_t*)ctx->ensure_scratch(65, 64);
            if (!P.producers_done) return VGK_ENOMEM;
            ctx->wfa_last = P; ctx->wfa_last_threads = t_threads;
            if ((rc = prepare_wave_form(ctx, H, P, true, 4))) return rc;
            WwParams& A = ctx->wfa_wave_last[0];
            A.base.counters = P.counters; A.producers_done = P.producers_done; A.n_producers = (t_threads + 63u) / 64u;
            A.hand_out = P.counters + 3;
            if ((rc = launch_hybrid_at_once(ctx))) return rc;
            ctx->wfa_last_valid = true; ctx->wfa_wave_last_valid = true; ctx->wfa_at_once = true;
        } else {
        ctx->wfa_at_once = false;
        if ((rc = be->zero(P.n_handed_over, 16))) return rc;
        be->reset_wfa_ms();
        if ((rc = be->run_wfa(P, threads))) return rc;
        ctx->wfa_last = P; ctx->wfa_last_threads = threads;
        const double ms_threads = be->last_ms(6);
        if ((rc = be->zero(P.counters + 2, 8))) return rc;                       // the hand-out counter starts over; paths / edits go on behind the first launch's
        if ((rc = run_wave_form(ctx, H, P, true))) return rc;
        ctx->wfa_wave_ms[1] = ctx->wfa_wave_ms[0]; ctx->wfa_wave_ms[0] = ms_threads;
        ctx->wfa_ms = ms_threads + ctx->wfa_wave_ms[1];
        ctx->wfa_last_valid = true;                                                // (vgk_wfa_rerun: both launches again)
        }
    } else {
        if ((rc = be->run_wfa(P, threads))) return rc;
        ctx->wfa_last = P; ctx->wfa_last_threads = threads; ctx->wfa_last_valid = true; ctx->wfa_wave_last_valid = false;
        ctx->wfa_ms = be->last_ms(6);
    }
    lap(2);                                                                      // launches, and the host's wait for them
    ctx->wfa_out.valid = true; ctx->wfa_out.n = n; ctx->wfa_out.res = P.results; ctx->wfa_out.paths = P.paths; ctx->wfa_out.edits = P.edits;
    ctx->wfa_out.path_cap = cap_p; ctx->wfa_out.edit_cap = cap_e; ctx->wfa_out.index = index;
    unsigned long long counters[2] = {0, 0};
    vgk_wfa_result* dres = H.dres.get(be, n);
    if (!dres) return VGK_ENOMEM;
    if ((rc = be->download(counters, P.counters, sizeof counters))) return rc;
    if ((rc = be->download(dres, P.results, sizeof(vgk_wfa_result) * n))) return rc;
    // a caller that wants scores only (no arrays, no room) gets just the results: the paths and edit runs stay on the device
    const bool scores_only = !paths && !edits && !path_cap && !edit_cap;
    const uint64_t np = scores_only ? 0 : std::min<uint64_t>(counters[0], cap_p), ne = scores_only ? 0 : std::min<uint64_t>(counters[1], cap_e);
    uint32_t* dpaths = H.dpaths.get(be, np + 1); uint32_t* dedits = H.dedits.get(be, ne + 1);
    if (!dpaths || !dedits) return VGK_ENOMEM;
    if (np && (rc = be->download(dpaths, P.paths, sizeof(uint32_t) * np))) return rc;
    if (ne && (rc = be->download(dedits, P.edits, sizeof(uint32_t) * ne))) return rc;
    lap(3);                                                                      // results (and paths / edit runs) down
    if (scores_only) {
        parallel_for(n, [&](uint32_t i, unsigned) { vgk_wfa_result r = dres[i]; if (r.status != VGK_OK) r.ok = 0; r.path_begin = r.path_len = r.edit_begin = r.n_edits = 0; results[i] = r; });
        lap(4);
        if (times) std::fprintf(stderr, "[wfa times] %u problems, %llu bases: descriptors + masking %.2f ms | uploads + order %.2f | launch + wait %.2f (kernels %.2f) | download %.2f | results %.2f\n",
                                n, (unsigned long long)n_seq, t_phase[0], t_phase[1], t_phase[2], ctx->wfa_ms, t_phase[3], t_phase[4]);
        return VGK_OK;
    }
    // the device packs alignments in completion order; hand them back in problem order
    std::vector<uint64_t> op(n + 1, 0), oe(n + 1, 0);
    int rc_all = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        vgk_wfa_result& r = dres[i];
        if (r.status == VGK_OK && r.ok && (op[i] + r.path_len > path_cap || oe[i] + r.n_edits > edit_cap || (r.path_len && !paths) || (r.n_edits && !edits))) r.status = VGK_EOPS;
        if (r.status == VGK_EOPS) rc_all = VGK_EOPS;
        if (r.status != VGK_OK) { r.ok = 0; r.path_len = r.n_edits = 0; }
        op[i + 1] = op[i] + r.path_len; oe[i + 1] = oe[i] + r.n_edits;
    }
    parallel_for(n, [&](uint32_t i, unsigned) {
        vgk_wfa_result r = dres[i];
        if (r.path_len) std::memcpy(paths + op[i], dpaths + r.path_begin, sizeof(uint32_t) * r.path_len);
        if (r.n_edits) std::memcpy(edits + oe[i], dedits + r.edit_begin, sizeof(uint32_t) * r.n_edits);
        r.path_begin = (uint32_t)op[i]; r.edit_begin = (uint32_t)oe[i];
        results[i] = r;
    });
    if (written) { written[0] = op[n]; written[1] = oe[n]; }
    return rc_all;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_wfa_rerun(vgk_ctx* ctx) try {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->wfa_wave_last_valid && ctx->wfa_last_valid && ctx->wfa_at_once) return launch_hybrid_at_once(ctx);      // hybrid, both kernels at once
    if (ctx->wfa_wave_last_valid && ctx->wfa_last_valid) {                         // hybrid: the thread kernel, then the wavefront kernel on what it hands over
        Backend* be = ctx->be.get();
        int rc;
        if ((rc = be->zero(ctx->wfa_last.counters, 64)) || (rc = be->zero(ctx->wfa_last.n_handed_over, 16))) return rc;
        be->reset_wfa_ms();
        if ((rc = be->run_wfa(ctx->wfa_last, ctx->wfa_last_threads))) return rc;
        const double ms_threads = be->last_ms(6);
        if ((rc = be->zero(ctx->wfa_last.counters + 2, 8))) return rc;
        if ((rc = launch_wave_form(ctx))) return rc;
        ctx->wfa_wave_ms[1] = ctx->wfa_wave_ms[0]; ctx->wfa_wave_ms[0] = ms_threads; ctx->wfa_ms = ms_threads + ctx->wfa_wave_ms[1];
        return VGK_OK;
    }
    if (ctx->wfa_wave_last_valid) {
        if (int rc = ctx->be->zero(ctx->wfa_wave_last[0].base.counters, 64)) return rc;
        return launch_wave_form(ctx);
    }
    if (!ctx->wfa_last_valid) return VGK_EINVAL;
    int rc;
    if ((rc = ctx->be->zero(ctx->wfa_last.counters, 64))) return rc;
    if ((rc = ctx->be->run_wfa(ctx->wfa_last, ctx->wfa_last_threads))) return rc;
    ctx->wfa_ms = ctx->be->last_ms(6);
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_wfa_last_ms(vgk_ctx* ctx) { return ctx ? ctx->wfa_ms : 0.0; }
// 0 = ms of the first launch (hybrid: the thread kernel; wave form: the only one), 1 = ms of the wavefront kernel behind the thread kernel (hybrid),
// 2 = problems the thread kernel handed over (hybrid) / that outgrew the small tables (wave form)
double vgk_wfa_last_wave(vgk_ctx* ctx, int which) { return !ctx ? 0.0 : which == 0 ? ctx->wfa_wave_ms[0] : which == 1 ? ctx->wfa_wave_ms[1] : (double)ctx->wfa_wave_retried; }
int vgk_wfa_set_cost_hints(vgk_ctx* ctx, const uint32_t* extra_bases, uint32_t n) try {
    if (!ctx || (!extra_bases && n)) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->wfa_cost_hints.assign(extra_bases, extra_bases + n);
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
int vgk_wfa_set_form(vgk_ctx* ctx, int form) try {
    if (!ctx || form < VGK_WFA_FORM_HYBRID || form > VGK_WFA_FORM_WAVE) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu); ctx->wfa_form = form;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)
int vgk_wfa_get_form(vgk_ctx* ctx) {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu); return ctx->wfa_form;
}
int vgk_wfa_set_point_budget(vgk_ctx* ctx, uint32_t points) { return vgk_wfa_set_point_budgets(ctx, points, points); }
int vgk_wfa_set_point_budgets(vgk_ctx* ctx, uint32_t connect_points, uint32_t tail_points) try {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lock(ctx->mu); ctx->wfa_point_budget = connect_points; ctx->wfa_point_budget_tail = tail_points;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

}  // extern "C"
