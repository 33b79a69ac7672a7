// ctx.hpp — the engine context shared by the C-ABI translation units (vgk_api.cpp, banded_api.cpp).
#pragma once
#include <memory>
#include <mutex>
#include <vector>
#include "backend.hpp"

// page-locked host staging kept on the context between calls: uninitialised, no page faults, full-rate DMA in both directions
template <class T> struct PinnedBuf {
    vgk::Backend* be = nullptr; T* p = nullptr; size_t cap = 0;
    T* get(vgk::Backend* b, size_t n) {
        if (n > cap || !p) { if (p) be->host_release(p); be = b; cap = n + n / 4 + 64; p = (T*)b->host_alloc(cap * sizeof(T)); if (!p) cap = 0; }
        return p;
    }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) be->host_release(p); }
};

// Runtime feedback for the speculative fill (GsswParams::spec_fill, DESIGN.md: "speculation with feedback").  A batch is packed so that it CAN
// speculate (one geometry, mostly local alignments with tracebacks); whether a run of it DOES is decided when it is launched, from what the
// context's earlier speculative runs cost: each leaves the number of wavefronts it had to fill a second time (refill_count, read back with the
// results), and a run whose share of such wavefronts passes `miss_max` paid the first fill for nothing (break-even on the MI355X: first fill
// 12.7 ms + m x 19.1 ms against 19.1 ms plain per million reads => m < 0.33, less the layout and the second walk).  Above it the context stops
// speculating and probes again after `interval` runs, doubling the interval while the probes keep failing (16, 32, ... 1024): a stream of
// reads full of indels then runs within a few per cent of the plain fill, and a stream that turns clean again is noticed.
// Guarded by vgk_ctx::mu.
struct SpecPolicy {
    int mode = 0;                    // vgk_set_speculation: 0 = by feedback, 1 = whenever the batch allows it, 2 = never
    double miss_max = 0.25;          // VGAMD_SPEC_MISS_MAX
    uint32_t probe_every = 16;       // VGAMD_SPEC_PROBE_EVERY: the first interval between probes while speculation is off
    bool on = true; uint32_t wait = 0, interval = 16;
    uint64_t observed = 0, turned_off = 0, turned_on = 0; double last_miss = 0.0;
    // *probe: this run is a probe (one speculative run while speculation is off); its observation alone may lengthen the interval
    bool decide(bool* probe = nullptr) {
        if (probe) *probe = false;
        if (mode == 1) return true;
        if (mode == 2) return false;
        if (on) return true;
        if (wait) { --wait; return false; }
        wait = interval;                                                   // a probe: one speculative run; its count decides
        if (probe) *probe = true;
        return true;
    }
    // was_probe: the observed run was launched as a probe.  A run launched while speculation was still ON and observed after it went off (two
    // batches in flight) is no probe: its high count neither turns anything off again nor doubles the interval.
    void observe(double miss, bool was_probe = false) {
        ++observed; last_miss = miss;
        if (miss > miss_max) {
            if (on) { on = false; ++turned_off; interval = probe_every; wait = interval; }
            else if (was_probe) { interval = interval >= 512 ? 1024 : interval * 2; wait = interval; }      // a probe that failed
        } else if (!on && (was_probe || mode == 0)) { on = true; ++turned_on; interval = probe_every; wait = 0; }
    }
};

struct vgk_ctx;
// may `user` run over read-only tables (a haplotype index, a minimizer index) that `owner` put into HBM?  Its own, or those of another context
// on the SAME device of the same library: the tables are device memory of that device, complete when their create call returned, and never
// written again (vgk_minimizer_set_policy aside, which both then see).  The owner must outlive every user.  (include/vgk.h: "sharing an index")
static inline bool vgk_tables_usable(const vgk_ctx* owner, const vgk_ctx* user);

struct vgk_ctx {
    vgk_scoring sc;
    SpecPolicy spec;               // speculative fill: on / off by the miss counts of this context's earlier runs
    std::unique_ptr<vgk::Backend> be;
    std::mutex mu;                 // guards the pools and the launch order of a context
    std::mutex stage_mu;           // taken BEFORE mu, for their whole duration, by the calls that make or consume the state one stage leaves for the next in HBM
                                   // (seeded clusters, extension sets, their scratch slots): vgk_minimizer_seeds, vgk_gapless_extend*, vgk_gapless_rerun, vgk_tail_stage* —
                                   // the tail stage lets go of mu around the window packer and the fills, never of this one
    uint32_t batch_seq = 0;        // batches packed so far: their launch lanes alternate
    uint32_t bias = 1; int32_t max_score = 0; int32_t max_bonus = 0;
    uint32_t prof4[6];
    uint32_t scale = 1;            // 8 when the scaled profile bytes still fit (GsswParams::scale)
    bool has_qa = false;           // quality-adjusted (QualAdjAligner) context
    std::vector<int8_t> qmat, qbon;
    // last vgk_banded_align call: kernel times (ms), band cells, algorithmic bytes
    double banded_ms[2] = {0, 0}; uint64_t banded_cells = 0, banded_bytes = 0;
    double gapless_ms = 0;         // kernel time of the last vgk_gapless_extend call
    double wide_ms[2] = {0, 0}; uint64_t wide_cells = 0, wide_tb_cells = 0, wide_launches = 0;     // the wide route's share of the last vgk_gssw_align call: fill | traceback kernels (ms), DP cells, cells with codes, launches
    uint64_t gapless_redone = 0;   // seeds of that call whose search branched on the merged-run index and ran again on the original one
    uint64_t gapless_retried = 0;  // reads of that call whose search outgrew the fast kernel's LDS store and ran in the slab kernel
    double wfa_ms = 0;             // and of the last vgk_wfa_extend call
    int wfa_form = 0;              // vgk_wfa_set_form
    uint32_t wfa_point_budget = 0, wfa_point_budget_tail = 0; // vgk_wfa_set_point_budget(s): connects | prefixes and suffixes (0 = the table's size)
    int8_t banded_mat_rows[72] = {0};   // the 5 x 5 table + its rows as 64-bit words, as the banded fill kernel reads them (banded_device.hpp BMAT_ROWS_AT)
    // what the last vgk_minimizer_seeds call left in HBM: the (masked) reads behind 8 bytes of padding, their offsets, the seeds per read —
    // vgk_gapless_extend_seeded takes its clusters from there
    struct Seeded { bool valid = false; uint32_t n = 0; const char* reads = nullptr; uint64_t bytes = 0; const uint64_t* read_off = nullptr;
                    const uint32_t* seed_off = nullptr; const vgk_seed* seeds = nullptr; uint64_t n_seeds = 0; const void* graph = nullptr; } seeded;
    // what the last vgk_gapless_extend(_seeded) call left in HBM: its inputs (descriptors, masked reads) and its sets in problem order —
    // vgk_tail_stage derives the tails from there
    struct Sets { bool valid = false; uint32_t n = 0; uint64_t n_ext = 0; const void* probs = nullptr; const char* reads = nullptr;
                  const void* res = nullptr; const void* ext = nullptr; const uint32_t* nodes = nullptr; const void* index = nullptr;
                  const uint32_t* read_of = nullptr; } sets;      // read_of[e]: the read extension e belongs to
    // VGK_GAPLESS_DEFER: the sets of that call are still on their way down (fetch stream -> page-locked staging) and have yet to be
    // copied into the caller's arrays; finished by vgk_tail_stage*, vgk_gapless_fetch_deferred, or the next extension call
    struct DeferredSpan { char* dst; const char* src; size_t bytes; };
    struct Deferred { bool pending = false, queued = false; DeferredSpan spans[4] = {}; const void* from[4] = {}; void* ev = nullptr; } deferred;
    int start_deferred();          // queue the copies on the fetch stream (behind everything the main stream holds now)
    int finish_deferred();
    double minimizer_ms = 0;       // device time of the last vgk_minimizer_seeds call
    double tail_stage_ms[4] = {0, 0, 0, 0};   // last vgk_tail_stage: tails derived | forest | windows packed | kernels + totals
    double tail_ms = 0;            // device time of the last vgk_tail_forest call
    // the last batch of either call stays resident in the cached device buffers: what a re-run needs to launch it again
    vgk::BandedParams banded_last{}; std::vector<vgk::BandedLaunch> banded_last_launches; bool banded_last_valid = false;
    vgk::GaplessParams gapless_last{}; uint32_t gapless_last_threads = 0; bool gapless_last_valid = false;
    vgk::WfaParams wfa_last{}; uint32_t wfa_last_threads = 0; bool wfa_last_valid = false;
    vgk::WwParams wfa_wave_last[2] = {}; uint32_t wfa_wave_waves[2] = {0, 0}; bool wfa_wave_last_valid = false;    // the wavefront form: small-table launch, large-table launch
    double wfa_wave_ms[2] = {0, 0}; uint64_t wfa_wave_retried = 0;
    bool wfa_at_once = false;            // the last hybrid call ran its two kernels at once (vgk_wfa_rerun does the same)
    // what the last vgk_wfa_extend call left in HBM — results in problem order, node paths and edit runs in completion order — for vgk_chain_stitch's LINK pieces
    struct WfaOut { bool valid = false; uint32_t n = 0; const vgk_wfa_result* res = nullptr; const uint32_t* paths = nullptr; const uint32_t* edits = nullptr;
                    uint64_t path_cap = 0, edit_cap = 0; const void* index = nullptr; } wfa_out;
    double chain_stitch_ms = 0;          // device time of the last vgk_chain_stitch call
    std::shared_ptr<void> chain_host;    // host staging arenas of chain_api.cpp
    std::vector<uint32_t> wfa_cost_hints;   // vgk_wfa_set_cost_hints: per problem of the NEXT vgk_wfa_extend, bases to add to its length when the hand-out order is made
    uint64_t multi_host_walks = 0;       // problems of the last vgk_gssw_align_multi whose alternates a host thread walked
    // device scratch kept between vgk_banded_align calls (grow-only; released with the context)
    struct DevBuf { void* p = nullptr; uint64_t bytes = 0; };
    DevBuf scratch[160];           // 140..149 chain_api.cpp; 150..157 minimizer_api.cpp (reads of any length); 66, 67 wfa_api.cpp (the sequences as the caller holds them, their offsets); 88..99 gssw_wide_api.cpp; 65 wfa_api.cpp (producers_done); 72..83 gssw_multi_api.cpp (the walk on the device); 0..14 + 31 banded_api.cpp (+ 124..138: its second sub-batch in flight), 15..30 + 59, 60 gapless_api.cpp, 32..39 + 61..63 wfa_api.cpp, 40..47 gssw_multi_api.cpp / xdrop_band_api.cpp (+ 48, 49, 87; its second sub-batch in flight: 100..123), 50..54 tail_api.cpp, 55..58 minimizer_api.cpp
    void* ensure_scratch(int slot, uint64_t bytes) {
        DevBuf& b = scratch[slot];
        if (b.p && b.bytes >= bytes) return b.p;
        if (b.p) { be->sync(); be->release(b.p); b.p = nullptr; b.bytes = 0; }
        const uint64_t want = bytes + bytes / 4 > 4096 ? bytes + bytes / 4 : 4096;
        b.p = be->alloc(want); b.bytes = want;
        if (!b.p) { b.p = be->alloc(bytes); b.bytes = b.p ? bytes : 0; }
        if (!b.p && !dev_pool.empty()) {                  // the pooled arenas of freed gssw batches go first
            be->sync(); while (!dev_pool.empty()) dev_pool_drop(dev_pool.size() - 1);
            b.p = be->alloc(bytes); b.bytes = b.p ? bytes : 0;
        }
        return b.p;
    }
    // Device arenas of freed gssw batches, kept for the next pack (callers hold `mu`): allocating the 35 GB of a million-read batch
    // takes the runtime 0.9-1.8 s, more than packing, aligning and fetching it.  Requests are rounded up by an eighth so that the
    // next, slightly larger batch still fits; a block serves requests down to half its size; at most 192 blocks / half of HBM stay (a batch holds 13 arenas, a fetch 4 more, a forest 9, a tail stage 13 and the packer 13 temporaries: with 64 the stage of §17 kept dropping and re-allocating blocks).
    struct Pooled { void* p; uint64_t bytes; };
    std::vector<Pooled> dev_pool; uint64_t dev_pool_bytes = 0;
    void dev_pool_drop(size_t k) { be->release(dev_pool[k].p); dev_pool_bytes -= dev_pool[k].bytes; dev_pool.erase(dev_pool.begin() + (long)k); }
    void* dev_take(uint64_t bytes, uint64_t& got) {
        size_t best = dev_pool.size();
        for (size_t k = 0; k < dev_pool.size(); ++k)
            if (dev_pool[k].bytes >= bytes && dev_pool[k].bytes / 2 <= bytes + 4096 && (best == dev_pool.size() || dev_pool[k].bytes < dev_pool[best].bytes)) best = k;
        if (best != dev_pool.size()) {
            void* p = dev_pool[best].p; got = dev_pool[best].bytes;
            dev_pool_bytes -= got; dev_pool.erase(dev_pool.begin() + (long)best);
            return p;
        }
        got = bytes + bytes / 8 + 256;
        void* p = be->alloc(got);
        if (!p) {                                         // make room: give the pooled blocks back, then ask for what is needed only
            be->sync();
            while (!dev_pool.empty()) dev_pool_drop(dev_pool.size() - 1);
            got = bytes ? bytes : 16; p = be->alloc(got);
        }
        return p;
    }
    void dev_give(void* p, uint64_t bytes) {
        dev_pool.push_back({p, bytes}); dev_pool_bytes += bytes;
        const uint64_t limit = be->memory_bytes() ? be->memory_bytes() / 2 : (1ull << 30);
        while (!dev_pool.empty() && (dev_pool.size() > 192 || dev_pool_bytes > limit)) dev_pool_drop(0);     // oldest first
    }
    // page-locked staging arenas of vgk_gssw_pack, handed out per pack (callers may pack concurrently) and kept for the next one
    struct Staging {
        vgk::Backend* be = nullptr; void* p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; uint64_t bytes[7] = {0, 0, 0, 0, 0, 0, 0};   // [5]: results and CIGAR ops on their way back, [6]: per-problem sizes while packing
        void* get(int k, uint64_t want) {
            if (bytes[k] >= want) return p[k];
            if (p[k]) be->host_release(p[k]);
            bytes[k] = want + want / 4 + 4096; p[k] = be->host_alloc(bytes[k]);
            if (!p[k]) bytes[k] = 0;
            return p[k];
        }
        ~Staging() { for (void* q : p) if (q) be->host_release(q); }
    };
    std::mutex staging_mu; std::vector<std::unique_ptr<Staging>> staging_free;
    std::unique_ptr<Staging> staging_acquire() {
        std::lock_guard<std::mutex> lock(staging_mu);
        if (staging_free.empty()) { auto s = std::make_unique<Staging>(); s->be = be.get(); return s; }
        auto s = std::move(staging_free.back()); staging_free.pop_back(); return s;
    }
    void staging_release(std::unique_ptr<Staging> s) { std::lock_guard<std::mutex> lock(staging_mu); if (staging_free.size() < 4) staging_free.push_back(std::move(s)); }
    // page-locked blocks that held the problem descriptors of freed batches (a batch keeps them until it is freed): no page
    // faults while the next batch is packed, no unmapping when it is freed, full-rate DMA
    std::vector<Pooled> host_pool;
    void* host_take(uint64_t bytes, uint64_t& got) {
        {
            std::lock_guard<std::mutex> lock(staging_mu);
            for (size_t k = 0; k < host_pool.size(); ++k)
                if (host_pool[k].bytes >= bytes && host_pool[k].bytes / 2 <= bytes + 4096) { void* p = host_pool[k].p; got = host_pool[k].bytes; host_pool.erase(host_pool.begin() + (long)k); return p; }
        }
        got = bytes + bytes / 8 + 256;
        return be->host_alloc(got);
    }
    void host_give(void* p, uint64_t bytes) {
        std::lock_guard<std::mutex> lock(staging_mu);
        host_pool.push_back({p, bytes});
        while (host_pool.size() > 4) { be->host_release(host_pool.front().p); host_pool.erase(host_pool.begin()); }
    }
    std::shared_ptr<void> banded_host;      // host staging arenas of banded_api.cpp
    std::shared_ptr<void> gapless_host;     // and of gapless_api.cpp
    std::shared_ptr<void> wfa_host;         // and of wfa_api.cpp
    std::shared_ptr<void> multi_host;       // and of gssw_multi_api.cpp
    std::shared_ptr<void> xband_host;       // and of xdrop_band_api.cpp
    double xband_ms = 0;                    // kernel time of the last vgk_xdrop_band_align call
    int xband_cells = 4;                    // ... and its cell form (vgk_xdrop_band_last_cells)
    uint64_t xband_class[3] = {0, 0, 0};    // ... and its problems by the lanes they ran on: 8 | 16 | 64 (vgk_xdrop_band_last_class)
    ~vgk_ctx() { if (be) { if (deferred.pending) be->sync_fetch(); if (deferred.ev) be->event_destroy(deferred.ev); for (DevBuf& b : scratch) if (b.p) be->release(b.p); for (Pooled& q : dev_pool) be->release(q.p); for (Pooled& q : host_pool) be->host_release(q.p); } }
};

static inline bool vgk_tables_usable(const vgk_ctx* owner, const vgk_ctx* user) {
    return owner && user && (owner == user || (owner->be && user->be && owner->be->device_index() == user->be->device_index()));
}
