// backend_emu.cpp — TEST-ONLY backend: runs the very same lane code
// (vg_amd/csrc/gssw_device.hpp) and the same packing layer (vgk_api.cpp) on the
// CPU by stepping all 64 lanes of every wavefront in lock-step, with the DPP
// wave_shr:1 exchange replaced by reading the previous step's values of lane-1.
// It lets the kernel logic be debugged in a container without a GPU.  It is
// never built into, nor loaded by, the product library.
#include <pthread.h>
#include <ucontext.h>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../vg_amd/csrc/backend.hpp"

namespace vgk {

// banded kernels: one CPU thread per lane, the cross-lane primitives of banded_device.hpp through a barrier
struct XlShared { pthread_barrier_t bar; int32_t buf[64]; unsigned long long wide[64]; uint8_t stage[64]; int32_t cache[2 * 2 * 64 * 8]; };        // stage: a small one, so that tests refill it
struct XlEmu {
    XlShared* sh; uint32_t lane; uint32_t lanes = 64;          // lanes: how many run together (64, or one DPP row of 16)
    uint32_t width() const { return lanes; }
    uint8_t* stage() { return sh->stage; }
    uint32_t stage_cap() const { return 64; }
    void stage_sync() { pthread_barrier_wait(&sh->bar); }
    int32_t* col_cache() { return sh->cache; }
    void lds_sync() { pthread_barrier_wait(&sh->bar); }
    int32_t exchange(int32_t v, int mode) {
        sh->buf[lane] = v;
        pthread_barrier_wait(&sh->bar);
        int32_t out = BNEG;
        if (mode == 0) { if (lane + 1 < lanes) out = sh->buf[lane + 1]; }
        else if (mode == 1) { if (lane > 0) out = sh->buf[lane - 1]; }
        else if (mode == 3) { for (uint32_t l = 0; l < lanes; ++l) out = bmax(out, sh->buf[l]); }
        else if (mode == 4) out = sh->buf[0];
        else if (mode == 5) out = sh->buf[lanes - 1];
        else for (uint32_t l = 0; l < lane; ++l) out = bmax(out, sh->buf[l]);
        pthread_barrier_wait(&sh->bar);
        return out;
    }
    int32_t up(int32_t v) { return exchange(v, 0); }
    int32_t down(int32_t v) { return exchange(v, 1); }
    int32_t up_sub(int32_t old, int32_t v, int32_t s) { const int32_t o = exchange(v, 0); return lane + 1 < lanes ? o - s : old; }
    int32_t down_sub(int32_t old, int32_t v, int32_t s) { const int32_t o = exchange(v, 1); return lane > 0 ? o - s : old; }
    int32_t in_lanes(int32_t s) { return s; }
    int32_t scalar(int32_t s) { return s; }
    int32_t add3(int32_t a, int32_t b, int32_t s) { return a + b + s; }
    int32_t scan_excl_keep(int32_t old, int32_t v) { const int32_t o = exchange(v, 2); return lane > 0 ? o : old; }
    int32_t scan_excl(int32_t v) { return exchange(v, 2); }
    void fence() { pthread_barrier_wait(&sh->bar); }
    bool any(int32_t flag) { return exchange(flag ? 1 : BNEG, 3) > 0; }
    unsigned long long ballot(bool flag) {
        sh->buf[lane] = flag ? 1 : 0;
        pthread_barrier_wait(&sh->bar);
        unsigned long long m = 0; for (uint32_t l = 0; l < lanes; ++l) if (sh->buf[l]) m |= 1ull << l;
        pthread_barrier_wait(&sh->bar);
        return m;
    }
    int32_t reduce_max(int32_t v) { return exchange(v, 3); }
    int32_t first_lane(int32_t v) { return exchange(v, 4); }
    int32_t last_lane(int32_t v) { return exchange(v, 5); }
    unsigned long long reduce_add(unsigned long long v) {
        sh->wide[lane] = v;
        pthread_barrier_wait(&sh->bar);
        unsigned long long t = 0; for (uint32_t l = 0; l < lanes; ++l) t += sh->wide[l];
        pthread_barrier_wait(&sh->bar);
        return t;
    }
};
template <int R, bool QA> static void banded_fill_emu(const BandedParams& P, uint32_t begin, uint32_t count) {
    XlShared sh; pthread_barrier_init(&sh.bar, nullptr, 64);
    std::vector<std::thread> ts;
    for (uint32_t lane = 0; lane < 64; ++lane) ts.emplace_back([&, lane]() {
        XlEmu xl{&sh, lane};
        for (uint32_t i = 0; i < count; ++i) {
            const BProb pb = P.probs[P.order[begin + i]];
            BSrc src; src.rd = P.reads + pb.read_off; src.q = QA ? P.quals + pb.read_off : nullptr; src.graph = P.graph + pb.graph_off; src.mat = P.mat;
            src.rows = reinterpret_cast<const uint64_t*>(P.mat + BMAT_ROWS_AT);
            constexpr bool FAST = !QA && R <= 4;                            // the kernel's choice for problems staged in LDS; both paths give the same cells
            if constexpr (R >= 64) {                                        // bands of more than 2048 diagonals: any number of blocks, state and first rows in the lane's stores
                std::vector<int32_t> store((size_t)(R / 8) * 3 * 8, BNEG), firsts((size_t)3 * (R / 8 + 1), BNEG);
                banded_fill_lane_blocks<0, QA>(P, pb, src, lane, xl, store.data(), 1u, (uint32_t)(R / 8), firsts.data());
            } else
            if constexpr (R >= 16) {                                        // wide bands: blocks of 8 rows per lane, the other blocks' rows in the lane's store
                std::vector<int32_t> store((size_t)(R / 8) * 3 * 8, BNEG);
                if (std::getenv("VGAMD_EMU_BANDED_TALL_LANES")) banded_fill_lane<R, QA, false>(P, pb, src, lane, xl);      // (the form it replaces: the two must agree)
                else banded_fill_lane_blocks<R / 8, QA>(P, pb, src, lane, xl, store.data(), 1u);
            } else
            if (FAST && (i & 1) == 0) banded_fill_lane<R, QA, FAST>(P, pb, src, lane, xl);
            else banded_fill_lane<R, QA, false>(P, pb, src, lane, xl);
            xl.fence();
        }
    });
    for (auto& t : ts) t.join();
    pthread_barrier_destroy(&sh.bar);
}

class EmuBackend final : public Backend {
public:
    // the wavefront form (64 threads as lanes, the banded emulation's cross-lane primitives); the one-thread form for go < ge
    template <int R> static void matrix_wave_emu(const GsswMatrixParams& P, uint32_t lo, uint32_t hi) {
        XlShared sh; pthread_barrier_init(&sh.bar, nullptr, 64);
        std::vector<std::thread> ts;
        for (uint32_t lane = 0; lane < 64; ++lane) ts.emplace_back([&, lane]() {
            XlEmu xl{&sh, lane};
            for (uint32_t i = 0; i < P.n; ++i) { if (P.probs[i].L <= lo || P.probs[i].L > hi) continue; gssw_matrix_wave_lane<R>(P, i, lane, xl); xl.fence(); }
        });
        for (auto& t : ts) t.join();
        pthread_barrier_destroy(&sh.bar);
    }
    // the wide route: the 256 lanes of a problem's skewed wavefront stepped one after the other, the row above from lane - 1's previous step
    template <int K> static void wide_fill(const WideParams& P, uint32_t i) {
        const WideProb d = P.probs[i];
        uint32_t* tb = (d.flags & VGK_GSSW_TRACEBACK) ? P.tb + d.tb_off : nullptr;
        std::vector<WLane<K>> lanes(WIDE_LANES);
        std::vector<int32_t> oh(WIDE_LANES), of(WIDE_LANES); std::vector<uint32_t> oi(WIDE_LANES);
        for (uint32_t strip = 0; strip < d.n_strips; ++strip) {
            for (uint32_t l = 0; l < WIDE_LANES; ++l) wide_lane_init<K>(lanes[l], P, d, strip, l);
            const uint32_t rows_left = d.L - strip * WIDE_LANES * (uint32_t)K;
            const uint32_t lanes_used = rows_left >= WIDE_LANES * (uint32_t)K ? WIDE_LANES : (rows_left + (uint32_t)K - 1u) / (uint32_t)K;
            const uint32_t n_steps = d.R + lanes_used - 1u;
            for (uint32_t t = 0; t < n_steps; ++t) {
                for (uint32_t l = 0; l < WIDE_LANES; ++l) { oh[l] = lanes[l].out_h; of[l] = lanes[l].out_f; oi[l] = lanes[l].info; }
                for (uint32_t l = 0; l < WIDE_LANES; ++l) {
                    uint32_t* rec = tb ? tb + (uint64_t)strip * d.strip_dwords + ((uint64_t)t * WIDE_LANES + l) * (K / 8) : nullptr;
                    wide_lane_step<K>(lanes[l], P, d, strip, l, t, l ? oh[l - 1] : 0, l ? of[l - 1] : 0, l ? oi[l - 1] : 0u, rec);
                }
            }
            for (uint32_t l = 0; l < WIDE_LANES; ++l) { unsigned long long key; if (wide_lane_best<K>(lanes[l], d, l, key) && key > P.best[i]) P.best[i] = key; }
        }
    }
    int run_gssw_wide(const WideParams& P, uint32_t n8, uint32_t n16) override {
        for (uint32_t k = 0; k < n8; ++k) wide_fill<8>(P, P.order[k]);
        for (uint32_t k = 0; k < n16; ++k) wide_fill<16>(P, P.order[n8 + k]);
        for (uint32_t i = 0; i < P.n; ++i) wide_walk_one(P, i);
        return VGK_OK;
    }
    int run_xdrop_band(const GsswMatrixParams& P) override {
        if (std::getenv("VGAMD_EMU_SKIP_XBAND")) {      // (host-side timing of the call on a machine without a GPU: the kernels answer "nothing aligned")
            if (P.xb_results) for (uint32_t k = 0; k < P.n; ++k) { vgk_result r{}; r.end_node = -1; r.end_offset = -1; r.end_read = -1; r.ops_begin = (uint32_t)P.xb_ops_off[k]; P.xb_results[k] = r; }
            return VGK_OK;
        }
        // the launch order's two classes: problems that run in one DPP row of 16 lanes (four to a wavefront on the GPU; here one after the
        // other — the rows do not talk to each other), then the ones that take a whole wavefront
        auto run = [&](uint32_t lanes, uint32_t begin, uint32_t count) {
            if (!count) return;
            XlShared sh; pthread_barrier_init(&sh.bar, nullptr, lanes);
            std::vector<std::thread> ts;
            for (uint32_t lane = 0; lane < lanes; ++lane) ts.emplace_back([&, lane]() {
                XlEmu xl{&sh, lane, lanes};
                for (uint32_t k = 0; k < count; ++k) { xdrop_band_wave_lane(P, P.xb_order ? P.xb_order[begin + k] : begin + k, lane, xl); xl.fence(); }
            });
            for (auto& t : ts) t.join();
            pthread_barrier_destroy(&sh.bar);
        };
        if (P.xb_order) { run(8, 0, P.xb_n8); run(16, P.xb_n8, P.xb_n16); run(64, P.xb_n8 + P.xb_n16, P.xb_n64); } else run(64, 0, P.n);
        if (P.xb_results) for (uint32_t k = 0; k < P.n; ++k) xdrop_band_walk_one(P, P.xb_order ? P.xb_order[k] : k);      // the walk kernel: one lane per problem
        return VGK_OK;
    }
    int run_gssw_matrix(const GsswMatrixParams& P) override {
        if (P.go < P.ge || std::getenv("VGAMD_MATRIX_THREADS")) { for (uint32_t i = 0; i < P.n; ++i) gssw_matrix_one(P, i); return VGK_OK; }
        matrix_wave_emu<2>(P, 0, 128); matrix_wave_emu<4>(P, 128, 256); matrix_wave_emu<8>(P, 256, 512); matrix_wave_emu<16>(P, 512, 1024);
        return VGK_OK;
    }
    int run_banded_geometry(const BGeomParams& p) override { for (uint32_t i = 0; i < p.n; ++i) banded_geometry_one(p, i); return VGK_OK; }
    int run_banded_multi(const BandedMultiParams& Q) override { for (uint32_t a = 0; a < Q.P.n; ++a) banded_multi_one(Q, a); return VGK_OK; }
    int run_gssw_multi(const GsswMultiParams& P) override { for (uint32_t i = 0; i < P.M.n; ++i) gssw_multi_one(P, i); return VGK_OK; }
    int gapless_order(const GOrderParams& P, int stage) override { for (uint32_t i = 0; i < P.n; ++i) { if (stage == 1) g_order_sizes_one(P, i); else g_order_gather_one(P, i); } return VGK_OK; }
    int gapless_seeded(const GSeededParams& P) override { for (uint32_t i = 0; i < P.n; ++i) g_seeded_one(P, i); return VGK_OK; }
    int sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n, int bits) override {
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (uint32_t i = 0; i < n; ++i) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
        return VGK_OK;
    }
    int mask_reads(char* reads, size_t bytes) override { for (size_t k = 0; k < bytes; ++k) reads[k] = g_mask_base(reads[k]); return VGK_OK; }
    int run_minimizer(const MinimizerParams& P) override {
        if (P.pass == 1) { for (uint32_t i = P.lo; i < P.hi; ++i) minimizer_one(P, i); } else for (uint32_t i = 0; i < P.n; ++i) minimizer_one(P, i);
        return VGK_OK;
    }
    int run_minimizer_list(const MzListParams& P) override { for (uint32_t i = 0; i <= P.n; ++i) mz_list_one(P, i); return VGK_OK; }
    int run_minimizer_seeds_of(const MzSeedsOfParams& P) override { for (uint32_t j = 0; j <= P.n; ++j) mz_seeds_of_one(P, j); return VGK_OK; }
    int run_wfa_mask(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, uint32_t n) override {
        for (uint32_t i = 0; i < n; ++i) for (uint32_t l = 0; l < 64; ++l) wfa_mask_one(probs, src_off, raw, seqs, i, l, 64);
        return VGK_OK;
    }
    int run_chain_stitch(const CsParams& P, int what) override {
        if (what == CS_GATHER) { for (uint32_t r = 0; r < P.n_reads; ++r) for (uint32_t l = 0; l < 64; ++l) cs_gather_one(P, r, l, 64); }
        else for (uint32_t r = 0; r <= P.n_reads; ++r) cs_one(P, what, r);
        return VGK_OK;
    }
    int run_rescue_requests(const RqParams& P, int what) override {
        if (what == RQ_FLAG) { for (uint32_t p = 0; p <= P.n_pairs; ++p) rq_flag_one(P, p); } else { for (uint32_t p = 0; p < P.n_pairs; ++p) rq_emit_one(P, p); }
        return VGK_OK;
    }
    int run_tail_stage(const TStageParams& P, int what) override { const uint32_t items = tstage_items(P, what); for (uint32_t i = 0; i < items; ++i) tstage_one(P, what, i); return VGK_OK; }
    int run_tail(const TailParams& P, uint32_t threads) override {
        for (uint32_t t = 0; t < threads; ++t) for (uint32_t i = t; i < P.n; i += threads) tail_walk_one(P, i, P.scratch[t]);
        return VGK_OK;
    }
    int scan_u32(const uint32_t* in, uint32_t* out, uint32_t n) override { uint32_t s = 0; for (uint32_t k = 0; k < n; ++k) { const uint32_t v = in[k]; out[k] = s; s += v; } return VGK_OK; }
    int forest_flags(const ForestParams& P) override { for (uint32_t v = 0; v < P.n_nodes; ++v) forest_flags_one(P, v); return VGK_OK; }
    int forest_emit(const ForestParams& P) override { for (uint32_t v = 0; v < P.n_nodes; ++v) forest_emit_one(P, v); return VGK_OK; }
    int fill(void* d, int byte, size_t n) override { std::memset(d, byte, n); return VGK_OK; }
    int run_wfa(const WfaParams& P, uint32_t threads) override {
        for (uint32_t t = 0; t < threads; ++t) wfa_thread(P, t, nullptr, 1);
        return VGK_OK;
    }
    // the wavefront form: the 64 lanes of ONE resident wavefront (slab 0), which takes every problem in turn.  The lanes are 64 fibers of
    // this thread, switched round-robin at every cross-lane operation: every lane executes the same sequence of such operations, so a
    // lane that has deposited its value and yielded once finds everyone else's value there when its turn comes again (two buffers, by
    // the parity of the operation's number: the lanes ahead are already depositing for the next one).  No OS threads, no barriers:
    // ~50 x faster than the barrier emulation above, and deterministic.
    struct FiberWave {
        static constexpr int N = 64;
        ucontext_t main_ctx, ctx[N]; std::vector<char> stack[N];
        int current = 0;
        int32_t buf[2][N]; unsigned long long wide[2][N]; uint32_t epoch[N];
        const WwParams* P = nullptr; WwSharedBoth shared;
        void yield() { const int me = current; current = (me + 1) % N; swapcontext(&ctx[me], &ctx[current]); }
    };
    struct WwXlEmu {
        FiberWave* w; uint32_t lane;
        int slot() { return (int)(w->epoch[lane]++ & 1u); }
        unsigned long long ballot(bool flag) {
            const int p = slot(); w->buf[p][lane] = flag ? 1 : 0; w->yield();
            unsigned long long m = 0; for (uint32_t l = 0; l < 64; ++l) if (w->buf[p][l]) m |= 1ull << l;
            return m;
        }
        uint32_t bcast(uint32_t v, uint32_t src) { const int p = slot(); w->buf[p][lane] = (int32_t)v; w->yield(); return (uint32_t)w->buf[p][src]; }
        unsigned long long reduce_min_u64(unsigned long long v) {
            const int p = slot(); w->wide[p][lane] = v; w->yield();
            unsigned long long m = ~0ull; for (uint32_t l = 0; l < 64; ++l) m = w->wide[p][l] < m ? w->wide[p][l] : m;
            return m;
        }
        int32_t reduce_max(int32_t v) {
            const int p = slot(); w->buf[p][lane] = v; w->yield();
            int32_t m = w->buf[p][0]; for (uint32_t l = 1; l < 64; ++l) m = w->buf[p][l] > m ? w->buf[p][l] : m;
            return m;
        }
        void fence() { slot(); w->yield(); }
        void fence_lds() { slot(); w->yield(); }
        unsigned long long load64(const unsigned long long* p) { return *p; }
        void store64(unsigned long long* p, unsigned long long v) { *p = v; }
        unsigned long long cas64(unsigned long long* p, unsigned long long expect, unsigned long long desired) { const unsigned long long old = *p; if (old == expect) *p = desired; return old; }
        uint32_t add32(uint32_t* p, uint32_t v) { const uint32_t old = *p; *p += v; return old; }
        uint32_t clock_us() { return 0u; }
        void or32(uint32_t* p, uint32_t v) { *p |= v; }
        uint32_t load32(const uint32_t* p) { return *p; }
    };
    static void fiber_entry(unsigned lo, unsigned hi, unsigned lane) {
        FiberWave* w = reinterpret_cast<FiberWave*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        WwXlEmu xl{w, lane};
        wfa_wave(*w->P, 0, lane, w->shared, xl);
    }
    int run_wfa_wave(const WwParams& P, uint32_t waves) override {
        if (!P.n_todo || !waves) return VGK_OK;
        std::unique_ptr<FiberWave> w(new FiberWave());
        w->P = &P;
        for (int l = 0; l < FiberWave::N; ++l) {
            w->epoch[l] = 0; w->stack[l].resize(512 * 1024);
            getcontext(&w->ctx[l]);
            w->ctx[l].uc_stack.ss_sp = w->stack[l].data(); w->ctx[l].uc_stack.ss_size = w->stack[l].size();
            w->ctx[l].uc_link = l + 1 < FiberWave::N ? &w->ctx[l + 1] : &w->main_ctx;      // a lane that returns hands over to the next; the last one to the caller
            const uintptr_t ptr = (uintptr_t)w.get();
            makecontext(&w->ctx[l], (void (*)())fiber_entry, 3, (unsigned)(ptr & 0xffffffffu), (unsigned)(ptr >> 32), (unsigned)l);
        }
        w->current = 0;
        swapcontext(&w->main_ctx, &w->ctx[0]);
        return VGK_OK;
    }
    int run_gapless(const GaplessParams& P, uint32_t threads) override {
        // the fast store first (here a plain local array, stride 1), then the slab store for the reads that outgrew it — as on the device
        const bool slab_only = std::getenv("VGAMD_GAPLESS_SLAB_ONLY") != nullptr;
        std::vector<uint32_t> lds(G_FAST_DW);
        if (!slab_only) {
            // the flat form: a lane takes reads from the counter until there are none (here the lanes run one after another, so lane t takes
            // every (threads)-th share by stopping after its part), then the rules kernel, then the slab kernel for the G_RETRY reads
            struct Wave {
                uint32_t left;
                GProf* prof() const { return nullptr; }
                int vote(bool idle, bool searching) const { return idle ? 1 : searching ? 2 : 0; }
                uint32_t next_read(const GaplessParams& p) { if (!left) return 0xffffffffu; const unsigned long long k = p.counters[4]; if (k >= p.n) return 0xffffffffu; ++p.counters[4]; --left; return (uint32_t)k; }
            };
            for (uint32_t t = 0; t < threads; ++t) {
                Wave w{(P.n + threads - 1) / threads};
                GStoreLds Q{lds.data(), 1u, P.scratch[t], 0u};
                gapless_search_lane<true>(P, Q, P.scratch[t], w);
            }
            for (uint32_t t = 0; t < threads; ++t) for (uint32_t k = t; k < P.n; k += threads) gapless_rules_one<true>(P, P.order[k], P.scratch[t].order);
            for (uint32_t t = 0; t < threads; ++t) for (uint32_t k = t; k < P.n; k += threads) {
                const uint32_t i = P.order[k];
                if (P.retry ? P.retry[i] != 0 : P.results[i].status == G_RETRY) { GStoreSlab Q{P.scratch[t]}; gapless_extend_one<true>(P, i, Q, P.scratch[t], P.cold[t]); }
            }
            return VGK_OK;
        }
        for (uint32_t t = 0; t < threads; ++t) for (uint32_t k = t; k < P.n; k += threads) { GStoreSlab Q{P.scratch[t]}; gapless_extend_one<true>(P, P.order[k], Q, P.scratch[t], P.cold[t]); }
        return VGK_OK;
    }
    int run_banded(const BandedParams& P, const BandedLaunch* launches, uint32_t n) override {
        if (std::getenv("VGAMD_EMU_SKIP_BANDED")) {     // (host-side timing of the call on a machine without a GPU: every problem answers "no alignment in the band")
            for (uint32_t a = 0; a < P.n; ++a) { BResult r{}; r.status = VGK_ENOBAND; P.results[a] = r; }
            return VGK_OK;
        }
        for (uint32_t i = 0; i < n; ++i) {
            const BandedLaunch& L = launches[i];
            switch (L.R) {
                case 1: if (P.quals) banded_fill_emu<1, true>(P, L.begin, L.count); else banded_fill_emu<1, false>(P, L.begin, L.count); break;
                case 2: if (P.quals) banded_fill_emu<2, true>(P, L.begin, L.count); else banded_fill_emu<2, false>(P, L.begin, L.count); break;
                case 4: if (P.quals) banded_fill_emu<4, true>(P, L.begin, L.count); else banded_fill_emu<4, false>(P, L.begin, L.count); break;
                case 8: if (P.quals) banded_fill_emu<8, true>(P, L.begin, L.count); else banded_fill_emu<8, false>(P, L.begin, L.count); break;
                case 16: if (P.quals) banded_fill_emu<16, true>(P, L.begin, L.count); else banded_fill_emu<16, false>(P, L.begin, L.count); break;
                case 32: if (P.quals) banded_fill_emu<32, true>(P, L.begin, L.count); else banded_fill_emu<32, false>(P, L.begin, L.count); break;
                case 64: if (P.quals) banded_fill_emu<64, true>(P, L.begin, L.count); else banded_fill_emu<64, false>(P, L.begin, L.count); break;
                case 128: if (P.quals) banded_fill_emu<128, true>(P, L.begin, L.count); else banded_fill_emu<128, false>(P, L.begin, L.count); break;
                case 256: if (P.quals) banded_fill_emu<256, true>(P, L.begin, L.count); else banded_fill_emu<256, false>(P, L.begin, L.count); break;
                case 512: if (P.quals) banded_fill_emu<512, true>(P, L.begin, L.count); else banded_fill_emu<512, false>(P, L.begin, L.count); break;
                default: return VGK_EINVAL;
            }
        }
        if (!P.scores) for (uint32_t i = 0; i < P.n; ++i) banded_walk_one(P, i);
        return VGK_OK;
    }
    const char* name() const override { return "cpu-lockstep-emulator"; }
    int compute_units() const override { return 0; }
    size_t memory_bytes() const override { return 0; }
    void* alloc(size_t bytes) override { return std::calloc(bytes ? bytes : 16, 1); }
    void release(void* p) override { std::free(p); }
    void* host_alloc(size_t bytes) override { return std::malloc(bytes ? bytes : 16); }
    void host_release(void* p) override { std::free(p); }
    int upload(void* d, const void* s, size_t n) override { std::memcpy(d, s, n); return VGK_OK; }
    int download(void* d, const void* s, size_t n) override { std::memcpy(d, s, n); return VGK_OK; }
    int zero(void* d, size_t n) override { std::memset(d, 0, n); return VGK_OK; }
    int sync() override { return VGK_OK; }
    // device-side packing of window problems: the same per-problem / per-wavefront functions, in loops
    int fill_side(void* d, int byte, size_t n) override { std::memset(d, byte, n); return VGK_OK; }
    int win_stage1(const WinParams& P, void*, size_t) override {
        WinAcc acc; win_acc_clear(acc);
        for (uint32_t i = 0; i < P.n; ++i) win_size_one(P, i, acc);
        win_acc_flush(P, acc);
        const uint32_t n1 = P.n + 1;
        for (uint32_t k = 0; k < WIN_COLS; ++k) {
            P.sizes[k * n1 + P.n] = 0;
            uint32_t at = 0;
            for (uint32_t i = 0; i < n1; ++i) { P.offs[k * n1 + i] = at; at += P.sizes[k * n1 + i]; }
        }
        return VGK_OK;
    }
    int win_stage2(const WinParams& P, void*, size_t) override {
        if (!P.n) return VGK_OK;
        std::vector<uint32_t> perm(P.n);
        for (uint32_t i = 0; i < P.n; ++i) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return (P.key[a] & 0x1ffffffu) < (P.key[b] & 0x1ffffffu); });
        for (uint32_t j = 0; j < P.n; ++j) { P.key_sorted[j] = P.key[perm[j]]; P.idx_sorted[j] = P.idx[perm[j]]; }
        std::memset(P.bucket_first, 0xff, sizeof(uint32_t) * WIN_BUCKETS);
        std::memset(P.wave_tb, 0, sizeof(unsigned long long) * ((size_t)P.n_waves_cap + 1));
        for (uint32_t j = 0; j < P.n; ++j) win_bucket_first_one(P, j);
        win_buckets(P);
        for (uint32_t w = 0; w < P.n_waves_cap; ++w) win_wave_one(P, w);
        unsigned long long at = 0;
        for (uint32_t w = 0; w <= P.n_waves_cap; ++w) { const unsigned long long s = P.wave_tb[w]; P.wave_tb[w] = at; at += s; }
        for (uint32_t w = 0; w < P.n_waves_cap; ++w) win_wave_tb_one(P, w);
        for (uint32_t i = 0; i < P.n; ++i) win_emit_one(P, i, 0, 1);
        return VGK_OK;
    }
    template <int K, bool S8> void fill(const GsswParams& P) {
        std::vector<Lane<K>> lanes(64);
        std::vector<uint32_t> oh(64), of(64), oi(64);
        constexpr uint32_t REC = (K + 3) / 4;
        for (uint32_t w = P.wave_begin; w < P.wave_begin + P.wave_count; ++w) {
            const WaveDesc wd = P.waves[w];
            for (uint32_t l = 0; l < 64; ++l) lane_init(lanes[l], P, wd, l);
            uint32_t steady_from = 0, steady_to = 0;                       // (the first fill's steady middle, as in the kernel)
            { uint32_t shortest = 0xffffffffu; for (uint32_t l = 0; l < 64; ++l) shortest = std::min(shortest, std::min(lanes[l].RA, lanes[l].RB)); steady_steps(wd.G, shortest, wd.n_steps, steady_from, steady_to); }
            for (uint32_t t = 0; t < wd.n_steps; ++t) {
                for (uint32_t l = 0; l < 64; ++l) { oh[l] = lanes[l].out_h; of[l] = lanes[l].out_f; oi[l] = lanes[l].info; }
                for (uint32_t l = 0; l < 64; ++l) {
                    if ((t & 3u) == 0) lane_prefetch(lanes[l], P, t);
                    const uint32_t rh = l ? oh[l - 1] : 0, rf = l ? of[l - 1] : 0, ri = l ? oi[l - 1] : 0;
                    if (P.spec_fill == 1) {                                    // the speculative batch's first fill: no codes
                        const bool av = t >= steady_from && t < steady_to;
                        if (S8 && P.key3) { if (av) lane_step<K, S8, false, true, false, true>(lanes[l], P, t, rh, rf, ri, nullptr, nullptr); else lane_step<K, S8, false, true>(lanes[l], P, t, rh, rf, ri, nullptr, nullptr); }
                        else { if (av) lane_step<K, S8, false, false, false, true>(lanes[l], P, t, rh, rf, ri, nullptr, nullptr); else lane_step<K, S8, false>(lanes[l], P, t, rh, rf, ri, nullptr, nullptr); }
                        continue;
                    }
                    if (P.tb_mode == TB_REWALK) {
                        lane_step<K, S8, false>(lanes[l], P, t, rh, rf, ri, nullptr, nullptr);
                        if (P.want_tb) { lane_store_boundary<K>(lanes[l], P, wd, t, l); lane_store_checkpoint<K>(lanes[l], P, wd, t, l); }
                        continue;
                    }
                    uint32_t* tb_a = P.want_tb ? P.tb + tb_dword(wd.tb_off, t, l, REC, 0) : nullptr;
                    uint32_t* tb_b = P.want_tb && REC > 4 ? P.tb + tb_dword(wd.tb_off, t, l, REC, 4) : nullptr;
                    lane_step<K, S8>(lanes[l], P, t, rh, rf, ri, tb_a, tb_b);
                }
            }
            for (uint32_t l = 0; l < 64; ++l) for (int half = 0; half < 2; ++half) {
                uint32_t prob; unsigned long long key;
                if (lane_best(lanes[l], half, prob, key) && key > P.best[prob]) P.best[prob] = key;
            }
        }
    }
    int run_gssw(const GsswParams& P0, const FillLaunch* launches, uint32_t n, bool walk) override {
        GsswParams P = P0;
        if (P0.restore_probs) for (uint32_t i = 0; i < P0.n_problems; ++i) refill_restore_one(P0, i);
        for (uint32_t i = 0; i < n; ++i) {
            const FillLaunch& L = launches[i];
            P.K = L.K; P.wave_begin = L.wave_begin; P.wave_count = L.wave_count;
            switch (P.K) {
                case 16: if (P.scale == 8) fill<16, true>(P); else fill<16, false>(P); break;
                case 19: if (P.scale == 8) fill<19, true>(P); else fill<19, false>(P); break;
                case 20: if (P.scale == 8) fill<20, true>(P); else fill<20, false>(P); break;
                case 24: if (P.scale == 8) fill<24, true>(P); else fill<24, false>(P); break;
                default: return VGK_EINVAL;
            }
        }
        if (walk && P.tb_mode == TB_REWALK) {
            // the band: every lane of every wavefront again over its band columns; the walks over the band records; then, for the reads
            // whose walk left its band, the on-demand form
            for (uint32_t li = 0; li < n; ++li) {
                const FillLaunch& L = launches[li];
                for (uint32_t w = L.wave_begin; w < L.wave_begin + L.wave_count; ++w) for (uint32_t l = 0; l < 64; ++l) {
                    const WaveDesc wd = P.waves[w]; const bool s8 = P.scale == 8;
                    switch (L.K) {
                        case 16: if (s8) band_fill_lane<16, true>(P, wd, l); else band_fill_lane<16, false>(P, wd, l); break;
                        case 19: if (s8) band_fill_lane<19, true>(P, wd, l); else band_fill_lane<19, false>(P, wd, l); break;
                        case 20: if (s8) band_fill_lane<20, true>(P, wd, l); else band_fill_lane<20, false>(P, wd, l); break;
                        case 24: if (s8) band_fill_lane<24, true>(P, wd, l); else band_fill_lane<24, false>(P, wd, l); break;
                        default: return VGK_EINVAL;
                    }
                }
            }
            for (uint32_t i = 0; i < P.n_problems; ++i) bandwalk_one(P, i, P.best[i]);
            std::vector<uint32_t> win((TB_CKPT / 2) * 6);          // one lane's window (stride 1)
            const uint32_t n_missed = *tb_miss_count(P);
            band_misses += n_missed;
            for (uint32_t km = 0; km < n_missed; ++km) {
                const uint32_t i = tb_miss_list(P)[km];
                const uint32_t K = P.probs[i].geom & 0xffu; const bool s8 = P.scale == 8;
                switch (K) {
                    case 16: if (s8) rewalk_one<16, true>(P, i, P.best[i], win.data(), 1); else rewalk_one<16, false>(P, i, P.best[i], win.data(), 1); break;
                    case 19: if (s8) rewalk_one<19, true>(P, i, P.best[i], win.data(), 1); else rewalk_one<19, false>(P, i, P.best[i], win.data(), 1); break;
                    case 20: if (s8) rewalk_one<20, true>(P, i, P.best[i], win.data(), 1); else rewalk_one<20, false>(P, i, P.best[i], win.data(), 1); break;
                    case 24: if (s8) rewalk_one<24, true>(P, i, P.best[i], win.data(), 1); else rewalk_one<24, false>(P, i, P.best[i], win.data(), 1); break;
                    default: return VGK_EINVAL;
                }
            }
            band_walks += P.n_problems;
        } else if (walk && P.walk_passes == 2) {                 // the two kernels of the device: diagonal runs alone, then the reads on the miss list
            { uint32_t blk[2 * WD_DWORDS]; int16_t tab[WD_TAB]; for (uint32_t t = 0; t < WD_TAB; ++t) tab[t] = wd_table_entry(P, t);
              for (uint32_t i = 0; i < P.n_problems; ++i) walk_first_one(P, i, P.best[i], blk, 1, tab); }
            if (P.spec_fill) {                                    // the reads it left: wavefronts of their own, filled again with codes
                const uint32_t gpw = 64u / P.refill_G, max_waves = (P.n_pairs + gpw - 1u) / gpw;
                for (uint32_t w2 = 0; w2 < max_waves; ++w2) refill_layout_one(P, w2);
                GsswParams P2 = P; P2.spec_fill = 2; P2.K = P.refill_K; P2.wave_begin = P.refill_wave0; P2.wave_count = P.refill_count[0];
                switch (P2.K) {
                    case 16: if (P2.scale == 8) fill<16, true>(P2); else fill<16, false>(P2); break;
                    case 19: if (P2.scale == 8) fill<19, true>(P2); else fill<19, false>(P2); break;
                    case 20: if (P2.scale == 8) fill<20, true>(P2); else fill<20, false>(P2); break;
                    case 24: if (P2.scale == 8) fill<24, true>(P2); else fill<24, false>(P2); break;
                    default: return VGK_EINVAL;
                }
                spec_refilled += P.refill_count[0];
            }
            const uint32_t n_missed = *tb_miss_count(P);
            for (uint32_t km = 0; km < n_missed; ++km) { const uint32_t i = tb_miss_list(P)[km]; walk_one(P, i, P.best[i]); }
            walk_first_settled += P.n_problems - n_missed; walk_first_missed += n_missed;
        } else if (walk) for (uint32_t i = 0; i < P.n_problems; ++i) walk_one(P, i, P.best[i]);
        return VGK_OK;
    }
    unsigned long long band_misses = 0, band_walks = 0, walk_first_settled = 0, walk_first_missed = 0, spec_refilled = 0;
    ~EmuBackend() override {
        if (std::getenv("VGAMD_EMU_STATS") && band_walks) std::fprintf(stderr, "[emu] band walks %llu, left their band %llu\n", band_walks, band_misses);
        if (std::getenv("VGAMD_EMU_STATS") && walk_first_settled + walk_first_missed) std::fprintf(stderr, "[emu] two-pass walks: %llu settled by diagonal runs, %llu by their codes (%llu wavefronts filled again)\n", walk_first_settled, walk_first_missed, spec_refilled);
    }
    double last_ms(int which) const override { return which == 2 ? 1.0 : which == 8 ? (double)band_misses : which == 9 ? (double)band_walks : which == 10 ? (double)walk_first_settled : which == 11 ? (double)walk_first_missed : which == 12 ? (double)spec_refilled : 0.0; }   // (12: wavefronts the speculative fill laid out again, all runs so far)
};

Backend* make_backend(int, std::string&) { return new EmuBackend(); }

}  // namespace vgk
