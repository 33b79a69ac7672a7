// backend_hip.hip — HIP runtime plumbing + the gfx950 kernel entry points.
// There is deliberately no CPU fallback here: if no HIP device answers,
// make_backend() fails and vgk_create() returns VGK_ENODEV.
#include <chrono>
#include <thread>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include "backend.hpp"
#include "pack_hip.hpp"

namespace vgk {

// one DPP wave_shr:1 — lane l receives lane l-1's value (lane 0 receives 0: bound_ctrl, so no old value has to be set up)
static __device__ __forceinline__ uint32_t from_lane_above(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}

// Fill: 4 independent wavefronts per workgroup; each wavefront owns
// floor(64/G) read pairs for the whole skewed sweep.  No barriers; LDS only as the staging buffer of the traceback records (below).
// Occupancy: K = 16 fits 128 VGPRs (4 waves/SIMD); K = 20/24 keep 4K+ state registers per
// lane and run 3 waves/SIMD (<= 168 VGPRs) rather than spill in the hot loop.
// Traceback records on their way out (tiled layout, gssw_device.hpp): a wavefront keeps the records of TB_TILE = 8 steps in LDS — written
// step-major, one 16-byte slot per lane and step (slot index XOR step: the flush below then reads conflict-free too) plus the part-B
// dwords — and every eighth step copies the tile to HBM in the layout the walker reads: 8 + 2 (K = 19 / 20) whole-wave 1-KB bursts per
// tile instead of two partial stores per step.  10 KB of LDS per wavefront (K = 19 / 20; 8 for K = 16, 12 for K = 24): 3 (4) wavefronts per
// SIMD still fit a CU's 160 KB.  One wavefront owns its buffer and LDS executes a wavefront's instructions in order: no barrier.
template <int K>
struct TbStage {
    static constexpr uint32_t REC = (K + 3) / 4, RB = REC - 4;                // dwords per record; of them in part B
    static constexpr uint32_t DWORDS = TB_TILE * 64u * REC;
    uint32_t* lds;                                                            // this wavefront's buffer: [8][64] x 16 B, then [8][64] x RB dwords
    __device__ __forceinline__ uint32_t* slot_a(uint32_t t, uint32_t lane) const { const uint32_t s = t % TB_TILE; return lds + (s * 64u + (lane ^ s)) * 4u; }
    __device__ __forceinline__ uint32_t* slot_b(uint32_t t, uint32_t lane) const { return lds + TB_TILE * 64u * 4u + ((t % TB_TILE) * 64u + lane) * RB; }
    __device__ __forceinline__ void flush(uint32_t* tile, uint32_t lane) const {      // tile = where this tile starts in HBM
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t j = 0; j < TB_TILE; ++j) {                              // part A: 16-byte slot g of the tile = (lane g / 8, step g % 8)
            const uint32_t g = j * 64u + lane, L = g / TB_TILE, s = g % TB_TILE;
            const uint4 v = *reinterpret_cast<const uint4*>(lds + (s * 64u + (L ^ s)) * 4u);
            reinterpret_cast<uint4*>(tile)[g] = v;
        }
        if constexpr (RB > 0) {
            uint32_t* part_b = tile + TB_TILE * 64u * 4u;
            const uint32_t* lb = lds + TB_TILE * 64u * 4u;
#pragma unroll
            for (uint32_t j = 0; j < 2u * RB; ++j) {                          // part B: dword d of it = (record d / RB, component d % RB), record r = (lane r / 8, step r % 8)
                const uint32_t q = j * 64u + lane;
                uint4 v; uint32_t* w = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                for (uint32_t c = 0; c < 4u; ++c) { const uint32_t d = 4u * q + c, r = d / RB, L = r / TB_TILE, st = r % TB_TILE; w[c] = lb[(st * 64u + L) * RB + d % RB]; }
                reinterpret_cast<uint4*>(part_b)[q] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// CODES = false: the recurrence alone and nothing of a traceback (GsswParams::spec_fill's first fill)
// NOKEY: the second fill of a speculative batch (spec_fill == 2) — codes only; the end cells are the first fill's, nothing is tracked or published
template <int K, bool S8, bool REWALK, bool CODES = true, bool KEY3 = false, bool NOKEY = false>
__global__ __launch_bounds__(256, (K <= 16 ? 4 : 3)) void gssw_fill_kernel(const GsswParams P) {
    constexpr uint32_t REC = (K + 3) / 4;   // dwords per (step, lane) traceback record
    __shared__ __attribute__((aligned(16))) uint32_t stage_lds[4][REWALK ? 64u * TB_BND_CHUNK * 2u : (TB_TILE > 1 ? TbStage<K>::DWORDS : 256u)];      // (also the fused walk's best keys, below)
    const uint32_t wave = P.wave_begin + blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= P.wave_begin + P.wave_count) return;
    if (P.wave_limit && wave - P.wave_begin >= *P.wave_limit) return;
    const WaveDesc wd = P.waves[wave];
    Lane<K> s;
#if VGK_PB_LDS
    __shared__ uint32_t pb_lds[4][64 * K];      // read B's profile words, [row][lane] per wavefront (gssw_device.hpp: VGK_PB_LDS)
    s.PBL = (typename Lane<K>::lds_u32*)(pb_lds[threadIdx.x >> 6] + lane);
#endif
    lane_init(s, P, wd, lane);
    asm volatile("" : "+v"(s.one));   // keep min(x,1) a packed min instead of cmp+cndmask
    uint32_t* tb = P.want_tb ? P.tb : nullptr;
    TbStage<K> stage{stage_lds[threadIdx.x >> 6]};
    // the first fill of a speculative batch runs its steady middle — every lane with a column of both reads — through a copy of the step that tests nothing for
    // being there (gssw_device.hpp, AV)
    uint32_t steady_from = 0, steady_to = 0;
    if constexpr (!CODES) {
        uint32_t shortest = s.RA < s.RB ? s.RA : s.RB;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)shortest, d, 64); shortest = o < shortest ? o : shortest; }
        steady_steps(wd.G, shortest, wd.n_steps, steady_from, steady_to);
        steady_from = (uint32_t)__builtin_amdgcn_readfirstlane((int)steady_from); steady_to = (uint32_t)__builtin_amdgcn_readfirstlane((int)steady_to);
    }
    if constexpr (!CODES) {
        // three loops, not a branch in one: the step's two copies share no block, and each loop keeps the registers of one
        uint32_t t = 0;
        for (; t < steady_from; ++t) {
            if ((t & 3u) == 0) lane_prefetch(s, P, t);
            const uint32_t rh = from_lane_above(s.out_h), rf = from_lane_above(s.out_f), ri = from_lane_above(s.info);
            lane_step<K, S8, false, KEY3>(s, P, t, rh, rf, ri, nullptr, nullptr);
        }
        for (; t < steady_to; ++t) {
            if ((t & 3u) == 0) lane_prefetch(s, P, t);
            const uint32_t rh = from_lane_above(s.out_h), rf = from_lane_above(s.out_f), ri = from_lane_above(s.info);
            lane_step<K, S8, false, KEY3, false, true>(s, P, t, rh, rf, ri, nullptr, nullptr);
        }
        for (; t < wd.n_steps; ++t) {
            if ((t & 3u) == 0) lane_prefetch(s, P, t);
            const uint32_t rh = from_lane_above(s.out_h), rf = from_lane_above(s.out_f), ri = from_lane_above(s.info);
            lane_step<K, S8, false, KEY3>(s, P, t, rh, rf, ri, nullptr, nullptr);
        }
    }
    for (uint32_t t = 0; CODES && t < wd.n_steps; ++t) {
        if ((t & 3u) == 0) lane_prefetch(s, P, t);
        const uint32_t rh = from_lane_above(s.out_h);
        const uint32_t rf = from_lane_above(s.out_f);
        const uint32_t ri = from_lane_above(s.info);
        if constexpr (!CODES) {
        } else if constexpr (REWALK) {
            // the recurrence alone; what the traceback needs to run a window of it again (gssw_device.hpp, TB_REWALK)
            lane_step<K, S8, false>(s, P, t, rh, rf, ri, nullptr, nullptr);
            if (tb) {
                // boundary rows: TB_BND_CHUNK steps gathered per lane in LDS (slot index XOR lane: the per-step writes and the flush's reads
                // both spread over the banks), then each lane's chunk leaves as one 128-byte line of the lane-major layout
                uint32_t* st = stage_lds[threadIdx.x >> 6] + lane * (TB_BND_CHUNK * 2u);
                const uint32_t slot = (t ^ lane) & (TB_BND_CHUNK - 1u);
                *reinterpret_cast<uint2*>(st + slot * 2u) = make_uint2(s.out_h, s.out_f);
                if ((t & (TB_BND_CHUNK - 1u)) == TB_BND_CHUNK - 1u || t + 1u == wd.n_steps) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    uint4* out = reinterpret_cast<uint4*>(tb + tb_bnd(wd.tb_off, wd.n_steps, t & ~(TB_BND_CHUNK - 1u), lane));
#pragma unroll
                    for (uint32_t j = 0; j < TB_BND_CHUNK; j += 2) {
                        const uint2 a = *reinterpret_cast<const uint2*>(st + ((j ^ lane) & (TB_BND_CHUNK - 1u)) * 2u), b = *reinterpret_cast<const uint2*>(st + (((j + 1u) ^ lane) & (TB_BND_CHUNK - 1u)) * 2u);
                        out[j >> 1] = make_uint4(a.x, a.y, b.x, b.y);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                lane_store_checkpoint<K>(s, P, wd, t, lane);
            }
        } else if constexpr (TB_TILE > 1) {
            lane_step<K, S8, true, false, NOKEY>(s, P, t, rh, rf, ri, tb ? stage.slot_a(t, lane) : nullptr, stage.slot_b(t, lane));
            if (tb && ((t % TB_TILE) == TB_TILE - 1u || t + 1u == wd.n_steps)) stage.flush(tb + tb_tile_base(wd.tb_off, t, REC), lane);
        } else
            lane_step<K, S8, true, false, NOKEY>(s, P, t, rh, rf, ri, tb ? tb + tb_dword(wd.tb_off, t, lane, REC, 0) : nullptr, tb ? tb + tb_dword(wd.tb_off, t, lane, REC, 4) : nullptr);
    }
    if constexpr (NOKEY) return;
    if (REWALK || !CODES || !P.fused) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t prob; unsigned long long key;
            if (lane_best(s, half, prob, key)) atomicMax(&P.best[prob], key);
        }
        return;
    }
    // Fused traceback: the wavefront that filled a read pair also walks it back while its
    // traceback codes are still in L2 / Infinity Cache.  The walk is a latency-bound pointer
    // chase (one lane per read); the other wavefronts resident on the SIMD keep the VALU busy.
    unsigned long long* mine = reinterpret_cast<unsigned long long*>(stage_lds[threadIdx.x >> 6]);      // 128 keys: 1 KB of the wavefront's staging buffer
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t prob; unsigned long long key = 0;
        if (!lane_best(s, half, prob, key)) key = 0;
        mine[lane * 2 + half] = key;
    }
    __threadfence();      // this wave's traceback codes / scratch columns -> visible to its walker lanes
    const uint32_t n_slots = 2u * (64u / wd.G);
    for (uint32_t slot = lane; slot < n_slots; slot += 64u) {
        const uint32_t q = slot >> 1, half = slot & 1u;
        if (wd.first_pair + q >= wd.pair_end) continue;
        const uint32_t prob = P.order[2u * (wd.first_pair + q) + half];
        if (prob == 0xffffffffu) continue;
        unsigned long long key = 0;
        for (uint32_t g = 0; g < wd.G; ++g) { const unsigned long long k = mine[(q * wd.G + g) * 2 + half]; key = k > key ? k : key; }
        walk_one(P, prob, key);
    }
}

// One thread per read, in the order the fill placed the reads (order[] = the reads of wavefront 0, pair by pair, then wavefront 1 ...):
// the threads of a walker wavefront then read traceback records that one or two fill wavefronts wrote next to each other, and the two
// reads of a pair — whose codes share every dword — sit in neighbouring lanes.
#ifndef VGK_WALK_WAVES
#define VGK_WALK_WAVES 0
#endif
#if VGK_WALK_WAVES
__global__ __launch_bounds__(256, VGK_WALK_WAVES) void gssw_walk_kernel(const GsswParams P, const int in_fill_order) {
#else
__global__ __launch_bounds__(256) void gssw_walk_kernel(const GsswParams P, const int in_fill_order) {
#endif
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (!in_fill_order) { if (k < P.n_problems) walk_one(P, k, P.best[k]); return; }
    if (k >= 2u * P.n_pairs) return;
    const uint32_t i = P.order[k];
    if (i != 0xffffffffu) walk_one(P, i, P.best[i]);
}
// the band geometry of a batch of banded problems (banded_geom_device.hpp): a lane per problem
__global__ __launch_bounds__(64) void banded_geometry_kernel(const BGeomParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n) banded_geometry_one(P, i);
}

// The tracebacks as two kernels (GsswParams::walk_passes == 2): every read by diagonal runs alone; then the reads that needed a code — one in
// eight on the headline batch — side by side, so that the wavefronts of the first kernel are never held by a lane that walks cell by cell
__global__ __launch_bounds__(64) void gssw_walk_first_kernel(const GsswParams P, const int in_fill_order) {
    __shared__ uint32_t blk[2 * WD_DWORDS * 64];                   // a lane's 160 read bytes and 160 column bytes, dwords interleaved across the lanes
    __shared__ int16_t tab[WD_TAB];
    if (threadIdx.x < WD_TAB) tab[threadIdx.x] = wd_table_entry(P, threadIdx.x);
    __syncthreads();
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (!in_fill_order) { if (k < P.n_problems) walk_first_one(P, k, P.best[k], blk + threadIdx.x, 64, tab); return; }
    if (k >= 2u * P.n_pairs) return;
    const uint32_t i = P.order[k];
    if (i != 0xffffffffu) walk_first_one(P, i, P.best[i], blk + threadIdx.x, 64, tab);
}
__global__ __launch_bounds__(64) void gssw_refill_layout_kernel(const GsswParams P) {
    refill_layout_one(P, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ __launch_bounds__(256) void gssw_refill_restore_kernel(const GsswParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_problems) refill_restore_one(P, i);
}
__global__ __launch_bounds__(256) void gssw_walk_missed_kernel(const GsswParams P) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *tb_miss_count(P)) return;
    const uint32_t i = tb_miss_list(P)[k];
    walk_one(P, i, P.best[i]);
}

// TB_REWALK (gssw_device.hpp), first the band: the fill's wavefronts again — same grid, same lane <-> (pair, lane block) map — each lane over
// the band columns of its lane block with the code-building lane code; no lane talks to another (the rows above come from HBM).
template <int K, bool S8>
__global__ __launch_bounds__(256, 2) void gssw_band_kernel(const GsswParams P) {
    const uint32_t wave = P.wave_begin + blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= P.wave_begin + P.wave_count) return;
    band_fill_lane<K, S8>(P, P.waves[wave], threadIdx.x & 63u);
}
// ... then the walks over the band's records, one lane per read in fill order (the stored-codes walk kernel's shape)
__global__ __launch_bounds__(256) void gssw_bandwalk_kernel(const GsswParams P) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2u * P.n_pairs) return;
    const uint32_t i = P.order[k];
    if (i != 0xffffffffu) bandwalk_one(P, i, P.best[i]);
}
// ... then, for the reads whose walk left its band (status W_MISSED), the on-demand form: one lane per read over the reads of ONE fill launch (the lane code is
// instantiated per rows-per-lane K), in the order the fill placed them.  A wavefront's windows live in LDS, lane-interleaved.
template <int K, bool S8>
__global__ __launch_bounds__(64, 2) void gssw_rewalk_kernel(const GsswParams P) {
    __shared__ uint32_t win[(TB_CKPT / 2) * ((K + 3) / 4) * 64];
    const uint32_t count = *tb_miss_count(P);
    const uint32_t* list = tb_miss_list(P);
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < count; k += gridDim.x * 64u) {
        const uint32_t i = list[k];
        rewalk_one<K, S8>(P, i, P.best[i], win + threadIdx.x, 64u);
    }
}

// ---- CIGAR ops on their way back: exclusive prefix sums of the per-problem op counts, then a gather ------------------
// One block scans OPS_SCAN_BLOCK = 1024 problems (4 per thread): wave-level inclusive scans on DPP-backed shuffles, the four
// wave totals through LDS.
static __device__ __forceinline__ uint32_t ops_of(const vgk_result* res, uint32_t i, uint32_t n) {
    return (i < n && res[i].status == VGK_OK) ? res[i].n_ops : 0u;
}
__global__ __launch_bounds__(256) void ops_scan_kernel(const vgk_result* res, uint32_t n, uint32_t* offs, uint32_t* sums) {
    __shared__ uint32_t wave_total[4];
    const uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u, lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t c0 = ops_of(res, i0, n), c1 = ops_of(res, i0 + 1, n), c2 = ops_of(res, i0 + 2, n), c3 = ops_of(res, i0 + 3, n);
    uint32_t incl = c0 + c1 + c2 + c3;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
    if (lane == 63) wave_total[w] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < w; ++k) before += wave_total[k];
    uint32_t at = before + incl - (c0 + c1 + c2 + c3);
    if (i0 < n) offs[i0] = at;
    at += c0; if (i0 + 1 < n) offs[i0 + 1] = at;
    at += c1; if (i0 + 2 < n) offs[i0 + 2] = at;
    at += c2; if (i0 + 3 < n) offs[i0 + 3] = at;
    if (threadIdx.x == 255) sums[blockIdx.x] = before + incl;
}
// one block turns the block totals into block starts (exclusive) and the grand total
__global__ __launch_bounds__(256) void ops_sums_kernel(uint32_t* sums, uint32_t n_blocks, unsigned long long* total) {
    __shared__ uint32_t wave_total[4];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < n_blocks ? sums[i] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
        if (lane == 63) wave_total[w] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t k = 0; k < w; ++k) before += wave_total[k];
        const unsigned long long start = carry + before + incl - c;
        if (i < n_blocks) sums[i] = (uint32_t)start;         // the caller has checked that all ops fit 32 bits
        __syncthreads();
        if (threadIdx.x == 255) carry += before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void ops_gather_kernel(const vgk_result* res, const vgk_op* ops, uint32_t n, const uint32_t* offs,
                                                         const uint32_t* sums, vgk_result* out_res, vgk_op* out_ops) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vgk_result r = res[i];
    const uint32_t at = sums[i / Backend::OPS_SCAN_BLOCK] + offs[i];
    if (r.status == VGK_OK && r.n_ops) {
        const vgk_op* from = ops + r.ops_begin;
        for (uint32_t k = 0; k < r.n_ops; ++k) out_ops[at + k] = from[k];
    } else r.n_ops = 0;
    r.ops_begin = at;
    out_res[i] = r;
}

// ---- banded global alignment (banded_device.hpp): cross-lane primitives on DPP ------------------------------------
struct XlDpp {
    static __device__ __forceinline__ int32_t dpp_up(int32_t v)   { return __builtin_amdgcn_update_dpp(BNEG, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
    static __device__ __forceinline__ int32_t dpp_down(int32_t v) { return __builtin_amdgcn_update_dpp(BNEG, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
    __device__ __forceinline__ int32_t up(int32_t v) const { return dpp_up(v); }
    __device__ __forceinline__ int32_t down(int32_t v) const { return dpp_down(v); }
    // lane +- 1's value minus s in ONE instruction; the lane without that neighbour has no DPP source, is skipped and keeps `old`
    // (s_nop: 2 wait states between a VALU write and a DPP read of the same VGPR)
    __device__ __forceinline__ int32_t up_sub(int32_t old, int32_t v, int32_t s) const {
        asm volatile("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(v), "v"(s));
        return old;
    }
    __device__ __forceinline__ int32_t down_sub(int32_t old, int32_t v, int32_t s) const {
        asm volatile("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(v), "v"(s));
        return old;
    }
    __device__ __forceinline__ int32_t scalar(int32_t s) const { int32_t o; const int32_t u = __builtin_amdgcn_readfirstlane(s); asm volatile("s_mov_b32 %0, %1" : "=s"(o) : "s"(u)); return o; }
    __device__ __forceinline__ int32_t add3(int32_t a, int32_t b, int32_t s) const { int32_t o; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "s"(s)); return o; }
    __device__ __forceinline__ int32_t in_lanes(int32_t s) const { int32_t v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }
    // exclusive max-scan over the 64 lanes: row_shr 1/2/4/8 inside each row of 16, then row_bcast:15 and row_bcast:31
    __device__ __forceinline__ int32_t scan_excl(int32_t v) const {
        // v_max_i32_dpp with the destination tied to both sources: a lane whose DPP source does not exist is disabled and
        // keeps its value, which is the identity we want — one instruction per step instead of mov_dpp + max + constant.
        // (s_nop: 2 wait states between a VALU write and a DPP read of the same VGPR; 5 after an EXEC write.)
        asm volatile("s_nop 4\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
        return dpp_down(v);
    }
    // the same with the last step's destination kept by the caller: lane 0 has no source and keeps `old`
    __device__ __forceinline__ int32_t scan_excl_keep(int32_t old, int32_t v) const {
        asm volatile("s_nop 4\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v), "+v"(old));
        return old;
    }
    __device__ __forceinline__ uint32_t width() const { return 64u; }
    __device__ __forceinline__ int32_t first_lane(int32_t v) const { return __builtin_amdgcn_readlane(v, 0); }
    __device__ __forceinline__ int32_t last_lane(int32_t v) const { return __builtin_amdgcn_readlane(v, 63); }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    __device__ __forceinline__ bool any(int32_t flag) const { return __ballot(flag != 0) != 0ull; }
    __device__ __forceinline__ unsigned long long ballot(bool flag) const { return __ballot(flag); }
    __device__ __forceinline__ int32_t reduce_max(int32_t v) const {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const int32_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
        return v;
    }
    __device__ __forceinline__ unsigned long long reduce_add(unsigned long long v) const {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        return v;
    }
};

template <int R, bool QA, bool LDS>
__global__ void __launch_bounds__(64) banded_fill_kernel(const BandedParams P, const uint32_t begin) {
    extern __shared__ uint8_t smem[];
    const BProb pb = P.probs[P.order[begin + blockIdx.x]];
    BSrc src;
    constexpr bool FAST = LDS && !QA && R <= 4;
    if (FAST) {      // the FAST score path (banded_device.hpp): padded copies of the read and graph codes, the table rows in scalar registers
        uint8_t* srd = smem + 48;                                            // 8 B of padding in front (windows start up to 8 rows early) ...
        uint8_t* sg = smem + ((48u + pb.L + 16u + 3u) & ~3u);                // ... and behind (a window is three dwords); graph codes dword-aligned
        for (uint32_t i = threadIdx.x; i < 48; i += 64) smem[i] = i < 40 ? (uint8_t)P.mat[BMAT_ROWS_AT + i] : (uint8_t)0;      // the table rows as 64-bit words, then the padding
        for (uint32_t i = threadIdx.x; i < pb.L + 16u; i += 64) srd[i] = i < pb.L ? P.reads[pb.read_off + i] : (uint8_t)0;
        for (uint32_t i = threadIdx.x; i < pb.graph_len + 8u; i += 64) sg[i] = i < pb.graph_len ? P.graph[pb.graph_off + i] : (uint8_t)0;
        __syncthreads();
        src.rd = srd; src.q = nullptr; src.graph = sg; src.mat = P.mat; src.rows = reinterpret_cast<const uint64_t*>(smem);
    } else if (LDS) {       // stage what every column reads — score table, read, qualities, graph bases — into LDS once
        constexpr uint32_t MAT = QA ? 6400u : 32u;
        int8_t* smat = reinterpret_cast<int8_t*>(smem);
        uint8_t* srd = smem + MAT; uint8_t* sq = srd + pb.L; uint8_t* sg = QA ? sq + pb.L : sq;
        for (uint32_t i = threadIdx.x; i < (QA ? 6400u : 25u); i += 64) smat[i] = P.mat[i];
        for (uint32_t i = threadIdx.x; i < pb.L; i += 64) { srd[i] = P.reads[pb.read_off + i]; if (QA) sq[i] = P.quals[pb.read_off + i]; }
        for (uint32_t i = threadIdx.x; i < pb.graph_len; i += 64) sg[i] = P.graph[pb.graph_off + i];
        __syncthreads();
        src.rd = srd; src.q = sq; src.graph = sg; src.mat = smat;
    } else {
        src.rd = P.reads + pb.read_off; src.q = QA ? P.quals + pb.read_off : nullptr; src.graph = P.graph + pb.graph_off; src.mat = P.mat;
    }
    XlDpp xl;
    banded_fill_lane<R, QA, FAST>(P, pb, src, threadIdx.x, xl);
}

// Wide bands (more than 512 diagonals): B blocks of 64 lanes x 8 rows per column, a lane's rows of the other blocks in LDS (banded_fill_lane_blocks)
// — instead of 16 / 32 rows per lane in registers, which needed 399-512 VGPRs and up to 1.6 KB of scratch per lane.
template <int B, bool QA, bool LDS>
__global__ void __launch_bounds__(64) banded_fill_blocks_kernel(const BandedParams P, const uint32_t begin) {
    extern __shared__ uint8_t smem[];
    __shared__ int32_t bstate[B * 3 * 8 * 64];          // element e of lane l at bstate[e * 64 + l]: no bank conflicts
    const BProb pb = P.probs[P.order[begin + blockIdx.x]];
    BSrc src;
    if (LDS) {       // stage what every column reads — score table, read, qualities, graph bases — into LDS once (as banded_fill_kernel)
        constexpr uint32_t MAT = QA ? 6400u : 32u;
        int8_t* smat = reinterpret_cast<int8_t*>(smem);
        uint8_t* srd = smem + MAT; uint8_t* sq = srd + pb.L; uint8_t* sg = QA ? sq + pb.L : sq;
        for (uint32_t i = threadIdx.x; i < (QA ? 6400u : 25u); i += 64) smat[i] = P.mat[i];
        for (uint32_t i = threadIdx.x; i < pb.L; i += 64) { srd[i] = P.reads[pb.read_off + i]; if (QA) sq[i] = P.quals[pb.read_off + i]; }
        for (uint32_t i = threadIdx.x; i < pb.graph_len; i += 64) sg[i] = P.graph[pb.graph_off + i];
        __syncthreads();
        src.rd = srd; src.q = sq; src.graph = sg; src.mat = smat;
    } else {
        src.rd = P.reads + pb.read_off; src.q = QA ? P.quals + pb.read_off : nullptr; src.graph = P.graph + pb.graph_off; src.mat = P.mat;
    }
    src.rows = nullptr;
    XlDpp xl;
    banded_fill_lane_blocks<B, QA>(P, pb, src, threadIdx.x, xl, bstate + threadIdx.x, 64u);
}
// Bands of more than 2048 diagonals: any number of blocks, their state in an HBM slab per wavefront (banded_fill_lane_blocks<0>); the inputs are read
// where they lie (such a problem's read and graph are long: no LDS staging)
template <bool QA>
__global__ void __launch_bounds__(64) banded_fill_blocks_any_kernel(const BandedParams P, const uint32_t begin, const uint32_t nb, int32_t* slabs) {
    const BProb pb = P.probs[P.order[begin + blockIdx.x]];
    BSrc src;
    src.rd = P.reads + pb.read_off; src.q = QA ? P.quals + pb.read_off : nullptr; src.graph = P.graph + pb.graph_off; src.mat = P.mat; src.rows = nullptr;
    XlDpp xl;
    int32_t* mine = slabs + (size_t)blockIdx.x * ((size_t)nb * 3u * 8u + 3u * ((size_t)nb + 1u)) * 64u;
    banded_fill_lane_blocks<0, QA>(P, pb, src, threadIdx.x, xl, mine + threadIdx.x, 64u, nb, mine + (size_t)nb * 3u * 8u * 64u + threadIdx.x);
}
template <int B>
static void launch_banded_fill_blocks(const BandedParams& p, const BandedLaunch& L, hipStream_t stream) {
    const dim3 grid(L.count), block(64);
    const bool qa = p.quals != nullptr;
    const uint32_t dyn = (L.lds_bytes && L.lds_bytes + (uint32_t)sizeof(int32_t) * B * 3 * 8 * 64 <= 64u * 1024u) ? L.lds_bytes : 0u;      // (the blocks' state and the staged inputs share a workgroup's 64 KB)
    if (dyn) {
        if (qa) hipLaunchKernelGGL((banded_fill_blocks_kernel<B, true, true>), grid, block, dyn, stream, p, L.begin);
        else    hipLaunchKernelGGL((banded_fill_blocks_kernel<B, false, true>), grid, block, dyn, stream, p, L.begin);
    } else {
        if (qa) hipLaunchKernelGGL((banded_fill_blocks_kernel<B, true, false>), grid, block, 0, stream, p, L.begin);
        else    hipLaunchKernelGGL((banded_fill_blocks_kernel<B, false, false>), grid, block, 0, stream, p, L.begin);
    }
}

template <int R>
static void launch_banded_fill(const BandedParams& p, const BandedLaunch& L, hipStream_t stream) {
    const dim3 grid(L.count), block(64);
    const bool qa = p.quals != nullptr;
    if (L.lds_bytes) {
        if (qa) hipLaunchKernelGGL((banded_fill_kernel<R, true, true>), grid, block, L.lds_bytes, stream, p, L.begin);
        else    hipLaunchKernelGGL((banded_fill_kernel<R, false, true>), grid, block, L.lds_bytes, stream, p, L.begin);
    } else {
        if (qa) hipLaunchKernelGGL((banded_fill_kernel<R, true, false>), grid, block, 0, stream, p, L.begin);
        else    hipLaunchKernelGGL((banded_fill_kernel<R, false, false>), grid, block, 0, stream, p, L.begin);
    }
}

__global__ void __launch_bounds__(64) banded_walk_kernel(const BandedParams P) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < P.n) banded_walk_one(P, i);
}

// ---- gapless extension (gapless_device.hpp): resident threads, one scratch slab per thread
// Three kernels.  gapless_search_kernel runs every read's searches as one flat loop per lane, the queue of a search in LDS
// (lane-interleaved dwords: no bank conflicts, no HBM traffic; 35 dwords per thread) and only the write-once path links in the
// thread's slab; gapless_rules_kernel applies the set rules to the winners, one read per lane; a search whose queue outgrows the LDS
// slots marks its read G_RETRY and the slab kernel below (nested loops, the whole search in the thread's 11 KB HBM slab, round 1's
// kernel) runs exactly those reads again.
template <bool MG> __global__ void __launch_bounds__(64, 4) gapless_kernel(const GaplessParams P, const uint32_t threads, const int retry_only) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= threads) return;
    GStoreSlab Q{P.scratch[t]};
    for (uint32_t k = t; k < P.n; k += threads) {
        const uint32_t i = P.order[k];
        if (retry_only && !(P.retry ? P.retry[i] != 0 : P.results[i].status == G_RETRY)) continue;
        gapless_extend_one<MG>(P, i, Q, P.scratch[t], P.cold[t]);
    }
}
// The flat form (gapless_device.hpp, "the flat form"): lanes take reads from a counter and never wait for another lane's search; the
// rules over the winners run in their own kernel.  Reads whose queue outgrew the LDS slots go to the slab kernel as before.
struct GWaveDev {
    GProf* p; uint32_t min_idle;
    __device__ GProf* prof() const { return p; }
    __device__ int vote(bool idle, bool searching) const {
        const unsigned long long bi = __ballot(idle), bs = __ballot(searching);
        if (!(bi | bs)) return 0;
        return ((uint32_t)__popcll(bi) >= min_idle || !bs) ? 1 : 2;
    }
    __device__ uint32_t next_read(const GaplessParams& P) const {
        const unsigned long long k = atomicAdd(P.counters + 4, 1ull);
        return k < (unsigned long long)P.n ? (uint32_t)k : 0xffffffffu;
    }
};
#ifndef VGK_GAPLESS_OCC
#define VGK_GAPLESS_OCC 4          // wavefronts per SIMD the search kernel is built for (tools/build_variant.sh: 5 and 6 measured in round 6, profiles/r06/NOTES.md §5)
#endif
template <bool MG> __global__ void __launch_bounds__(64, VGK_GAPLESS_OCC) gapless_search_kernel(const GaplessParams P, const uint32_t threads) {
    __shared__ uint32_t lds[64 * G_FAST_DW];
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= threads) return;
    GStoreLds Q{lds + threadIdx.x, 64u, P.scratch[t], 0u};
#if defined(VGAMD_GAPLESS_PROF)
    GProf prof; prof.start();
    GWaveDev wave{&prof, P.flat_min_idle};
    gapless_search_lane<MG>(P, Q, P.scratch[t], wave);
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 12; ++i) atomicAdd(P.counters + 8 + i, prof.acc[i]);
#else
    GWaveDev wave{nullptr, P.flat_min_idle};
    gapless_search_lane<MG>(P, Q, P.scratch[t], wave);
#endif
}
template <bool MG> __global__ void __launch_bounds__(64, 6) gapless_rules_kernel(const GaplessParams P) {
    __shared__ uint8_t order[64 * G_SEEDS];           // the permutation the rules sort, per lane
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= P.n) return;
    gapless_rules_one<MG>(P, P.order[k], order + threadIdx.x * G_SEEDS);
}

__global__ void __launch_bounds__(256) gapless_order_kernel(const GOrderParams P, const int stage) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    if (stage == 1) g_order_sizes_one(P, i); else g_order_gather_one(P, i);
}
__global__ void __launch_bounds__(256) gapless_seeded_kernel(const GSeededParams P) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < P.n) g_seeded_one(P, i);
}
// hipMemsetAsync's fill kernel was measured at ~30 GB/s for the stage's 22 MB table (0.7 ms per step): large clears go through this one
__global__ void __launch_bounds__(256) zero_kernel(uint4* dst, const size_t vecs, unsigned char* tail, const uint32_t tail_bytes) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < vecs; i += stride) dst[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < tail_bytes) tail[threadIdx.x] = 0;
}
__global__ void __launch_bounds__(256) mask_reads_kernel(char* reads, const size_t bytes) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= bytes) return;
    if (i + 16 <= bytes) {
        uint4 w = *reinterpret_cast<const uint4*>(reads + i);
        char* c = reinterpret_cast<char*>(&w);
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = g_mask_base(c[k]);
        *reinterpret_cast<uint4*>(reads + i) = w;
    } else for (size_t k = i; k < bytes; ++k) reads[k] = g_mask_base(reads[k]);
}

// ---- minimizer seeding (minimizer_device.hpp): one wavefront per read, one lane per k-mer position
// Rounds of 64 k-mer positions, 65 - w of them new (a round's first w - 1 lanes only complete the windows that end behind them).
// Per round every lane builds the forward and reverse keys of ITS k-mer from unaligned 8-byte loads, hashes both and keeps the
// canonical one; whether a k-mer is a candidate is k consecutive set bits in the ballots of "this base is ACGT"; the minimum of the w
// candidates ending at a lane — (hash, position) lexicographic = the leftmost smallest — comes from doubling steps over lane
// shuffles plus one combining step; a window reports its minimum when that lies behind everything reported before (window minima
// move right only): a prefix maximum.  The round's reported minimizers (at most 65 - w) are compacted into LDS by ballot + popcount,
// one lane each probes the table, and their hits become seeds in order, a (node, diagonal) pair the read already has being dropped by
// one ballot over the seeds kept so far.  Results are those of minimizer_one, which the emulator and the index builder run.
struct MzMin { uint64_t key; uint32_t hash_lo; uint32_t pos_rev; };        // pos_rev = read offset << 1 | reverse
// Four reads per workgroup, one wavefront each (no workgroup barrier anywhere: a wavefront's LDS operations complete in order, the
// fence only keeps the compiler from moving them).  The seeds go to slots of MZ_MAX_SEEDS per read; minimizer_gather_kernel packs
// them behind each other once the prefix sums of the counts are known — one pass over the reads instead of a counting and a writing one.
#define MZ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
__global__ void __launch_bounds__(256) minimizer_kernel(const MinimizerParams P, vgk_seed* slots) {
    __shared__ MzMin mins_all[4][64];
    __shared__ unsigned long long seen_all[4][MZ_MAX_SEEDS];
    __shared__ unsigned long long cand_all[4][64];                           // a round's candidate seeds, in hit order
    __shared__ uint8_t own_all[4][64];                                       // the minimizer (lane) each of them belongs to
    // find_seeds' choice (MinimizerParams::policy): the read's minimizers in read order — key, hits, the hits of the key's run, score — and
    // the order the filters take them in
    __shared__ unsigned long long pkey_all[4][MZ_POLICY_MAX];
    __shared__ double psc_all[4][MZ_POLICY_MAX];
    __shared__ uint32_t phit_all[4][MZ_POLICY_MAX], prun_all[4][MZ_POLICY_MAX];
    __shared__ uint8_t pord_all[4][MZ_POLICY_MAX];
    __shared__ uint8_t ptie_all[4][3 * (MZ_POLICY_MAX + 1)];                  // the top tie's runs, their permutation and the new order (mz_shuffle_top_ties)
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t i = P.lo + blockIdx.x * 4u + wv;
    if (i >= P.hi) return;
    MzMin* mins = mins_all[wv]; unsigned long long* seen = seen_all[wv]; unsigned long long* cand = cand_all[wv]; uint8_t* own = own_all[wv];
    const uint64_t a = P.read_off[i]; const uint32_t L = (uint32_t)(P.read_off[i + 1] - a);
    const uint32_t k = P.index.k, w = P.index.w;
    const char* rd = P.reads + a;
    vgk_seed* dst = slots + (size_t)i * MZ_MAX_SEEDS;
    uint32_t n_min = 0, n_seeds = 0; bool truncated = false;
    unsigned long long* pkey = pkey_all[wv]; double* psc = psc_all[wv]; uint32_t* phit = phit_all[wv]; uint32_t* prun = prun_all[wv]; uint8_t* pord = pord_all[wv];
    unsigned long long chosen = ~0ull; bool skipped = false;
    // With a policy the rounds run twice: first only to list the read's minimizers and their hit counts (the choice needs all of them
    // before the first seed is written), then — the choice made, across the lanes, between the two — as without one, the minimizers that were
    // not chosen giving no seeds.
    if (L >= k + w - 1) for (int phase = P.policy.on ? 0 : 1; phase < 2; ++phase) {
        if (phase == 1 && P.policy.on) {
            const uint32_t np = n_min;
            if (np > MZ_POLICY_MAX) skipped = true;
            else {
                const MzPolicy& Q = P.policy;
                const unsigned long long my_key = lane < np ? pkey[lane] : 0ull; const uint32_t my_hits = lane < np ? phit[lane] : 0u;
                uint32_t run = 0;
                for (uint32_t j = 0; j < np; ++j) run += pkey[j] == my_key ? phit[j] : 0u;
                const double my_sc = mz_score(Q, my_hits);
                if (lane < np) { psc[lane] = my_sc; prun[lane] = run; }
                MZ_WAVE_SYNC();
                uint32_t rank = 0;
                for (uint32_t j = 0; j < np; ++j) rank += mz_better(psc[j], pkey[j], j, my_sc, my_key, lane) ? 1u : 0u;
                if (lane < np) pord[rank] = (uint8_t)lane;
                MZ_WAVE_SYNC();
                // the runs tied at the top, shuffled as the reference shuffles them (minimizer_device.hpp) — when their order can change the choice at
                // all: tied runs with at most hit_cap hits are all taken whatever their order (nearly every read: its unique minimizers tie at the
                // top), so the generator is only made for a read whose BEST minimizers are repetitive.  The seed is folded across the lanes
                // (seed = sum of byte_i x 13^(L - 1 - i) in 32 bits: a lane's bytes times their powers, summed over the wavefront).
                { // (the tie itself across the lanes: lane r looks at the r-th minimizer of the order — tied with the first one? the first of its run?)
                  const uint32_t xr = lane < np ? pord[lane] : 0u, x0 = np ? pord[0] : 0u;
                  const bool tied = lane < np && psc[xr] == psc[x0];
                  const bool opens = tied && (lane == 0 || pkey[pord[lane - 1]] != pkey[xr]);
                  const uint32_t t_elems = (uint32_t)__popcll(__ballot(tied)), t_runs = (uint32_t)__popcll(__ballot(opens)), t_hits = np ? phit[x0] : 0u;
                  if (mz_tie_matters(Q, t_runs, t_hits)) {
                      uint32_t part = 0; bool other = false;
                      for (uint32_t b = lane; b < L; b += 64u) {
                          const char c = rd[b]; other = other || !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                          uint32_t pw = 1u, base = 13u;
                          for (uint32_t e = L - 1u - b; e; e >>= 1) { if (e & 1u) pw *= base; base *= base; }
                          part += (uint32_t)(uint8_t)c * pw;
                      }
#pragma unroll
                      for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
                      if (__ballot(other) != 0ull || Q.paired) skipped = true;          // a masked base — or a mate of a pair, whose generator is the pair's (include/vgk.h): the reference's seed cannot be made here — not chosen for
                      else if (lane == 0) { uint8_t* t = ptie_all[wv]; mz_shuffle_top_ties(pord, pkey, t_elems, t_runs, part, t, t + (MZ_POLICY_MAX + 1), t + 2 * (MZ_POLICY_MAX + 1)); }
                  }
                  MZ_WAVE_SYNC(); }
                // the filters, in that order: every lane the same few dozen steps (the sums must be made in this order)
                const bool use_score = Q.hit_cap != 0 || Q.fraction != 1.0;
                double target = 0.0, selected = 0.0;
                if (use_score) { double base = 0.0; for (uint32_t r = 0; r < np; ++r) base = mz_add(base, psc[pord[r]]); target = mz_add(mz_mul(base, Q.fraction), 0.000001); }
                unsigned long long mask = 0ull; bool taking = false; unsigned long long prev = 0ull;
                for (uint32_t r = 0; r < np; ++r) {
                    const uint32_t x = pord[r]; const unsigned long long kx = pkey[x]; const uint32_t hx = phit[x]; const double sx = psc[x];
                    if (r == 0 || kx != prev) taking = false;
                    prev = kx;
                    bool pass = hx != 0u && prun[x] <= Q.hard_hit_cap;
                    if (pass && use_score) {
                        if (hx <= Q.hit_cap || mz_add(selected, sx) <= target || taking) selected = mz_add(selected, sx);
                        else { pass = false; target = selected; }
                    }
                    if (pass) { mask |= 1ull << x; taking = true; }
                }
                chosen = skipped ? ~0ull : mask;                             // (a read that is not chosen for is seeded as without a policy)
            }
            n_min = 0;
        }
        const uint32_t cap_now = P.policy.on && !skipped ? 0xffffffffu : P.hit_cap;      // (the choice held the run's hits against the hard cap)
        const uint32_t n_kmers = L - k + 1, step = 65 - w;
        const uint64_t kmask = (1ull << (2 * k)) - 1ull;                   // k <= 31
        const unsigned long long need = (1ull << k) - 1ull;
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t last_plus1 = 0;                                            // 1 + the last reported position (0: none yet)
        for (uint32_t base = 0; base + w - 1 < n_kmers; base += step) {
            const uint32_t q = base + lane;                                 // this lane's k-mer position = the end of its window
            // The 2-bit codes of bases base .. base + 127 as two bit planes held in four ballots; a lane's k-mer is k consecutive positions
            // from bit `lane` on: funnel-shifted out of the planes, interleaved once into the reverse-complement key's layout (base t at
            // bits 2t), from which the forward key is the same pairs in reverse order.  No per-base loop, no byte loads beyond two per lane.
            const char c0 = q < L ? rd[q] : 'N', c1 = q + 64 < L ? rd[q + 64] : 'N';
            const int x0 = mz_code(c0), x1 = mz_code(c1);
            const unsigned long long v0 = __ballot(x0 >= 0), v1 = __ballot(x1 >= 0);
            const unsigned long long lo0 = __ballot(x0 & 1), lo1 = __ballot(x1 & 1), hi0 = __ballot((x0 >> 1) & 1), hi1 = __ballot((x1 >> 1) & 1);      // (x < 0: both bits set, and the k-mer is no candidate)
            auto from_lane = [&](unsigned long long a0, unsigned long long a1) { return lane ? ((a0 >> lane) | (a1 << (64 - lane))) : a0; };
            const unsigned long long vbits = from_lane(v0, v1);
            const bool valid = q < n_kmers && (vbits & need) == need;
            const uint32_t b_lo = (uint32_t)(from_lane(lo0, lo1) & need), b_hi = (uint32_t)(from_lane(hi0, hi1) & need);
            auto spread = [](uint32_t v) {                                  // bit t -> bit 2t
                uint64_t z = v;
                z = (z | (z << 16)) & 0x0000ffff0000ffffull; z = (z | (z << 8)) & 0x00ff00ff00ff00ffull; z = (z | (z << 4)) & 0x0f0f0f0f0f0f0f0full;
                z = (z | (z << 2)) & 0x3333333333333333ull; z = (z | (z << 1)) & 0x5555555555555555ull;
                return z;
            };
            const uint64_t pairs = spread(b_lo) | (spread(b_hi) << 1);       // base t's code at bits 2t, 2t + 1
            uint64_t rev = ~pairs & kmask;                                   // complement of every base, the k-mer's last base in the highest pair
            uint64_t br = __builtin_bitreverse64(pairs);                     // pairs in reverse order, the two bits of a pair swapped
            br = ((br >> 1) & 0x5555555555555555ull) | ((br & 0x5555555555555555ull) << 1);
            uint64_t fwd = (br >> (64 - 2 * k)) & kmask;                      // base 0 in the highest pair
            if (!valid) { fwd = 0; rev = 0; }
            const uint64_t hf = mz_hash(fwd), hr = mz_hash(rev);
            const bool reverse = hr < hf;
            const uint64_t my_hash = reverse ? hr : hf, my_key = reverse ? rev : fwd;
            uint64_t h = valid ? my_hash : ~0ull;                           // ~0: no candidate
            uint32_t hp = q;
            uint32_t span = 1;
            while (2 * span <= w) {
                const uint64_t oh = __shfl_up(h, span, 64); const uint32_t op = __shfl_up(hp, span, 64);
                if (lane >= span && (oh < h || (oh == h && op < hp))) { h = oh; hp = op; }
                span *= 2;
            }
            if (span < w) {
                const uint32_t d = w - span;
                const uint64_t oh = __shfl_up(h, d, 64); const uint32_t op = __shfl_up(hp, d, 64);
                if (lane >= d && (oh < h || (oh == h && op < hp))) { h = oh; hp = op; }
            }
            const bool ends = lane >= w - 1 && q < n_kmers && h != ~0ull;   // a complete window with a candidate
            uint32_t run = ends ? hp + 1 : 0u;                              // prefix maximum of the minima so far in this round
            for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(run, d, 64); if (lane >= d && o > run) run = o; }
            uint32_t before = __shfl_up(run, 1, 64); if (lane == 0) before = 0;
            if (last_plus1 > before) before = last_plus1;
            const bool report = ends && hp + 1 > before;
            const uint32_t owner = hp - base;                               // the lane that built the minimum's k-mer (inside this round for every complete window)
            const uint64_t okey = __shfl(my_key, owner & 63u, 64), ohash = __shfl(my_hash, owner & 63u, 64);
            const uint32_t orev = __shfl((uint32_t)reverse, owner & 63u, 64);
            const unsigned long long rb = __ballot(report);
            const uint32_t n_round = (uint32_t)__popcll(rb);
            if (report) { MzMin& mm = mins[__popcll(rb & below)]; mm.key = okey; mm.hash_lo = (uint32_t)ohash; mm.pos_rev = (hp << 1) | orev; }
            const uint32_t round_max = __shfl(run, 63, 64);
            if (round_max > last_plus1) last_plus1 = round_max;
            const uint32_t n_before = n_min;
            n_min += n_round;
            MZ_WAVE_SYNC();
            if (phase == 0) {                                                // list them; nothing else
                if (lane < n_round && n_before + lane < MZ_POLICY_MAX) {
                    const MzMin mm = mins[lane];
                    MzKmer km; km.key = mm.key; km.hash = mm.hash_lo; km.reverse = (mm.pos_rev & 1u) != 0;
                    uint32_t f0 = 0, c0 = 0; MzPos o0{0u, 0u};
                    pkey[n_before + lane] = mm.key; phit[n_before + lane] = mz_find(P.index, km, f0, c0, o0) ? c0 : 0u;
                }
                MZ_WAVE_SYNC();
                continue;
            }
            // this round's minimizers: one lane each probes the table (a key's single position comes with its slot); then ALL their hits at
            // once — hit c of the round in lane c: an exclusive prefix sum of the counts, the owners spread through LDS, one load for the
            // positions that are not in their slots, every candidate held against the seeds kept so far and the candidates before it
            // (uniform LDS reads), the survivors ranked by ballot.  The result is the serial order's (minimizer_one): a candidate is looked
            // at iff the read is not full when its turn comes, the first of equal (node, diagonal) pairs is kept.  A round with more than
            // 64 hits takes the serial form below.
            uint32_t first = 0, count = 0, p = 0, rv = 0; MzPos one{0u, 0u};
            if (lane < n_round) {                                            // (also for a read that is full already: whether hits are left decides its truncated flag)
                const MzMin mm = mins[lane];
                MzKmer km; km.key = mm.key; km.hash = mm.hash_lo; km.reverse = (mm.pos_rev & 1u) != 0;
                p = mm.pos_rev >> 1; rv = mm.pos_rev & 1u;
                if (!mz_find(P.index, km, first, count, one) || count > cap_now) count = 0;
                if (n_before + lane < 64u && !((chosen >> (n_before + lane)) & 1ull)) count = 0;
            }
            const unsigned long long hb = __ballot(count != 0);
            const uint32_t cc = count > 65u ? 65u : count;                   // (clamped: the sum must not wrap; a round within 64 hits has no clamped count)
            uint32_t incl = cc;
            for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            const uint32_t total = __shfl(incl, 63, 64), excl = incl - cc;
            if (hb && total <= 64u) {
                for (uint32_t h0 = 0; h0 < count; ++h0) own[excl + h0] = (uint8_t)lane;
                MZ_WAVE_SYNC();
                const uint32_t j = lane < total ? own[lane] : 0u;
                const uint32_t fj = __shfl(first, j, 64), ej = __shfl(excl, j, 64), pj = __shfl(p, j, 64), rj = __shfl(rv, j, 64);
                MzPos qp; qp.node = __shfl(one.node, j, 64); qp.offset = __shfl(one.offset, j, 64);
                if (lane < total && fj != MZ_ONE) qp = P.index.pos[fj + (lane - ej)];
                const vgk_seed sd = mz_seed(qp, pj, rj != 0u, k);
                const unsigned long long key = ((unsigned long long)sd.node << 32) | (uint32_t)sd.diff;
                cand[lane] = key;
                MZ_WAVE_SYNC();
                bool dup = false;
                for (uint32_t x = 0; x < n_seeds; ++x) dup |= seen[x] == key;
                for (uint32_t x = 0; x + 1u < total; ++x) dup |= x < lane && cand[x] == key;
                const bool keep = lane < total && !dup;
                const unsigned long long kb = __ballot(keep);
                const uint32_t rank = n_seeds + (uint32_t)__popcll(kb & below);      // seeds kept when this candidate's turn comes
                if (keep && rank < MZ_MAX_SEEDS) { seen[rank] = key; dst[rank] = sd; }
                if (__ballot(lane < total && rank >= MZ_MAX_SEEDS) != 0ull) truncated = true;      // the cap: that hit and the rest are never looked at
                n_seeds += (uint32_t)__popcll(kb); if (n_seeds > MZ_MAX_SEEDS) n_seeds = MZ_MAX_SEEDS;
            } else
            for (unsigned long long todo = hb; todo; todo &= todo - 1) {
                if (n_seeds >= MZ_MAX_SEEDS) { truncated = true; break; }    // the cap: these hits are never looked at
                const uint32_t j = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
                const uint32_t cj = __shfl(count, j, 64), fj = __shfl(first, j, 64), pj = __shfl(p, j, 64), rj = __shfl(rv, j, 64);
                const uint32_t onode = __shfl(one.node, j, 64), ooff = __shfl(one.offset, j, 64);
                for (uint32_t h0 = 0; h0 < cj; h0 += 64) {
                    if (n_seeds >= MZ_MAX_SEEDS) { truncated = true; break; }
                    unsigned long long key = 0; vgk_seed sd; sd.node = 0; sd.diff = 0;
                    if (h0 + lane < cj) {
                        MzPos qp; qp.node = onode; qp.offset = ooff;
                        if (fj != MZ_ONE) qp = P.index.pos[fj + h0 + lane];
                        sd = mz_seed(qp, pj, rj != 0u, k);
                        key = ((unsigned long long)sd.node << 32) | (uint32_t)sd.diff;
                    }
                    const uint32_t in_group = cj - h0 < 64 ? cj - h0 : 64;
                    for (uint32_t x = 0; x < in_group; ++x) {                // in hit order; all lanes agree on every decision
                        if (n_seeds >= MZ_MAX_SEEDS) { truncated = true; break; }
                        const unsigned long long kx = __shfl(key, x, 64);
                        const bool dup = __ballot(lane < n_seeds && seen[lane] == kx) != 0ull;
                        if (!dup) {
                            if (lane == x) { seen[n_seeds] = key; dst[n_seeds] = sd; }
                            ++n_seeds;
                            MZ_WAVE_SYNC();
                        }
                    }
                }
            }
            MZ_WAVE_SYNC();
        }
    }
    if (lane == 0) { P.counts[i] = n_seeds; if (P.mins) P.mins[i] = n_min | (truncated ? VGK_MINIMIZERS_TRUNCATED : 0u) | (skipped ? VGK_MINIMIZERS_POLICY_SKIPPED : 0u); }
}
__global__ void __launch_bounds__(256) minimizer_gather_kernel(const MinimizerParams P, const vgk_seed* slots) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= P.n) return;
    const uint32_t at = P.first[i], cnt = P.first[i + 1] - at;
    if (lane < cnt) P.seeds[at + lane] = slots[(size_t)i * MZ_MAX_SEEDS + lane];
}

// ---- tail forests (tail_device.hpp): resident lanes take the tails in turn for the walks; one lane per tree node for the graph tables
__global__ void __launch_bounds__(64) tail_walk_kernel(const TailParams P, const uint32_t threads) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= threads) return;
    for (uint32_t i = t; i < P.n; i += threads) tail_walk_one(P, i, P.scratch[t]);
}
__global__ void __launch_bounds__(256) tail_stage_kernel(const TStageParams P, const int what, const uint32_t items) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < items) tstage_one(P, what, i);
}
__global__ void __launch_bounds__(256) rescue_requests_kernel(const RqParams P, const int what) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (what == RQ_FLAG) { if (p <= P.n_pairs) rq_flag_one(P, p); }
    else if (p < P.n_pairs) rq_emit_one(P, p);
}
// ---- seeding reads of any length (minimizer_device.hpp): a lane per read lists its minimizers, a lane per minimizer writes its seeds
__global__ void __launch_bounds__(64) minimizer_list_kernel(const MzListParams P) { mz_list_one(P, blockIdx.x * 64 + threadIdx.x); }
__global__ void __launch_bounds__(256) minimizer_seeds_of_kernel(const MzSeedsOfParams P) { mz_seeds_of_one(P, blockIdx.x * 256 + threadIdx.x); }
// ---- one Path per read (chain_device.hpp): a lane per read for the bounds and the composition, a wavefront per read for the dense copy
__global__ void __launch_bounds__(64) chain_stitch_kernel(const CsParams P, const int what) {
    cs_one(P, what, blockIdx.x * 64 + threadIdx.x);
}
__global__ void __launch_bounds__(256) chain_gather_kernel(const CsParams P) {
    const uint32_t r = blockIdx.x * 4 + threadIdx.x / 64;
    if (r < P.n_reads) cs_gather_one(P, r, threadIdx.x & 63u, 64);
}
__global__ void __launch_bounds__(256) forest_flags_kernel(const ForestParams P) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v < P.n_nodes) forest_flags_one(P, v);
}
__global__ void __launch_bounds__(256) forest_emit_kernel(const ForestParams P) {
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    if (v < P.n_nodes) forest_emit_one(P, v);
}

// ---- the WFA problems' sequences masked / flipped / laid out on the device: a wavefront per problem, four to a block
__global__ void __launch_bounds__(256) wfa_mask_kernel(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, const uint32_t n) {
    const uint32_t i = blockIdx.x * 4 + threadIdx.x / 64;
    if (i < n) wfa_mask_one(probs, src_off, raw, seqs, i, threadIdx.x & 63u, 64);
}
// ---- wavefront alignment (wfa_device.hpp): the same launch shape
__global__ void __launch_bounds__(64, 4) wfa_kernel(const WfaParams P, const uint32_t threads) {
    __shared__ uint32_t node_end[W_NODES * 64];                    // [trie node][lane]: conflict-free, 8 KB per wavefront
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t < threads) wfa_thread(P, t, node_end + threadIdx.x, 64);
    if (P.producers_done) {                                        // run beside the wavefront kernel: this wavefront will hand nothing over any more
        __threadfence();                                           // (its list entries before the count the consumers trust)
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) atomicAdd(P.producers_done, 1u);
    }
}

// ---- wavefront alignment, one wavefront per problem (wfa_wave_device.hpp): trie nodes, possible penalties and the lanes' work lists in
//      LDS, the wavefront table / its log / the path pool in the wavefront's slab (workgroup-scope atomics: one wavefront = one workgroup)
struct WwXlHip {
    __device__ __forceinline__ unsigned long long ballot(bool flag) const { return __ballot(flag); }
    __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src) const { return (uint32_t)__shfl((int)v, (int)src, 64); }
    __device__ __forceinline__ unsigned long long reduce_min_u64(unsigned long long v) const {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
        return v;
    }
    __device__ __forceinline__ int32_t reduce_max(int32_t v) const {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const int32_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
        return v;
    }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    // the same for LDS alone: a wavefront's LDS instructions execute in order, so its lanes see each other's LDS writes once the compiler
    // keeps the order — no wait for the global stores in flight (a full fence after a step's table stores waits ~1.5 us for them)
    __device__ __forceinline__ void fence_lds() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ unsigned long long load64(const unsigned long long* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ void store64(unsigned long long* p, unsigned long long v) const { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ unsigned long long cas64(unsigned long long* p, unsigned long long expect, unsigned long long desired) const {
        __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return expect;                                                         // the value found there
    }
    __device__ __forceinline__ uint32_t add32(uint32_t* p, uint32_t v) const { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ uint32_t clock_us() const { return (uint32_t)(wall_clock64() / 100ull); }      // (the constant 100 MHz counter; statistics only)
    __device__ __forceinline__ void or32(uint32_t* p, uint32_t v) const { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ uint32_t load32(const uint32_t* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
__global__ void __launch_bounds__(64, VGK_WW_OCC) wfa_wave_kernel(const WwParams P) {
    __shared__ WwSharedBoth sh;
    WwXlHip xl;
    wfa_wave(P, blockIdx.x, threadIdx.x, sh, xl);
}

// ---- pinned gssw fill with full matrices (gssw_matrix_device.hpp): one wavefront per problem, R read rows per lane; the
//      one-thread-per-problem form remains for scorings with gap_open < gap_extend
// the k-best tracebacks of vgk_gssw_align_multi over the kept matrices: one lane per problem (gssw_multi_device.hpp)
__global__ void __launch_bounds__(64) banded_multi_kernel(const BandedMultiParams Q) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < Q.P.n) banded_multi_one(Q, a);
}
__global__ void __launch_bounds__(64) gssw_multi_kernel(const GsswMultiParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.M.n) gssw_multi_one(P, i);
}
__global__ void __launch_bounds__(64) gssw_matrix_kernel(const GsswMatrixParams P) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < P.n) gssw_matrix_one(P, i);
}
// The same primitives over ONE ROW of 16 lanes (a DPP row): four problems share a wavefront, each in a row of its own.  Rows run loops
// of different lengths; a row is active or masked off as a whole, and nothing here reaches outside its row.
constexpr uint32_t XB_STAGE16 = 1024, XB_STAGE64 = 4096;          // graph bases staged in LDS per problem (xdrop_band_wave_lane)
struct XlDpp16 {
    uint8_t* stg; int32_t* cc; uint32_t cap = XB_STAGE16;
    __device__ __forceinline__ int32_t* col_cache() const { return cc; }
    __device__ __forceinline__ void lds_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ uint8_t* stage() const { return stg; }
    __device__ __forceinline__ uint32_t stage_cap() const { return cap; }
    __device__ __forceinline__ void stage_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ uint32_t width() const { return 16u; }
    __device__ __forceinline__ int32_t down(int32_t v) const { return __builtin_amdgcn_update_dpp(BNEG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false); }
    __device__ __forceinline__ int32_t scan_excl(int32_t v) const {       // exclusive max-scan over the row (XlDpp::scan_excl without the steps across rows)
        asm volatile("s_nop 4\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1" : "+v"(v));
        return down(v);
    }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    __device__ __forceinline__ unsigned long long ballot(bool flag) const { return (__ballot(flag) >> (threadIdx.x & 48u)) & 0xffffull; }
    __device__ __forceinline__ bool any(int32_t flag) const { return ballot(flag != 0) != 0ull; }
    __device__ __forceinline__ int32_t reduce_max(int32_t v) const {      // rotations inside the row (row_ror): no trip through the LDS crossbar
        int32_t o;
        o = __builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false); v = o > v ? o : v;
        o = __builtin_amdgcn_update_dpp(v, v, 0x124 /* row_ror:4 */, 0xf, 0xf, false); v = o > v ? o : v;
        o = __builtin_amdgcn_update_dpp(v, v, 0x122 /* row_ror:2 */, 0xf, 0xf, false); v = o > v ? o : v;
        o = __builtin_amdgcn_update_dpp(v, v, 0x121 /* row_ror:1 */, 0xf, 0xf, false); v = o > v ? o : v;
        return v;
    }
    __device__ __forceinline__ unsigned long long reduce_add(unsigned long long v) const {
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) v += __shfl_xor(v, d, 16);
        return v;
    }
};
struct XlDppStaged : XlDpp {
    uint8_t* stg; int32_t* cc;
    __device__ __forceinline__ uint32_t width() const { return 64u; }
    __device__ __forceinline__ int32_t* col_cache() const { return cc; }
    __device__ __forceinline__ void lds_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ uint8_t* stage() const { return stg; }
    __device__ __forceinline__ uint32_t stage_cap() const { return XB_STAGE64; }
    __device__ __forceinline__ void stage_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
};
// (CT: the planes' cell type — int32_t, or int16_t when the call's bounds allow, GsswMatrixParams::xb_cell16)
template <class CT>
__global__ void __launch_bounds__(64) xdrop_band_kernel(const GsswMatrixParams P) {
    __shared__ uint8_t stg[XB_STAGE64];
    __shared__ int32_t cc[2 * 2 * 64 * 8];                       // two last columns (H | E) of up to 512 rows
    XlDppStaged xl; xl.stg = stg; xl.cc = cc;
    xdrop_band_wave_lane_t<CT>(P, P.xb_order ? P.xb_order[P.xb_n8 + P.xb_n16 + blockIdx.x] : blockIdx.x, threadIdx.x, xl);
}
// three wavefronts per SIMD (168 VGPRs, five spilled dwords): 26.2 -> 21.0 ms per 200 000 tails against the compiler's own 175 VGPRs = two;
// four (128 VGPRs, 38 spilled): the same 20.9 ms
#ifndef VGK_XB_OCC
#define VGK_XB_OCC 3
#endif
template <class CT>
__global__ void __launch_bounds__(64, VGK_XB_OCC) xdrop_band_kernel16(const GsswMatrixParams P) {
    __shared__ uint8_t stg[4 * XB_STAGE16];
    __shared__ int32_t cc[4 * 2 * 2 * 16 * 8];                   // per problem: two last columns (H | E) of up to 128 rows
    const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 4);
    if (slot >= P.xb_n16) return;
    XlDpp16 xl; xl.stg = stg + (threadIdx.x >> 4) * XB_STAGE16; xl.cc = cc + (threadIdx.x >> 4) * (2 * 2 * 16 * 8);
    xdrop_band_wave_lane_t<CT>(P, P.xb_order[P.xb_n8 + slot], threadIdx.x & 15u, xl);
}
// the tracebacks of the X-drop band path, one lane per problem, in the fills' launch order (neighbours walk graphs of like size)
// the packed fill (xdrop_band_pk_lane): two rows to a register, half the registers, a quarter of the LDS (a word per row pair in the column
// cache, 512 staged bases per problem)
#ifndef VGK_XBP_OCC
#define VGK_XBP_OCC 4
#endif
constexpr uint32_t XBP_STAGE16 = 512;
__global__ void __launch_bounds__(64, VGK_XBP_OCC) xdrop_band_pk_kernel16(const GsswMatrixParams P) {
    __shared__ uint8_t stg[4 * XBP_STAGE16];
    __shared__ int32_t cc[4 * 2 * 2 * 16 * 4];
    const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 4);
    if (slot >= P.xb_n16) return;
    XlDpp16 xl; xl.stg = stg + (threadIdx.x >> 4) * XBP_STAGE16; xl.cc = cc + (threadIdx.x >> 4) * (2 * 2 * 16 * 4); xl.cap = XBP_STAGE16;
    xdrop_band_pk_lane(P, P.xb_order[P.xb_n8 + slot], threadIdx.x & 15u, xl);
}
// The same over HALF a DPP row: eight problems share a wavefront, 8 lanes (64 rows) each — tails of at most 63 bases, which left half of a
// 16-lane row idle.  The DPP shifts stay row-wide; what would cross from lane 7 into lane 8 is masked: bank_mask for the shift by 4 (banks are
// four lanes), a select per lane for the shifts by 1 and 2; the maximum over the group is two quad permutes and a half-row mirror.
struct XlDpp8 {
    uint8_t* stg; int32_t* cc; uint32_t cap;
    __device__ __forceinline__ int32_t* col_cache() const { return cc; }
    __device__ __forceinline__ void lds_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ uint8_t* stage() const { return stg; }
    __device__ __forceinline__ uint32_t stage_cap() const { return cap; }
    __device__ __forceinline__ void stage_sync() const { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    __device__ __forceinline__ uint32_t width() const { return 8u; }
    __device__ __forceinline__ int32_t down(int32_t v) const {
        const int32_t r = __builtin_amdgcn_update_dpp(BNEG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
        return (threadIdx.x & 7u) ? r : BNEG;
    }
    __device__ __forceinline__ int32_t scan_excl(int32_t v) const {       // exclusive max-scan over the group's 8 lanes
        const uint32_t l = threadIdx.x & 7u;
        int32_t t = __builtin_amdgcn_update_dpp(BNEG, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false); t = l >= 1u ? t : BNEG; v = t > v ? t : v;
        t = __builtin_amdgcn_update_dpp(BNEG, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false); t = l >= 2u ? t : BNEG; v = t > v ? t : v;
        t = __builtin_amdgcn_update_dpp(BNEG, v, 0x114 /* row_shr:4 */, 0xf, 0xa /* lanes 4-7 and 12-15 */, false); v = t > v ? t : v;
        return down(v);
    }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    __device__ __forceinline__ unsigned long long ballot(bool flag) const { return (__ballot(flag) >> (threadIdx.x & 56u)) & 0xffull; }
    __device__ __forceinline__ bool any(int32_t flag) const { return ballot(flag != 0) != 0ull; }
    __device__ __forceinline__ int32_t reduce_max(int32_t v) const {
        int32_t o;
        o = __builtin_amdgcn_update_dpp(v, v, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false); v = o > v ? o : v;
        o = __builtin_amdgcn_update_dpp(v, v, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false); v = o > v ? o : v;
        o = __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false); v = o > v ? o : v;
        return v;
    }
    __device__ __forceinline__ unsigned long long reduce_add(unsigned long long v) const {
#pragma unroll
        for (int d = 4; d > 0; d >>= 1) v += __shfl_xor(v, d, 8);
        return v;
    }
};
constexpr uint32_t XBP_STAGE8 = 256;
__global__ void __launch_bounds__(64, VGK_XBP_OCC) xdrop_band_pk_kernel8(const GsswMatrixParams P) {
    __shared__ uint8_t stg[8 * XBP_STAGE8];
    __shared__ int32_t cc[8 * 2 * 2 * 8 * 4];
    const uint32_t slot = blockIdx.x * 8u + (threadIdx.x >> 3);
    if (slot >= P.xb_n8) return;
    XlDpp8 xl; xl.stg = stg + (threadIdx.x >> 3) * XBP_STAGE8; xl.cc = cc + (threadIdx.x >> 3) * (2 * 2 * 8 * 4); xl.cap = XBP_STAGE8;
    xdrop_band_pk_lane(P, P.xb_order[slot], threadIdx.x & 7u, xl);
}
__global__ void __launch_bounds__(64) xdrop_band_pk_kernel(const GsswMatrixParams P) {
    __shared__ uint8_t stg[XB_STAGE64];
    __shared__ int32_t cc[2 * 2 * 64 * 4];
    XlDppStaged xl; xl.stg = stg; xl.cc = cc;
    xdrop_band_pk_lane(P, P.xb_order ? P.xb_order[P.xb_n8 + P.xb_n16 + blockIdx.x] : blockIdx.x, threadIdx.x, xl);
}
template <class CT>
__global__ void __launch_bounds__(64) xdrop_band_walk_kernel(const GsswMatrixParams P) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k < P.n) xdrop_band_walk_one_t<CT>(P, P.xb_order ? P.xb_order[k] : k);
}
template <int R>
__global__ void __launch_bounds__(64) gssw_matrix_wave_kernel(const GsswMatrixParams P, const uint32_t rows_lo, const uint32_t rows_hi) {
    const uint32_t i = blockIdx.x;
    const uint32_t L = P.probs[i].L;
    if (L <= rows_lo || L > rows_hi) return;           // another rows-per-lane class takes this problem
    XlDpp xl;
    gssw_matrix_wave_lane<R>(P, i, threadIdx.x, xl);
}

// The wide route (gssw_wide_device.hpp): one workgroup of four wavefronts per problem; its 256 lanes run ONE skewed wavefront.  Inside a
// wavefront the row above arrives by DPP; lane 0 of wavefronts 1-3 takes it from the three words the last lane of the wavefront before
// left in LDS at the end of the previous step (double-buffered by step parity); one barrier per step keeps the four in step.
template <int K>
__global__ void __launch_bounds__(256) gssw_wide_kernel(const WideParams P) {
    __shared__ int32_t xh[2][4], xf[2][4];
    __shared__ uint32_t xi[2][4];
    const uint32_t i = P.order[P.order_begin + blockIdx.x];
    const WideProb d = P.probs[i];
    const uint32_t lane = threadIdx.x, wv = lane >> 6, wl = lane & 63u;
    uint32_t* tb = (d.flags & VGK_GSSW_TRACEBACK) ? P.tb + d.tb_off : nullptr;
    for (uint32_t strip = 0; strip < d.n_strips; ++strip) {
        WLane<K> s;
        wide_lane_init<K>(s, P, d, strip, lane);
        const uint32_t rows_left = d.L - strip * WIDE_LANES * (uint32_t)K;
        const uint32_t lanes_used = rows_left >= WIDE_LANES * (uint32_t)K ? WIDE_LANES : (rows_left + (uint32_t)K - 1u) / (uint32_t)K;
        const uint32_t n_steps = d.R + lanes_used - 1u;
        if (wl == 63u) { xh[0][wv] = xh[1][wv] = s.out_h; xf[0][wv] = xf[1][wv] = s.out_f; xi[0][wv] = xi[1][wv] = s.info; }      // "no column yet" for the wavefront after this one
        __syncthreads();
        for (uint32_t t = 0; t < n_steps; ++t) {
            int32_t rh = (int32_t)from_lane_above((uint32_t)s.out_h);
            int32_t rf = (int32_t)from_lane_above((uint32_t)s.out_f);
            uint32_t ri = from_lane_above(s.info);
            if (wl == 0 && wv > 0) { const uint32_t b = (t + 1u) & 1u; rh = xh[b][wv - 1]; rf = xf[b][wv - 1]; ri = xi[b][wv - 1]; }
            uint32_t* rec = tb ? tb + (uint64_t)strip * d.strip_dwords + ((uint64_t)t * WIDE_LANES + lane) * (K / 8) : nullptr;
            wide_lane_step<K>(s, P, d, strip, lane, t, rh, rf, ri, rec);
            if (wl == 63u) { const uint32_t b = t & 1u; xh[b][wv] = s.out_h; xf[b][wv] = s.out_f; xi[b][wv] = s.info; }
            __syncthreads();
        }
        unsigned long long key;
        if (wide_lane_best<K>(s, d, lane, key)) atomicMax(&P.best[i], key);
        __threadfence();            // this strip's carry row and saved columns -> the next strip's lanes
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64) gssw_wide_walk_kernel(const WideParams P) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k < P.n) wide_walk_one(P, k);
}

class HipBackend final : public Backend {
public:
    int dev = 0; int n_launches = 1; hipStream_t stream = nullptr, copy = nullptr, fetch = nullptr, side[2] = {nullptr, nullptr}; hipEvent_t side_done[2] = {nullptr, nullptr}; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t alt = nullptr; hipEvent_t evb[3] = {nullptr, nullptr, nullptr}; float ms_fill_b = 0.f, ms_walk_b = 0.f; bool pending_b = false, timed_walk_b = false, lane1_on_main = true;   // launch lane 1
    hipEvent_t fill_done[2] = {nullptr, nullptr}; bool fill_done_set[2] = {false, false};
    const bool walk_in_fill_order = std::getenv("VGAMD_WALK_PROBLEM_ORDER") == nullptr;
    const bool fills_take_turns = std::getenv("VGAMD_FILLS_TAKE_TURNS") != nullptr;   // experiment: a lane's fill waits for the other lane's fill, so that only a traceback
                                                                                     // runs beside a fill — no faster than one stream (the traceback takes the fill's wave slots); off
    hipDeviceProp_t prop;
    float ms_fill = 0.f, ms_walk = 0.f; bool timed_walk = false, pending = false;
    hipEvent_t sev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; bool sev_set[2] = {false, false}; float ms_refill[2] = {0.f, 0.f};      // the speculative fill's second fill (layout + fill), per launch lane
    float ms_gapless = 0.f, ms_wfa = 0.f, ms_xband = 0.f;
    float ms_bfill = 0.f, ms_bwalk = 0.f; hipEvent_t bev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t xbev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // run_xdrop_band_async: start / end per slot
    hipEvent_t bbev[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};      // run_banded_async: start / fills done / walk done per slot
    void* scan_tmp = nullptr; size_t scan_tmp_bytes = 0;      // rocPRIM's scratch for scan_u32 (grow-only)
    void* mz_slots = nullptr; size_t mz_slots_bytes = 0;      // per-read seed slots of run_minimizer (grow-only)
    void* banded_slab = nullptr; size_t banded_slab_bytes = 0;      // the block states of the banded fill's HBM-state classes (grow-only)
    ~HipBackend() override {
        hipSetDevice(dev);
        for (auto& e : ev) if (e) hipEventDestroy(e);
        for (auto& e : bev) if (e) hipEventDestroy(e);
        for (auto& pair : xbev) for (auto& e : pair) if (e) hipEventDestroy(e);
        for (auto& trio : bbev) for (auto& e : trio) if (e) hipEventDestroy(e);
        for (auto& pair : sev) for (auto& e : pair) if (e) hipEventDestroy(e);
        if (scan_tmp) hipFree(scan_tmp);
        if (mz_slots) hipFree(mz_slots);
        if (banded_slab) hipFree(banded_slab);
        for (int i = 0; i < 2; ++i) { if (side[i]) hipStreamDestroy(side[i]); if (side_done[i]) hipEventDestroy(side_done[i]); }
        if (stream) hipStreamDestroy(stream);
        if (copy) hipStreamDestroy(copy);
        if (fetch) hipStreamDestroy(fetch);
        if (alt) hipStreamDestroy(alt);
        for (auto& e : evb) if (e) hipEventDestroy(e);
        for (auto& e : fill_done) if (e) hipEventDestroy(e);
    }
    const char* name() const override { return prop.name; }
    int compute_units() const override { return prop.multiProcessorCount; }
    size_t memory_bytes() const override { return prop.totalGlobalMem; }
    void* alloc(size_t bytes) override {
        hipSetDevice(dev);
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        return p;
    }
    void release(void* p) override { if (p) { hipSetDevice(dev); hipFree(p); } }
    void* host_alloc(size_t bytes) override {
        hipSetDevice(dev);
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
        return p;
    }
    void host_release(void* p) override { if (p) { hipSetDevice(dev); hipHostFree(p); } }
    int host_register(const void* p, size_t bytes) override { hipSetDevice(dev); return hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess ? VGK_OK : VGK_EINVAL; }
    int host_unregister(const void* p) override { hipSetDevice(dev); return hipHostUnregister(const_cast<void*>(p)) == hipSuccess ? VGK_OK : VGK_EINVAL; }
    int upload(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int upload_side(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, copy) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    void* event_create() override {
        hipSetDevice(dev);
        hipEvent_t e = nullptr;
        return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? (void*)e : nullptr;
    }
    void event_destroy(void* ev) override { if (ev) { hipSetDevice(dev); hipEventDestroy((hipEvent_t)ev); } }
    int event_record(void* ev) override {
        if (!ev) return VGK_OK;
        hipSetDevice(dev);
        return hipEventRecord((hipEvent_t)ev, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    bool event_done(void* ev) override {
        if (!ev) return false;
        hipSetDevice(dev);
        const hipError_t e = hipEventQuery((hipEvent_t)ev);
        if (e != hipSuccess && e != hipErrorNotReady) (void)hipGetLastError();
        return e == hipSuccess;
    }
    int event_wait(void* ev) override {
        if (!ev) return sync();
        hipSetDevice(dev);
        for (;;) {
            const hipError_t e = hipEventQuery((hipEvent_t)ev);
            if (e == hipSuccess) return VGK_OK;
            if (e != hipErrorNotReady) { (void)hipGetLastError(); return VGK_ENODEV; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    int fetch_after(void* ev) override {
        if (!ev) return VGK_OK;
        hipSetDevice(dev);
        return hipStreamWaitEvent(fetch, (hipEvent_t)ev, 0) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int sync_fetch() override {
        hipSetDevice(dev);
        return hipStreamSynchronize(fetch) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int download_fetch(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, fetch) != hipSuccess) return VGK_ENODEV;
        return hipStreamSynchronize(fetch) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int download_fetch_async(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, fetch) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    hipEvent_t side_mark = nullptr;
    int main_after_side() override {
        hipSetDevice(dev);
        if (!side_mark && hipEventCreateWithFlags(&side_mark, hipEventDisableTiming) != hipSuccess) return VGK_ENODEV;
        if (hipEventRecord(side_mark, copy) != hipSuccess || hipStreamWaitEvent(stream, side_mark, 0) != hipSuccess) return VGK_ENODEV;
        return VGK_OK;
    }
    int sync_side() override {
        hipSetDevice(dev);
        return hipStreamSynchronize(copy) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int device_index() const override { return dev; }
    int download(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return VGK_ENODEV;
        return hipStreamSynchronize(stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int zero(void* dst, size_t bytes) override {
        hipSetDevice(dev);
        if (bytes >= ((size_t)1 << 18) && ((uintptr_t)dst & 15u) == 0) {
            const size_t vecs = bytes / 16;
            const unsigned blocks = (unsigned)std::min<size_t>((vecs + 255) / 256, (size_t)std::max(1, prop.multiProcessorCount) * 8);
            hipLaunchKernelGGL(zero_kernel, dim3(blocks), dim3(256), 0, stream, (uint4*)dst, vecs, (unsigned char*)dst + vecs * 16, (uint32_t)(bytes - vecs * 16));
            return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
        }
        return hipMemsetAsync(dst, 0, bytes, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int sync() override {
        hipSetDevice(dev);
        return hipStreamSynchronize(stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    size_t win_tmp_bytes(uint32_t n, uint32_t n_waves_cap) override { hipSetDevice(dev); return hip_win_tmp_bytes(n, n_waves_cap); }
    int win_stage1(const WinParams& P, void* tmp, size_t tmp_bytes) override { hipSetDevice(dev); return hip_win_stage1(P, tmp, tmp_bytes, copy); }
    int win_stage2(const WinParams& P, void* tmp, size_t tmp_bytes) override { hipSetDevice(dev); return hip_win_stage2(P, tmp, tmp_bytes, copy); }
    int download_side(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, copy) != hipSuccess) return VGK_ENODEV;
        return hipStreamSynchronize(copy) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int fill_side(void* dst, int byte, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemsetAsync(dst, byte, bytes, copy) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    template <int K> int launch_fill_k(const GsswParams& p, hipStream_t stream) {
        const dim3 grid((p.wave_count + 3) / 4), block(256);
        const bool s8 = p.scale == 8, re = p.tb_mode == TB_REWALK;
        if (p.spec_fill == 1) {                                      // the first fill of a speculative batch: no codes
            if (s8 && p.key3) hipLaunchKernelGGL((gssw_fill_kernel<K, true, false, false, true>), grid, block, 0, stream, p);
            else if (s8) hipLaunchKernelGGL((gssw_fill_kernel<K, true, false, false>), grid, block, 0, stream, p); else hipLaunchKernelGGL((gssw_fill_kernel<K, false, false, false>), grid, block, 0, stream, p);
            return VGK_OK;
        }
        if (p.spec_fill == 2 && s8 && !re && !p.fused) { hipLaunchKernelGGL((gssw_fill_kernel<K, true, false, true, false, true>), grid, block, 0, stream, p); return VGK_OK; }      // (codes only: the end cells are the first fill's)
        if (re) { if (s8) hipLaunchKernelGGL((gssw_fill_kernel<K, true, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((gssw_fill_kernel<K, false, true>), grid, block, 0, stream, p); }
        else    { if (s8) hipLaunchKernelGGL((gssw_fill_kernel<K, true, false>), grid, block, 0, stream, p); else hipLaunchKernelGGL((gssw_fill_kernel<K, false, false>), grid, block, 0, stream, p); }
        return VGK_OK;
    }
    int launch_fill(const GsswParams& p, hipStream_t stream) {
        switch (p.K) {
            case 16: return launch_fill_k<16>(p, stream);
            case 19: return launch_fill_k<19>(p, stream);
            case 20: return launch_fill_k<20>(p, stream);
            case 24: return launch_fill_k<24>(p, stream);
            default: return VGK_EINVAL;
        }
    }
    // the recomputing tracebacks of one fill launch's reads (p.K / wave_begin / wave_count as for the fill); the grid covers every pair
    // of the batch (the launch's own count is a device-side fact for batches packed there): blocks beyond its reads leave at once
    template <int K> void launch_rewalk_k(const GsswParams& p, hipStream_t stream) {
        const dim3 grid(1024), block(64);                   // (the reads that left their band: a list in HBM whose length only the device knows; each lane takes every 65536th)
        if (p.scale == 8) hipLaunchKernelGGL((gssw_rewalk_kernel<K, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((gssw_rewalk_kernel<K, false>), grid, block, 0, stream, p);
    }
    template <int K> void launch_band_k(const GsswParams& p, hipStream_t stream) {
        const dim3 grid((p.wave_count + 3) / 4), block(256);
        if (p.scale == 8) hipLaunchKernelGGL((gssw_band_kernel<K, true>), grid, block, 0, stream, p); else hipLaunchKernelGGL((gssw_band_kernel<K, false>), grid, block, 0, stream, p);
    }
    void launch_walk(const GsswParams& p0, const FillLaunch* launches, uint32_t n, hipStream_t stream, int lane = 0) {
        sev_set[lane] = false;
        if (p0.tb_mode != TB_REWALK) {
            if (p0.walk_passes == 2) {
                hipLaunchKernelGGL(gssw_walk_first_kernel, dim3((2 * p0.n_pairs + 63) / 64), dim3(64), 0, stream, p0, walk_in_fill_order ? 1 : 0);
                if (p0.spec_fill) {
                    // the reads the first kernel left: wavefronts of their own, filled again with codes (their number is a device-side fact:
                    // the grids cover the most there can be, what lies beyond leaves at once)
                    const uint32_t max_waves = (p0.n_pairs + 64u / p0.refill_G - 1u) / (64u / p0.refill_G);
                    for (int k = 0; k < 2; ++k) if (!sev[lane][k]) hipEventCreate(&sev[lane][k]);
                    hipEventRecord(sev[lane][0], stream);
                    hipLaunchKernelGGL(gssw_refill_layout_kernel, dim3((max_waves + 63) / 64), dim3(64), 0, stream, p0);
                    GsswParams p = p0;
                    p.spec_fill = 2; p.K = p0.refill_K; p.wave_begin = p0.refill_wave0; p.wave_count = max_waves; p.wave_limit = p0.refill_count;
                    launch_fill(p, stream);
                    hipEventRecord(sev[lane][1], stream); sev_set[lane] = true;
                }
                hipLaunchKernelGGL(gssw_walk_missed_kernel, dim3((p0.n_problems + 255) / 256), dim3(256), 0, stream, p0);      // (lanes beyond the list's end leave at once)
            } else hipLaunchKernelGGL(gssw_walk_kernel, dim3((2 * p0.n_pairs + 255) / 256), dim3(256), 0, stream, p0, walk_in_fill_order ? 1 : 0);
            return;
        }
        GsswParams p = p0;
        for (uint32_t i = 0; i < n; ++i) {                              // the band records of every fill launch's wavefronts
            if (!launches[i].wave_count) continue;
            p.K = launches[i].K; p.wave_begin = launches[i].wave_begin; p.wave_count = launches[i].wave_count;
            switch (p.K) { case 16: launch_band_k<16>(p, stream); break; case 19: launch_band_k<19>(p, stream); break; case 20: launch_band_k<20>(p, stream); break; case 24: launch_band_k<24>(p, stream); break; default: break; }
        }
        hipLaunchKernelGGL(gssw_bandwalk_kernel, dim3((2 * p0.n_pairs + 255) / 256), dim3(256), 0, stream, p0);
        for (uint32_t i = 0; i < n; ++i) {                              // what left its band
            if (!launches[i].wave_count) continue;
            p.K = launches[i].K; p.wave_begin = launches[i].wave_begin; p.wave_count = launches[i].wave_count;
            switch (p.K) { case 16: launch_rewalk_k<16>(p, stream); break; case 19: launch_rewalk_k<19>(p, stream); break; case 20: launch_rewalk_k<20>(p, stream); break; case 24: launch_rewalk_k<24>(p, stream); break; default: break; }
        }
    }
    // One fill launch per rows-per-lane instantiation K (lanes-per-pair G is a per-wavefront value), then one traceback launch over all
    // reads.  (Running walk(c) under fill(c+1) on a second stream, and fusing the walk into the fill kernel,
    // were both measured slower than this plain sequence on MI355X — DESIGN.md §5; `fused` is kept as an option.)
    int run_gssw(const GsswParams& p0, const FillLaunch* launches, uint32_t n, bool walk) override {
        hipSetDevice(dev);
        n_launches = n ? (int)n : 1;
        if (p0.n_problems == 0 || n == 0) { ms_fill = ms_walk = 0; pending = false; return VGK_OK; }
        GsswParams p = p0;
        if (p0.restore_probs) hipLaunchKernelGGL(gssw_refill_restore_kernel, dim3((p0.n_problems + 255) / 256), dim3(256), 0, stream, p0);
        if (fills_take_turns && fill_done_set[1]) hipStreamWaitEvent(stream, fill_done[1], 0);       // (measured: DESIGN.md §5)
        hipEventRecord(ev[0], stream);
        // the (up to three) rows-per-lane instantiations are independent: side streams let a small bucket's
        // launch fill the CUs another bucket leaves idle; everything joins `stream` again before the walk
        for (uint32_t i = 0; i < n; ++i) {
            const FillLaunch& L = launches[i];
            p.K = L.K; p.wave_begin = L.wave_begin; p.wave_count = L.wave_count;
            if (L.wave_count == 0) continue;
            hipStream_t st = (i == 0 || i > 2) ? stream : side[i - 1];
            if (st != stream) hipStreamWaitEvent(st, ev[0], 0);
            int rc = launch_fill(p, st);
            if (rc) return rc;
            if (st != stream) { hipEventRecord(side_done[i - 1], st); hipStreamWaitEvent(stream, side_done[i - 1], 0); }
        }
        hipEventRecord(ev[1], stream);
        hipEventRecord(fill_done[0], stream); fill_done_set[0] = true;
        timed_walk = walk && (!p.fused || p.tb_mode == TB_REWALK);
        if (timed_walk) {
            launch_walk(p0, launches, n, stream);
            hipEventRecord(ev[2], stream);
        }
        pending = true;
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    // lane 1: a second launch stream with timing events of its own, for batches with a single fill launch (several rows-per-lane
    // launches share the side streams and stay on lane 0)
    int run_gssw_on(int lane, const GsswParams& p0, const FillLaunch* launches, uint32_t n, bool walk, void* done) override {
        hipSetDevice(dev);
        if (lane != 1 || n != 1 || p0.n_problems == 0) {
            if (lane == 1) lane1_on_main = true;
            if (hipMemsetAsync(p0.best, 0, ((size_t)p0.n_problems + 1) * sizeof(unsigned long long), stream) != hipSuccess) return VGK_ENODEV;
            int rc = run_gssw(p0, launches, n, walk);
            if (!rc && done) rc = hipEventRecord((hipEvent_t)done, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
            return rc;
        }
        lane1_on_main = false;
        if (hipMemsetAsync(p0.best, 0, ((size_t)p0.n_problems + 1) * sizeof(unsigned long long), alt) != hipSuccess) return VGK_ENODEV;
        GsswParams p = p0;
        p.K = launches[0].K; p.wave_begin = launches[0].wave_begin; p.wave_count = launches[0].wave_count;
        if (p0.restore_probs) hipLaunchKernelGGL(gssw_refill_restore_kernel, dim3((p0.n_problems + 255) / 256), dim3(256), 0, alt, p0);
        if (fills_take_turns && fill_done_set[0]) hipStreamWaitEvent(alt, fill_done[0], 0);
        hipEventRecord(evb[0], alt);
        if (p.wave_count) { const int rc = launch_fill(p, alt); if (rc) return rc; }
        hipEventRecord(evb[1], alt);
        hipEventRecord(fill_done[1], alt); fill_done_set[1] = true;
        timed_walk_b = walk && (!p.fused || p.tb_mode == TB_REWALK);
        if (timed_walk_b) {
            launch_walk(p0, launches, n, alt, 1);
            hipEventRecord(evb[2], alt);
        }
        pending_b = true;
        if (done && hipEventRecord((hipEvent_t)done, alt) != hipSuccess) return VGK_ENODEV;
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    double last_ms_on(int lane, int which) const override {
        if (which == 12) {                                           // the second fill of a speculative batch (0 when the last run had none)
            const int l = (lane == 1 && !lane1_on_main) ? 1 : 0;
            HipBackend* self = const_cast<HipBackend*>(this);
            if (!sev_set[l]) return 0.0;
            hipSetDevice(dev); hipStreamSynchronize(l ? alt : stream);
            float ms = 0.f; if (hipEventElapsedTime(&ms, sev[l][0], sev[l][1]) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
            self->ms_refill[l] = ms; return ms;
        }
        if (lane != 1 || lane1_on_main || which > 2 || which < 0) return last_ms(which);
        HipBackend* self = const_cast<HipBackend*>(this);
        if (self->pending_b) {
            hipSetDevice(dev);
            hipStreamSynchronize(alt);
            hipEventElapsedTime(&self->ms_fill_b, evb[0], evb[1]);
            self->ms_walk_b = 0.f;
            if (timed_walk_b) hipEventElapsedTime(&self->ms_walk_b, evb[1], evb[2]);
            self->pending_b = false;
        }
        return which == 0 ? ms_fill_b : which == 1 ? ms_walk_b : 1.0;
    }
    int ops_offsets(const vgk_result* res, uint32_t n, uint32_t* offs, uint32_t* sums, uint64_t* total) override {
        hipSetDevice(dev);
        const uint32_t blocks = (n + OPS_SCAN_BLOCK - 1) / OPS_SCAN_BLOCK;
        unsigned long long* total_dev = (unsigned long long*)(sums + ((blocks + 2) & ~1u));       // 8-byte aligned, behind the block sums
        hipLaunchKernelGGL(ops_scan_kernel, dim3(blocks), dim3(256), 0, fetch, res, n, offs, sums);
        hipLaunchKernelGGL(ops_sums_kernel, dim3(1), dim3(256), 0, fetch, sums, blocks, total_dev);
        unsigned long long t = 0;
        if (hipMemcpyAsync(&t, total_dev, sizeof t, hipMemcpyDeviceToHost, fetch) != hipSuccess) return VGK_ENODEV;
        if (hipStreamSynchronize(fetch) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        *total = t;
        return VGK_OK;
    }
    int ops_gather(const vgk_result* res, const vgk_op* ops, uint32_t n, const uint32_t* offs, const uint32_t* sums, vgk_result* out_res, vgk_op* out_ops) override {
        hipSetDevice(dev);
        hipLaunchKernelGGL(ops_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, fetch, res, ops, n, offs, sums, out_res, out_ops);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_banded(const BandedParams& p, const BandedLaunch* launches, uint32_t n) override {
        ms_bfill = ms_bwalk = 0.f;
        if (p.n == 0) return VGK_OK;
        const int rc = banded_launch(p, launches, n, bev);
        if (rc != VGK_OK) return rc;
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        hipEventElapsedTime(&ms_bfill, bev[0], bev[1]);
        hipEventElapsedTime(&ms_bwalk, bev[1], bev[2]);
        return VGK_OK;
    }
    int run_banded_async(const BandedParams& p, const BandedLaunch* launches, uint32_t n, int slot) override {
        if (p.n == 0) return VGK_OK;
        hipSetDevice(dev);
        hipEvent_t* ev = bbev[slot & 1];
        for (int k = 0; k < 3; ++k) if (!ev[k] && hipEventCreate(&ev[k]) != hipSuccess) return VGK_ENODEV;
        return banded_launch(p, launches, n, ev);
    }
    int run_banded_geometry(const BGeomParams& p) override {
        if (p.n == 0) return VGK_OK;
        hipSetDevice(dev);
        hipLaunchKernelGGL(banded_geometry_kernel, dim3((p.n + 63) / 64), dim3(64), 0, copy, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    double banded_ms(int slot, int which) override {
        hipEvent_t* ev = bbev[slot & 1]; float ms = 0.f;
        if (!ev[0] || hipEventElapsedTime(&ms, ev[which ? 1 : 0], ev[which ? 2 : 1]) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
        return ms;
    }
    int banded_launch(const BandedParams& p, const BandedLaunch* launches, uint32_t n, hipEvent_t* bev) {
        hipSetDevice(dev);
        hipEventRecord(bev[0], stream);
        // the rows-per-lane classes are independent: the small ones run on the side streams under the big one
        for (uint32_t i = 0; i < n; ++i) {
            const BandedLaunch& L = launches[i];
            if (!L.count) continue;
            hipStream_t st = (i == 0 || i > 2 || L.R >= 64) ? stream : side[i - 1];      // (the HBM-state classes share one slab: one after the other on the main stream)
            if (st != stream) hipStreamWaitEvent(st, bev[0], 0);
            switch (L.R) {
                case 1:  launch_banded_fill<1>(p, L, st); break;
                case 2:  launch_banded_fill<2>(p, L, st); break;
                case 4:  launch_banded_fill<4>(p, L, st); break;
                case 8:  launch_banded_fill<8>(p, L, st); break;
                case 16: launch_banded_fill_blocks<2>(p, L, st); break;      // bands of more than 512 diagonals: blocks of 8 rows per lane (banded_fill_lane_blocks)
                case 32: launch_banded_fill_blocks<4>(p, L, st); break;
                case 64: case 128: case 256: case 512: {                     // 8 ... 64 blocks: their state in HBM (grow-only slab, one stretch per problem of the launch)
                    const uint32_t nb = L.R / 8u;
                    const size_t need = (size_t)L.count * ((size_t)nb * 3u * 8u + 3u * ((size_t)nb + 1u)) * 64u * sizeof(int32_t);
                    if (need > banded_slab_bytes) {
                        hipStreamSynchronize(stream); for (hipStream_t x : side) hipStreamSynchronize(x);
                        if (banded_slab) hipFree(banded_slab);
                        banded_slab = nullptr; banded_slab_bytes = 0;
                        if (hipMalloc(&banded_slab, need + need / 4) != hipSuccess) return VGK_ENOMEM;
                        banded_slab_bytes = need + need / 4;
                    }
                    if (p.quals) hipLaunchKernelGGL(banded_fill_blocks_any_kernel<true>, dim3(L.count), dim3(64), 0, stream, p, L.begin, nb, (int32_t*)banded_slab);
                    else         hipLaunchKernelGGL(banded_fill_blocks_any_kernel<false>, dim3(L.count), dim3(64), 0, stream, p, L.begin, nb, (int32_t*)banded_slab);
                    break; }
                default: return VGK_EINVAL;
            }
            if (st != stream) { hipEventRecord(side_done[i - 1], st); hipStreamWaitEvent(stream, side_done[i - 1], 0); }
        }
        hipEventRecord(bev[1], stream);
        if (!p.scores) hipLaunchKernelGGL(banded_walk_kernel, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);      // k-best mode: the host walks the score matrices
        hipEventRecord(bev[2], stream);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    template <bool MG> void launch_gapless(const GaplessParams& p, uint32_t threads) {
        if (std::getenv("VGAMD_GAPLESS_SLAB_ONLY")) { hipLaunchKernelGGL(gapless_kernel<MG>, dim3((threads + 63) / 64), dim3(64), 0, stream, p, threads, 0); return; }
        hipLaunchKernelGGL(gapless_search_kernel<MG>, dim3((threads + 63) / 64), dim3(64), 0, stream, p, threads);
        // the rules over the searched reads and the slab kernel over the few G_RETRY reads touch different reads (the rules skip
        // G_RETRY, the slab kernel everything else; both take output space from the same atomic counters): side by side
        hipEventRecord(side_done[0], stream);
        hipStreamWaitEvent(side[0], side_done[0], 0);
        hipLaunchKernelGGL(gapless_kernel<MG>, dim3((threads + 63) / 64), dim3(64), 0, side[0], p, threads, 1);
        hipEventRecord(side_done[1], side[0]);
        hipLaunchKernelGGL(gapless_rules_kernel<MG>, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
        hipStreamWaitEvent(stream, side_done[1], 0);
    }
    int run_gapless(const GaplessParams& p, uint32_t threads) override {
        hipSetDevice(dev);
        ms_gapless = 0.f;
        if (!p.n || !threads) return VGK_OK;
        hipEventRecord(bev[0], stream);
        if (p.merge.on) launch_gapless<true>(p, threads); else launch_gapless<false>(p, threads);      // (the kernels without the merged-run code keep the plain search's registers)
        hipEventRecord(bev[1], stream);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        hipEventElapsedTime(&ms_gapless, bev[0], bev[1]);
        return VGK_OK;
    }
    int run_gssw_matrix(const GsswMatrixParams& p) override {
        hipSetDevice(dev);
        if (!p.n) return VGK_OK;
        if (p.go < p.ge || std::getenv("VGAMD_MATRIX_THREADS")) hipLaunchKernelGGL(gssw_matrix_kernel, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
        else {      // one launch per rows-per-lane class; a wavefront whose problem belongs to another class returns at once
            hipLaunchKernelGGL((gssw_matrix_wave_kernel<2>), dim3(p.n), dim3(64), 0, stream, p, 0u, 128u);
            hipLaunchKernelGGL((gssw_matrix_wave_kernel<4>), dim3(p.n), dim3(64), 0, stream, p, 128u, 256u);
            hipLaunchKernelGGL((gssw_matrix_wave_kernel<8>), dim3(p.n), dim3(64), 0, stream, p, 256u, 512u);
            hipLaunchKernelGGL((gssw_matrix_wave_kernel<16>), dim3(p.n), dim3(64), 0, stream, p, 512u, 1024u);
        }
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        return VGK_OK;
    }
    int run_banded_multi(const BandedMultiParams& q) override {
        hipSetDevice(dev);
        if (!q.P.n) return VGK_OK;
        hipLaunchKernelGGL(banded_multi_kernel, dim3((q.P.n + 63) / 64), dim3(64), 0, stream, q);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        return VGK_OK;
    }
    int run_gssw_multi(const GsswMultiParams& p) override {
        hipSetDevice(dev);
        if (!p.M.n) return VGK_OK;
        hipLaunchKernelGGL(gssw_multi_kernel, dim3((p.M.n + 63) / 64), dim3(64), 0, stream, p);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        return VGK_OK;
    }
    int run_gssw_wide(const WideParams& p0, uint32_t n8, uint32_t n16) override {
        hipSetDevice(dev);
        if (!p0.n) return VGK_OK;
        WideParams p = p0;
        for (int k = 0; k < 3; ++k) if (!wev[k]) hipEventCreate(&wev[k]);
        hipEventRecord(wev[0], stream);
        if (n8) { p.order_begin = 0; p.order_count = n8; hipLaunchKernelGGL(gssw_wide_kernel<8>, dim3(n8), dim3(256), 0, stream, p); }
        if (n16) { p.order_begin = n8; p.order_count = n16; hipLaunchKernelGGL(gssw_wide_kernel<16>, dim3(n16), dim3(256), 0, stream, p); }
        hipEventRecord(wev[1], stream);
        hipLaunchKernelGGL(gssw_wide_walk_kernel, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
        hipEventRecord(wev[2], stream); wide_pending = true;
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_xdrop_band(const GsswMatrixParams& p) override {
        ms_xband = 0.f;
        if (!p.n) return VGK_OK;
        const int rc = run_xdrop_band_async(p, 0);
        if (rc != VGK_OK) return rc;
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        ms_xband = (float)xdrop_band_ms(0);
        return VGK_OK;
    }
    double xdrop_band_ms(int slot) override {
        float ms = 0.f;
        if (!xbev[slot & 1][0] || hipEventElapsedTime(&ms, xbev[slot & 1][0], xbev[slot & 1][1]) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
        return ms;
    }
    int run_xdrop_band_async(const GsswMatrixParams& p, int slot) override {
        hipSetDevice(dev);
        if (!p.n) return VGK_OK;
        hipEvent_t* ev = xbev[slot & 1];
        for (int k = 0; k < 2; ++k) if (!ev[k] && hipEventCreate(&ev[k]) != hipSuccess) return VGK_ENODEV;
        hipEventRecord(ev[0], stream);
        auto launch = [&](auto cell) {
            using CT = decltype(cell);
            if (p.xb_order) {
                if (p.xb_n16) hipLaunchKernelGGL(xdrop_band_kernel16<CT>, dim3((p.xb_n16 + 3) / 4), dim3(64), 0, stream, p);
                if (p.xb_n64) hipLaunchKernelGGL(xdrop_band_kernel<CT>, dim3(p.xb_n64), dim3(64), 0, stream, p);
            } else hipLaunchKernelGGL(xdrop_band_kernel<CT>, dim3(p.n), dim3(64), 0, stream, p);
            if (p.xb_results) hipLaunchKernelGGL(xdrop_band_walk_kernel<CT>, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
        };
        if (p.xb_cell16 == 2) {
            if (p.xb_order) {
                if (p.xb_n8) hipLaunchKernelGGL(xdrop_band_pk_kernel8, dim3((p.xb_n8 + 7) / 8), dim3(64), 0, stream, p);
                if (p.xb_n16) hipLaunchKernelGGL(xdrop_band_pk_kernel16, dim3((p.xb_n16 + 3) / 4), dim3(64), 0, stream, p);
                if (p.xb_n64) hipLaunchKernelGGL(xdrop_band_pk_kernel, dim3(p.xb_n64), dim3(64), 0, stream, p);
            } else hipLaunchKernelGGL(xdrop_band_pk_kernel, dim3(p.n), dim3(64), 0, stream, p);
            if (p.xb_results) hipLaunchKernelGGL(xdrop_band_walk_kernel<uint16_t>, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
        } else if (p.xb_cell16) launch(int16_t{}); else launch(int32_t{});
        hipEventRecord(ev[1], stream);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int gapless_order(const GOrderParams& p, int stage) override {
        hipSetDevice(dev);
        if (!p.n) return VGK_OK;
        hipLaunchKernelGGL(gapless_order_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p, stage);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int gapless_seeded(const GSeededParams& p) override {
        hipSetDevice(dev);
        if (!p.n) return VGK_OK;
        hipLaunchKernelGGL(gapless_seeded_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n, int bits) override {
        hipSetDevice(dev);
        if (!n) return VGK_OK;
        const size_t need = hip_sort_tmp_bytes(n);
        if (need > scan_tmp_bytes) {
            if (scan_tmp) { hipStreamSynchronize(stream); hipFree(scan_tmp); scan_tmp = nullptr; scan_tmp_bytes = 0; }
            if (hipMalloc(&scan_tmp, need + need / 4) != hipSuccess) return VGK_ENOMEM;
            scan_tmp_bytes = need + need / 4;
        }
        return hip_sort_pairs_u32(kin, kout, vin, vout, n, bits, scan_tmp, scan_tmp_bytes, stream);
    }
    int mask_reads(char* reads, size_t bytes) override {
        hipSetDevice(dev);
        if (!bytes) return VGK_OK;
        hipLaunchKernelGGL(mask_reads_kernel, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, stream, reads, bytes);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_minimizer(const MinimizerParams& p) override {
        hipSetDevice(dev);
        if (!p.n) return VGK_OK;
        const size_t need = sizeof(vgk_seed) * (size_t)p.n * MZ_MAX_SEEDS;       // a slot of MZ_MAX_SEEDS seeds per read between the two launches
        if (need > mz_slots_bytes) {
            if (mz_slots) { hipStreamSynchronize(stream); hipFree(mz_slots); mz_slots = nullptr; mz_slots_bytes = 0; }
            if (hipMalloc(&mz_slots, need + need / 8) != hipSuccess) return VGK_ENOMEM;
            mz_slots_bytes = need + need / 8;
        }
        if (p.pass == 1) { if (p.hi > p.lo) hipLaunchKernelGGL(minimizer_kernel, dim3((p.hi - p.lo + 3) / 4), dim3(256), 0, stream, p, (vgk_seed*)mz_slots); }
        else hipLaunchKernelGGL(minimizer_gather_kernel, dim3((p.n + 3) / 4), dim3(256), 0, stream, p, (const vgk_seed*)mz_slots);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_tail_stage(const TStageParams& p, int what) override {
        hipSetDevice(dev);
        const uint32_t items = tstage_items(p, what);
        if (!items) return VGK_OK;
        hipLaunchKernelGGL(tail_stage_kernel, dim3((items + 255) / 256), dim3(256), 0, stream, p, what, items);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_minimizer_list(const MzListParams& p) override {
        hipSetDevice(dev);
        hipLaunchKernelGGL(minimizer_list_kernel, dim3((p.n + 64) / 64), dim3(64), 0, stream, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_minimizer_seeds_of(const MzSeedsOfParams& p) override {
        hipSetDevice(dev);
        hipLaunchKernelGGL(minimizer_seeds_of_kernel, dim3((p.n + 256) / 256), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_wfa_mask(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, uint32_t n) override {
        hipSetDevice(dev);
        if (n) hipLaunchKernelGGL(wfa_mask_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, probs, src_off, raw, seqs, n);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_chain_stitch(const CsParams& p, int what) override {
        hipSetDevice(dev);
        if (what == CS_GATHER) hipLaunchKernelGGL(chain_gather_kernel, dim3((p.n_reads + 3) / 4), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(chain_stitch_kernel, dim3((p.n_reads + 64) / 64), dim3(64), 0, stream, p, what);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_rescue_requests(const RqParams& p, int what) override {
        hipSetDevice(dev);
        hipLaunchKernelGGL(rescue_requests_kernel, dim3((p.n_pairs + 256) / 256), dim3(256), 0, stream, p, what);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_tail(const TailParams& p, uint32_t threads) override {
        hipSetDevice(dev);
        if (!p.n || !threads) return VGK_OK;
        hipLaunchKernelGGL(tail_walk_kernel, dim3((threads + 63) / 64), dim3(64), 0, stream, p, threads);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int scan_u32(const uint32_t* in, uint32_t* out, uint32_t n) override {
        hipSetDevice(dev);
        if (!n) return VGK_OK;
        const size_t need = hip_scan_tmp_bytes(n);
        if (need > scan_tmp_bytes) {
            if (scan_tmp) { hipStreamSynchronize(stream); hipFree(scan_tmp); scan_tmp = nullptr; scan_tmp_bytes = 0; }
            if (hipMalloc(&scan_tmp, need + need / 4) != hipSuccess) return VGK_ENOMEM;
            scan_tmp_bytes = need + need / 4;
        }
        return hip_scan_u32(in, out, n, scan_tmp, scan_tmp_bytes, stream);
    }
    int forest_flags(const ForestParams& p) override {
        hipSetDevice(dev);
        if (!p.n_nodes) return VGK_OK;
        hipLaunchKernelGGL(forest_flags_kernel, dim3((p.n_nodes + 255) / 256), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int forest_emit(const ForestParams& p) override {
        hipSetDevice(dev);
        if (!p.n_nodes) return VGK_OK;
        hipLaunchKernelGGL(forest_emit_kernel, dim3((p.n_nodes + 255) / 256), dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int fill(void* dst, int byte, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemsetAsync(dst, byte, bytes, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    void watch(int which) override { hipSetDevice(dev); hipEventRecord(bev[which ? 1 : 0], stream); }
    double watch_ms() override { float ms = 0.f; hipEventElapsedTime(&ms, bev[0], bev[1]); return ms; }
    int run_wfa(const WfaParams& p, uint32_t threads) override {
        hipSetDevice(dev);
        ms_wfa = 0.f;
        if (!p.n || !threads) return VGK_OK;
        hipEventRecord(bev[0], stream);
        hipLaunchKernelGGL(wfa_kernel, dim3((threads + 63) / 64), dim3(64), 0, stream, p, threads);
        hipEventRecord(bev[1], stream);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        hipEventElapsedTime(&ms_wfa, bev[0], bev[1]);
        return VGK_OK;
    }
    int run_wfa_wave(const WwParams& p, uint32_t waves) override {
        hipSetDevice(dev);
        if (!p.n_todo || !waves) return VGK_OK;
        hipEventRecord(bev[0], stream);
        hipLaunchKernelGGL(wfa_wave_kernel, dim3(waves), dim3(64), 0, stream, p);
        hipEventRecord(bev[1], stream);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        float ms = 0.f; hipEventElapsedTime(&ms, bev[0], bev[1]);
        ms_wfa += ms;
        return VGK_OK;
    }
    bool wfa_concurrent() const override { return alt != nullptr; }
    int run_wfa_hybrid(const WfaParams& p, uint32_t threads, const WwParams& a, uint32_t waves) override {
        hipSetDevice(dev);
        ms_wfa = 0.f;
        if (!p.n || !threads || !waves) return VGK_OK;
        hipEventRecord(bev[0], stream);
        hipStreamWaitEvent(alt, bev[0], 0);                        // both start behind whatever the main stream held
        hipLaunchKernelGGL(wfa_kernel, dim3((threads + 63) / 64), dim3(64), 0, stream, p, threads);
        hipLaunchKernelGGL(wfa_wave_kernel, dim3(waves), dim3(64), 0, alt, a);
        hipEventRecord(bev[2], alt);
        hipStreamWaitEvent(stream, bev[2], 0);
        hipEventRecord(bev[1], stream);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) return VGK_ENODEV;
        hipEventElapsedTime(&ms_wfa, bev[0], bev[1]);
        return VGK_OK;
    }
    void reset_wfa_ms() override { ms_wfa = 0.f; }
    hipEvent_t wev[3] = {nullptr, nullptr, nullptr}; bool wide_pending = false; float ms_wide_fill = 0.f, ms_wide_walk = 0.f;
    double last_ms(int which) const override {
        if (which == 13 || which == 14) {                                // the wide route's last launch: fill | traceback
            HipBackend* self = const_cast<HipBackend*>(this);
            if (self->wide_pending) { hipSetDevice(dev); hipEventSynchronize(wev[2]); hipEventElapsedTime(&self->ms_wide_fill, wev[0], wev[1]); hipEventElapsedTime(&self->ms_wide_walk, wev[1], wev[2]); self->wide_pending = false; }
            return which == 13 ? ms_wide_fill : ms_wide_walk;
        }
        if (which == 6) return ms_wfa;
        if (which == 7) return ms_xband;
        if (which == 5) return ms_gapless;
        if (which == 3) return ms_bfill;
        if (which == 4) return ms_bwalk;
        HipBackend* self = const_cast<HipBackend*>(this);
        if (self->pending) {
            hipSetDevice(dev);
            hipStreamSynchronize(stream);
            hipEventElapsedTime(&self->ms_fill, ev[0], ev[1]);
            self->ms_walk = 0.f;
            if (timed_walk) hipEventElapsedTime(&self->ms_walk, ev[1], ev[2]);
            self->pending = false;
        }
        return which == 0 ? ms_fill : which == 1 ? ms_walk : (double)n_launches;
    }
};

Backend* make_backend(int device, std::string& err) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { err = std::string("no HIP device: ") + hipGetErrorString(e); return nullptr; }
    if (device < 0 || device >= n) { err = "HIP device index out of range"; return nullptr; }
    auto* b = new HipBackend();
    b->dev = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&b->prop, device) != hipSuccess) {
        err = "cannot select HIP device"; delete b; return nullptr; }
    // The side (copy) and fetch streams carry what the host WAITS for beside a main stream that is busy — small kernels (packers, the banded
    // geometry), descriptors up, results down: they get the device's highest stream priority, so that their work takes the slots a
    // saturating fill frees first instead of queueing behind all of its workgroups (VGAMD_STREAM_PRIORITY=0: all streams alike).
    int prio_least = 0, prio_greatest = 0;
    const bool use_prio = !(std::getenv("VGAMD_STREAM_PRIORITY") && std::atoi(std::getenv("VGAMD_STREAM_PRIORITY")) == 0) &&
                          hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_greatest != prio_least;
    auto make_stream = [&](hipStream_t* st, bool high) {
        if (high && use_prio && hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest) == hipSuccess) return true;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
    };
    if (!make_stream(&b->stream, false) || !make_stream(&b->copy, true) || !make_stream(&b->fetch, true) || !make_stream(&b->alt, false)) { err = "cannot create HIP stream"; delete b; return nullptr; }
    for (int i = 0; i < 2; ++i)
        if (hipStreamCreateWithFlags(&b->side[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&b->side_done[i], hipEventDisableTiming) != hipSuccess) { err = "cannot create HIP side stream"; delete b; return nullptr; }
    for (auto& ev : b->bev) if (hipEventCreate(&ev) != hipSuccess) { err = "cannot create HIP event"; delete b; return nullptr; }
    for (auto& ev : b->ev) if (hipEventCreate(&ev) != hipSuccess) { err = "cannot create HIP event"; delete b; return nullptr; }
    for (auto& ev : b->evb) if (hipEventCreate(&ev) != hipSuccess) { err = "cannot create HIP event"; delete b; return nullptr; }
    for (auto& ev : b->fill_done) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { err = "cannot create HIP event"; delete b; return nullptr; }
    return b;
}

}  // namespace vgk
