// pack_hip.hip — device-side packing of window problems (gssw_pack_device.hpp) on gfx950: four small kernels around a radix
// sort and a few prefix sums.  The sort and the scans are rocPRIM's (through hipcub): plumbing over a million 4-byte keys that
// takes ~0.2 ms; everything that knows about graphs, reads and wavefronts is in gssw_pack_device.hpp.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "pack_hip.hpp"

namespace vgk {

static __device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
}
static __device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_down(v, d, 64); v = o > v ? o : v; }
    return v;
}

// sizes: a fixed grid strides over the problems; per-thread sums -> wavefront -> block (LDS) -> one set of atomics per block
__global__ __launch_bounds__(256) void win_size_kernel(const WinParams P) {
    __shared__ unsigned long long part[4][WIN_COLS + 3];
    __shared__ uint32_t part_max[4], part_tb[4];
    WinAcc acc; win_acc_clear(acc);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < P.n; i += gridDim.x * 256u) win_size_one(P, i, acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) for (uint32_t k = 0; k < WIN_COLS; ++k) P.sizes[k * (P.n + 1) + P.n] = 0;   // the scans run over n + 1 entries
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long v[WIN_COLS + 3];
#pragma unroll
    for (uint32_t k = 0; k < WIN_COLS; ++k) v[k] = acc.tot[k];
    v[WIN_COLS] = acc.cells; v[WIN_COLS + 1] = acc.tb_cells; v[WIN_COLS + 2] = acc.in_bytes;
#pragma unroll
    for (uint32_t k = 0; k < WIN_COLS + 3; ++k) { const unsigned long long s = wave_sum(v[k]); if (lane == 0) part[w][k] = s; }
    const uint32_t mx = wave_max(acc.max_rows), tb = wave_max(acc.want_tb);
    if (lane == 0) { part_max[w] = mx; part_tb[w] = tb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        WinAcc tot; win_acc_clear(tot);
        for (uint32_t q = 0; q < 4; ++q) {
            for (uint32_t k = 0; k < WIN_COLS; ++k) tot.tot[k] += part[q][k];
            tot.cells += part[q][WIN_COLS]; tot.tb_cells += part[q][WIN_COLS + 1]; tot.in_bytes += part[q][WIN_COLS + 2];
            tot.max_rows = tot.max_rows > part_max[q] ? tot.max_rows : part_max[q]; tot.want_tb |= part_tb[q];
        }
        win_acc_flush(P, tot);
    }
}
__global__ __launch_bounds__(256) void win_bucket_first_kernel(const WinParams P) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j < P.n) win_bucket_first_one(P, j);
}
__global__ void win_buckets_kernel(const WinParams P) { if (threadIdx.x == 0 && blockIdx.x == 0) win_buckets(P); }
__global__ __launch_bounds__(256) void win_wave_kernel(const WinParams P) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w < P.n_waves_cap) win_wave_one(P, w);
}
__global__ __launch_bounds__(256) void win_wave_tb_kernel(const WinParams P) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w < P.n_waves_cap) win_wave_tb_one(P, w);
}
// one wavefront per problem: 64 lanes share its nodes, columns and read bases
__global__ __launch_bounds__(256) void win_emit_kernel(const WinParams P) {
    const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i < P.n) win_emit_one(P, i, threadIdx.x & 63u, 64u);
}

size_t hip_win_tmp_bytes(uint32_t n, uint32_t n_waves_cap) {
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 25, (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, n + 1, (hipStream_t)0);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, (unsigned long long*)nullptr, (unsigned long long*)nullptr, n_waves_cap + 1, (hipStream_t)0);
    size_t m = a > b ? a : b; m = m > c ? m : c;
    return m + 256;
}

int hip_win_stage1(const WinParams& P, void* tmp, size_t tmp_bytes, hipStream_t st) {
    if (!P.n) return VGK_OK;
    const uint32_t blocks = (P.n + 255) / 256 < 1024u ? (P.n + 255) / 256 : 1024u;
    hipLaunchKernelGGL(win_size_kernel, dim3(blocks), dim3(256), 0, st, P);
    const uint32_t n1 = P.n + 1;
    for (uint32_t k = 0; k < WIN_COLS; ++k) {
        size_t bytes = tmp_bytes;
        if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, (const uint32_t*)(P.sizes + (size_t)k * n1), P.offs + (size_t)k * n1, n1, st) != hipSuccess) return VGK_ENODEV;
    }
    return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
}

int hip_win_stage2(const WinParams& P, void* tmp, size_t tmp_bytes, hipStream_t st) {
    if (!P.n) return VGK_OK;
    size_t bytes = tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, bytes, (const uint32_t*)P.key, P.key_sorted, (const uint32_t*)P.idx, P.idx_sorted, P.n, 0, 25, st) != hipSuccess) return VGK_ENODEV;
    if (hipMemsetAsync(P.bucket_first, 0xff, sizeof(uint32_t) * WIN_BUCKETS, st) != hipSuccess) return VGK_ENODEV;
    if (hipMemsetAsync(P.wave_tb, 0, sizeof(unsigned long long) * ((size_t)P.n_waves_cap + 1), st) != hipSuccess) return VGK_ENODEV;
    hipLaunchKernelGGL(win_bucket_first_kernel, dim3((P.n + 255) / 256), dim3(256), 0, st, P);
    hipLaunchKernelGGL(win_buckets_kernel, dim3(1), dim3(64), 0, st, P);
    hipLaunchKernelGGL(win_wave_kernel, dim3((P.n_waves_cap + 255) / 256), dim3(256), 0, st, P);
    bytes = tmp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, P.wave_tb, P.wave_tb, P.n_waves_cap + 1, st) != hipSuccess) return VGK_ENODEV;
    hipLaunchKernelGGL(win_wave_tb_kernel, dim3((P.n_waves_cap + 255) / 256), dim3(256), 0, st, P);
    hipLaunchKernelGGL(win_emit_kernel, dim3((P.n + 3) / 4), dim3(256), 0, st, P);
    return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
}

size_t hip_sort_tmp_bytes(uint32_t n) {
    size_t b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 32, (hipStream_t)0);
    return b + 256;
}
int hip_sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n, int bits, void* tmp, size_t tmp_bytes, hipStream_t st) {
    size_t bytes = tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, bytes, kin, kout, vin, vout, n, 0, bits, st) != hipSuccess) return VGK_ENODEV;
    return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
}
size_t hip_scan_tmp_bytes(uint32_t n) {
    size_t b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, (hipStream_t)0);
    return b + 256;
}
int hip_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* tmp, size_t tmp_bytes, hipStream_t st) {
    size_t bytes = tmp_bytes;
    if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, in, out, n, st) != hipSuccess) return VGK_ENODEV;
    return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
}

}  // namespace vgk
