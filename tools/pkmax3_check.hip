// Is gfx950's v_pk_maximum3_f16, applied to the BIT PATTERNS of unsigned 16-bit values below 0x7c00, the unsigned max of three per
// half?  (Non-negative f16 values order like their bit patterns; the question is what the instruction does with denormals, i.e. with
// every DP value below 0x0400.)  Exhaustive over the low halves' (a, b) pairs, the high halves running a permutation of them, c over
// a set of edge patterns + a and b themselves.  Prints the number of mismatches (0 = usable by gssw_device.hpp's VGK_H_MAX3) and the
// instruction's issue rate beside v_pk_max_u16's.
//   hipcc -O3 --offload-arch=gfx950 tools/pkmax3_check.hip -o tools/pkmax3_check && tools/pkmax3_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr uint32_t LIM = 0x7c00;
__device__ __forceinline__ uint32_t max3f(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__global__ void check(unsigned long long* bad, uint32_t* first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)LIM * LIM) return;
    const uint32_t a = (uint32_t)(i % LIM), b = (uint32_t)(i / LIM);
    const uint32_t a2 = (a * 7u + 3u) % LIM, b2 = (b * 13u + 11u) % LIM;
    const uint32_t cs[12] = {0, 1, 3, 7, 8, 0x3ff, 0x400, 0x401, 0x3ff8, 0x7bff, a, b};
    for (int k = 0; k < 12; ++k) {
        const uint32_t c = cs[k], c2 = (c * 5u + 1u) % LIM;
        const uint32_t got = max3f(a | (a2 << 16), b | (b2 << 16), c | (c2 << 16));
        const uint32_t want = umax(umax(a, b), c) | (umax(umax(a2, b2), c2) << 16);
        if (got != want) { if (atomicAdd(bad, 1ull) == 0) { first[0] = a | (a2 << 16); first[1] = b | (b2 << 16); first[2] = c | (c2 << 16); first[3] = got; first[4] = want; } }
    }
}
#define N_ITER 4096
#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY(ASM) \
    uint32_t a[8]; for (int i = 0; i < 8; ++i) a[i] = p[threadIdx.x + i * 64] & 0x3fff3fffu; uint32_t b = p[1] & 0x3fff3fffu, c = p[2] & 0x3fff3fffu; \
    for (int it = 0; it < N_ITER; ++it) { REP16( \
        asm volatile(ASM : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[1]) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[3]) : "v"(b), "v"(c)); ) } \
    uint32_t r = 0; for (int i = 0; i < 8; ++i) r += a[i]; p[threadIdx.x] = r;
__global__ void k_pk_max(uint32_t* p)   { BODY("v_pk_max_u16 %0, %0, %1") }
__global__ void k_pk_max3(uint32_t* p)  { BODY("v_pk_maximum3_f16 %0, %0, %1, %2") }
__global__ void k_pk_maxf(uint32_t* p)  { BODY("v_pk_max_f16 %0, %0, %1") }
__global__ void k_lshlor(uint32_t* p)   { BODY("v_lshl_or_b32 %0, %0, 4, %1") }
__global__ void k_pk_mad(uint32_t* p)   { BODY("v_pk_mad_u16 %0, %0, %1, %2") }
template <class F> void run(const char* name, F f, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(e0); hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * N_ITER * 64.0 / (256.0 * 4);
    printf("%-18s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
}
int main() {
    unsigned long long* bad; uint32_t* first; hipMalloc(&bad, 8); hipMalloc(&first, 32); hipMemset(bad, 0, 8); hipMemset(first, 0, 32);
    const uint64_t n = (uint64_t)LIM * LIM;
    hipLaunchKernelGGL(check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, bad, first);
    unsigned long long h = 0; uint32_t f[5]; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 20, hipMemcpyDeviceToHost);
    printf("pk_maximum3_f16 as unsigned max3 below 0x7c00: %llu mismatches of %llu\n", h, (unsigned long long)n * 12);
    if (h) printf("  first: a %08x b %08x c %08x got %08x want %08x\n", f[0], f[1], f[2], f[3], f[4]);
    uint32_t* d; hipMalloc(&d, 1 << 20); hipMemset(d, 1, 1 << 20);
    run("pk_max_u16", k_pk_max, d); run("pk_maximum3_f16", k_pk_max3, d); run("pk_max_f16", k_pk_maxf, d); run("lshl_or_b32", k_lshlor, d); run("pk_mad_u16", k_pk_mad, d);
    return 0;
}
