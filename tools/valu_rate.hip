// Micro-benchmark: issue rate of the VALU instructions the DP kernels are built from
// (gfx950).  Each kernel runs N_ITER x 64 independent instructions of one kind per wave
// with 8 waves/SIMD resident; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 4096
#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY(ASM) \
    uint32_t a[8]; for (int i = 0; i < 8; ++i) a[i] = p[threadIdx.x + i * 64]; uint32_t b = p[1], c = p[2]; \
    for (int it = 0; it < N_ITER; ++it) { REP16( \
        asm volatile(ASM : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[1]) : "v"(b), "v"(c)); \
        asm volatile(ASM : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[3]) : "v"(b), "v"(c)); ) } \
    uint32_t r = 0; for (int i = 0; i < 8; ++i) r += a[i]; p[threadIdx.x] = r;
__global__ void k_pk_add(uint32_t* p)  { BODY("v_pk_add_u16 %0, %0, %1") }
__global__ void k_pk_max(uint32_t* p)  { BODY("v_pk_max_u16 %0, %0, %1") }
__global__ void k_pk_subc(uint32_t* p) { BODY("v_pk_sub_u16 %0, %0, %1 clamp") }
__global__ void k_pk_mad(uint32_t* p)  { BODY("v_pk_mad_u16 %0, %0, %1, %2") }
__global__ void k_add(uint32_t* p)     { BODY("v_add_u32 %0, %0, %1") }
__global__ void k_max(uint32_t* p)     { BODY("v_max_u32 %0, %0, %1") }
__global__ void k_max3(uint32_t* p)    { BODY("v_max3_u32 %0, %0, %1, %2") }
__global__ void k_subc(uint32_t* p)    { BODY("v_sub_u32_e64 %0, %0, %1 clamp") }
__global__ void k_perm(uint32_t* p)    { BODY("v_perm_b32 %0, %0, %1, %2") }
__global__ void k_bfe(uint32_t* p)     { BODY("v_bfe_i32 %0, %0, %1, 8") }
__global__ void k_cmp(uint32_t* p)     { BODY("v_cmp_ne_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc") }
__global__ void k_max16(uint32_t* p)   { BODY("v_max_u16 %0, %0, %1") }
__global__ void k_lshlor(uint32_t* p)  { BODY("v_lshl_or_b32 %0, %0, 1, %1") }
__global__ void k_fma(uint32_t* p)     { BODY("v_fma_f32 %0, %0, %1, %2") }
__global__ void k_pkfma(uint32_t* p)   { BODY("v_pk_fma_f16 %0, %0, %1, %2") }
__global__ void k_and(uint32_t* p)     { BODY("v_and_b32 %0, %0, %1") }
__global__ void k_or(uint32_t* p)      { BODY("v_or_b32 %0, %0, %1") }
__global__ void k_bfi(uint32_t* p)     { BODY("v_bfi_b32 %0, %0, %1, %2") }
__global__ void k_mov(uint32_t* p)     { BODY("v_mov_b32 %0, %1") }
__global__ void k_lshl(uint32_t* p)    { BODY("v_lshlrev_b32 %0, 3, %0") }
__global__ void k_andor(uint32_t* p)   { BODY("v_and_or_b32 %0, %0, %1, %2") }
__global__ void k_min16(uint32_t* p)   { BODY("v_min_u16 %0, %0, %1") }
__global__ void k_sub16(uint32_t* p)   { BODY("v_sub_u16 %0, %0, %1") }
__global__ void k_add16s(uint32_t* p)  { BODY("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") }
__global__ void k_pklshl(uint32_t* p)  { BODY("v_pk_lshlrev_b16 %0, 3, %0 op_sel_hi:[0,1]") }
template <class F> void run(const char* name, F f, uint32_t* d, int per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 blocks of 256 threads per CU = 8 waves per SIMD
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d);
    hipEventRecord(e0); hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * N_ITER * 64.0 * per_iter;     // wave-instructions
    double per_simd = winstr / (256.0 * 4);
    printf("%-10s %8.3f ms  %6.2f ns/winstr/SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    uint32_t* d; hipMalloc(&d, 1 << 20); hipMemset(d, 1, 1 << 20);
    run("pk_add_u16", k_pk_add, d, 1); run("pk_max_u16", k_pk_max, d, 1); run("pk_sub_clamp", k_pk_subc, d, 1); run("pk_mad_u16", k_pk_mad, d, 1);
    run("add_u32", k_add, d, 1); run("max_u32", k_max, d, 1); run("max3_u32", k_max3, d, 1); run("sub_clamp", k_subc, d, 1);
    run("perm_b32", k_perm, d, 1); run("bfe_i32", k_bfe, d, 1); run("cmp+addc", k_cmp, d, 2); run("max_u16", k_max16, d, 1);
    run("lshl_or", k_lshlor, d, 1); run("fma_f32", k_fma, d, 1); run("pk_fma_f16", k_pkfma, d, 1);
    run("and_b32", k_and, d, 1); run("or_b32", k_or, d, 1); run("bfi_b32", k_bfi, d, 1); run("mov_b32", k_mov, d, 1);
    run("lshlrev_b32", k_lshl, d, 1); run("and_or_b32", k_andor, d, 1); run("min_u16", k_min16, d, 1); run("sub_u16", k_sub16, d, 1);
    run("add_u16_sdwa", k_add16s, d, 1); run("pk_lshlrev", k_pklshl, d, 1);
    return 0;
}
