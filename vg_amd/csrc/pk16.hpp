// pk16.hpp — two unsigned 16-bit DP cells per 32-bit register.
//
// On gfx950 these map 1:1 onto CDNA4 packed VALU instructions (v_pk_add_u16,
// v_pk_sub_u16 [clamp], v_pk_max_u16, v_pk_min_u16, v_pk_mad_u16, v_perm_b32).
// The scalar bodies under !__HIP_DEVICE_COMPILE__ exist so the very same lane
// code can be stepped on a CPU by the lock-step emulator in tests/emu (a test
// of the kernel logic; never part of the product path).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VGK_HD __host__ __device__ __forceinline__
#else
#define VGK_HD inline
#endif

namespace vgk {

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2u16 as_v(uint32_t x) { return __builtin_bit_cast(v2u16, x); }
static __device__ __forceinline__ uint32_t as_u(v2u16 x) { return __builtin_bit_cast(uint32_t, x); }
static __device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_v(a) + as_v(b)); }
static __device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_v(a) - as_v(b)); }
static __device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_sub_sat(as_v(a), as_v(b))); }
static __device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_v(a), as_v(b))); }
static __device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_v(a), as_v(b))); }
static __device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return as_u(as_v(a) * as_v(b) + as_v(c)); }
// byte permute: selector byte k picks byte (sel&7) of {hi:lo} = {a:b}; 0x0c -> 0x00
static __device__ __forceinline__ uint32_t byte_perm(uint32_t a, uint32_t b, uint32_t sel) { return __builtin_amdgcn_perm(a, b, sel); }
// a*M + c per half, M an inline constant: exactly one v_pk_mad_u16 (hipcc otherwise
// expands small-constant multiplies into shift + SDWA-or sequences)
template <int M> static __device__ __forceinline__ uint32_t pk_mul_add_imm(uint32_t a, uint32_t c) {
    uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "n"(M), "v"(c)); return r; }
// a*b + C per half, b a wave-uniform packed multiplier, C an inline constant
template <int C> static __device__ __forceinline__ uint32_t pk_mad_add_imm(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "s"(b), "n"(C)); return r; }
// (mask & a) | (~mask & b) and (a << N) | c as the single instructions they are (hipcc splits them into and / shift / or3 chains)
template <uint32_t MASK> static __device__ __forceinline__ uint32_t bit_select(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(MASK), "v"(a), "v"(b)); return r; }
template <int N> static __device__ __forceinline__ uint32_t shl_or(uint32_t a, uint32_t c) {
    uint32_t r; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(N), "v"(c)); return r; }
// max of three per half in ONE instruction: gfx950's v_pk_maximum3_f16 on the bit patterns.  Non-negative f16 values order like their
// bit patterns, so for halves below 0x7c00 (no Inf / NaN; every DP quantity stays below 0x4000) this IS the unsigned 16-bit max3 —
// provided denormals are not flushed (tools/pkmax3_check.hip verifies that on the device).
static __device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// a prefix maximum over the rows of packed pairs (xdrop_band_pk_lane): VOP3P's op_sel picks, per result half, which half of each source it reads
//   pk_max_lo_into_hi(a):  lo = a.lo,               hi = max(a.hi, a.lo)
//   pk_max_bhi(a, b):      lo = max(a.lo, b.hi),    hi = max(a.hi, b.hi)
static __device__ __forceinline__ uint32_t pk_max_lo_into_hi(uint32_t a) {
    uint32_t r; asm("v_pk_max_u16 %0, %1, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r; }
static __device__ __forceinline__ uint32_t pk_max_bhi(uint32_t a, uint32_t b) {
    uint32_t r; asm("v_pk_max_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the 32 bits that start 16 bits into {hi:lo}: (lo >> 16) | (hi << 16) — the rows of two neighbouring pairs moved down by one
static __device__ __forceinline__ uint32_t align16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#else
static inline uint32_t pk_max_lo_into_hi(uint32_t a) { const uint32_t lo = a & 0xffffu, hi = a >> 16; return lo | ((hi > lo ? hi : lo) << 16); }
static inline uint32_t pk_max_bhi(uint32_t a, uint32_t b) { const uint32_t lo = a & 0xffffu, hi = a >> 16, t = b >> 16; return (lo > t ? lo : t) | ((hi > t ? hi : t) << 16); }
static inline uint32_t align16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }
template <uint32_t MASK> static inline uint32_t bit_select(uint32_t a, uint32_t b) { return (MASK & a) | (~MASK & b); }
template <int N> static inline uint32_t shl_or(uint32_t a, uint32_t c) { return (a << N) | c; }
static inline uint32_t pk_lo(uint32_t x) { return x & 0xffffu; }
static inline uint32_t pk_hi(uint32_t x) { return x >> 16; }
static inline uint32_t pk_mk(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
static inline uint32_t pk_add(uint32_t a, uint32_t b) { return pk_mk(pk_lo(a) + pk_lo(b), pk_hi(a) + pk_hi(b)); }
static inline uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_mk(pk_lo(a) - pk_lo(b), pk_hi(a) - pk_hi(b)); }
static inline uint32_t sat_sub16(uint32_t a, uint32_t b) { return a > b ? a - b : 0; }
static inline uint32_t pk_subs(uint32_t a, uint32_t b) { return pk_mk(sat_sub16(pk_lo(a), pk_lo(b)), sat_sub16(pk_hi(a), pk_hi(b))); }
static inline uint32_t pk_max(uint32_t a, uint32_t b) { return pk_mk(pk_lo(a) > pk_lo(b) ? pk_lo(a) : pk_lo(b), pk_hi(a) > pk_hi(b) ? pk_hi(a) : pk_hi(b)); }
static inline uint32_t pk_min(uint32_t a, uint32_t b) { return pk_mk(pk_lo(a) < pk_lo(b) ? pk_lo(a) : pk_lo(b), pk_hi(a) < pk_hi(b) ? pk_hi(a) : pk_hi(b)); }
static inline uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return pk_mk(pk_lo(a) * pk_lo(b) + pk_lo(c), pk_hi(a) * pk_hi(b) + pk_hi(c)); }
static inline uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) { return pk_max(pk_max(a, b), c); }
template <int M> static inline uint32_t pk_mul_add_imm(uint32_t a, uint32_t c) { return pk_mad(a, (uint32_t)M * 0x00010001u, c); }
template <int C> static inline uint32_t pk_mad_add_imm(uint32_t a, uint32_t b) { return pk_mad(a, b, (uint32_t)C * 0x00010001u); }
static inline uint32_t byte_perm(uint32_t a, uint32_t b, uint32_t sel) {
    uint64_t src = ((uint64_t)a << 32) | b;
    uint32_t out = 0;
    for (int k = 0; k < 4; ++k) {
        uint32_t s = (sel >> (8 * k)) & 0xff, byte;
        if (s <= 7) byte = (uint32_t)(src >> (8 * s)) & 0xff;
        else if (s == 0x0c) byte = 0x00;
        else if (s >= 0x0d) byte = 0xff;
        else byte = 0;   // 8..11 (sign replicate) never used here
        out |= byte << (8 * k);
    }
    return out;
}
#endif

// Full-width adds / subtracts that are exact on packed halves when no carry / borrow can cross:
// pk_add_nc needs lo(a)+lo(b) < 2^16, pk_sub_nb needs every half of a >= the same half of b.
// v_add_u32 / v_sub_u32 issue at twice the rate of the v_pk_* forms on gfx950 (tools/valu_rate.hip).
VGK_HD uint32_t pk_add_nc(uint32_t a, uint32_t b) { return a + b; }
VGK_HD uint32_t pk_sub_nb(uint32_t a, uint32_t b) { return a - b; }

// replace the low / high 16-bit half
VGK_HD uint32_t set_lo(uint32_t x, uint32_t v) { return (x & 0xffff0000u) | (v & 0xffffu); }
VGK_HD uint32_t set_hi(uint32_t x, uint32_t v) { return (x & 0x0000ffffu) | (v << 16); }

}  // namespace vgk
