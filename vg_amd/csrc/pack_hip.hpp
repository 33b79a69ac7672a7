// pack_hip.hpp — entry points of pack_hip.hip (device-side packing of window problems), called by backend_hip.hip
#pragma once
#include <hip/hip_runtime.h>
#include "gssw_pack_device.hpp"

namespace vgk {
size_t hip_win_tmp_bytes(uint32_t n, uint32_t n_waves_cap);                               // scratch the sort / scans need
int    hip_win_stage1(const WinParams& P, void* tmp, size_t tmp_bytes, hipStream_t st);   // sizes + their prefix sums
int    hip_win_stage2(const WinParams& P, void* tmp, size_t tmp_bytes, hipStream_t st);   // order, wavefronts, arenas
size_t hip_sort_tmp_bytes(uint32_t n);                                                    // a stable radix sort of (key, value) pairs (rocPRIM)
int    hip_sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n, int bits, void* tmp, size_t tmp_bytes, hipStream_t st);
size_t hip_scan_tmp_bytes(uint32_t n);                                                    // a plain exclusive prefix sum of n 32-bit values (rocPRIM)
int    hip_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* tmp, size_t tmp_bytes, hipStream_t st);
}
