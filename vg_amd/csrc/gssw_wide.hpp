// gssw_wide.hpp — the wide (int32, any read length) route of vgk_gssw_align: gssw_wide_api.cpp, gssw_wide_device.hpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/vgk.h"

struct vgk_ctx;

namespace vgk {

// VGK_OK when the wide kernels take the problem; else the status the problem is answered with
int wide_problem_status(const vgk_ctx* ctx, const vgk_gssw_problem& p);

// Aligns problems[idx[0 .. m)]; results[idx[k]] and the ops (appended to ops[] at *ops_at, which moves on) as vgk_gssw_align returns them.
int wide_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, const uint32_t* idx, uint32_t m,
               vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_at);

}  // namespace vgk
