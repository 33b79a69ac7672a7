// dgraph.hpp — the resident graph handle shared by window_api.cpp (which builds it from a caller's arrays) and tail_api.cpp (which
// builds it on the device from a tail forest).
#pragma once
#include <vector>
#include "ctx.hpp"

struct vgk_dgraph {
    vgk_ctx* ctx = nullptr;
    vgk::WinGraph g{};
    std::vector<void*> dev;          // device allocations (released with the graph)
    std::vector<uint64_t> dev_size;  // when as long as `dev`: the blocks came from the context's pool of device arenas and go back there
    uint64_t dev_bytes = 0;
    // far_prefix[v] = nodes below v with a predecessor that ends more than TB_JUMP columns before them (empty: unknown — a graph built on
    // the device); a window without such a node keeps its tracebacks near a diagonal (batch.hpp, default_tb_mode)
    std::vector<uint32_t> far_prefix;
};
