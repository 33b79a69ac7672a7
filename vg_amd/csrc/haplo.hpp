// haplo.hpp — the haplotype index handle shared by gapless_api.cpp (which builds it) and wfa_api.cpp.
#pragma once
#include <vector>
#include "ctx.hpp"

struct vgk_haplo {
    vgk_ctx* ctx = nullptr;
    vgk::GIndex dev{};                  // device pointers
    std::vector<void*> held;
    uint32_t n_oriented = 0;
    std::vector<uint32_t> len;          // host copy, for validation
};
