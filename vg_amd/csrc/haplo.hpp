// haplo.hpp — the haplotype index handle shared by gapless_api.cpp (which builds it) and wfa_api.cpp.
#pragma once
#include <vector>
#include "ctx.hpp"
#include "gapless_device.hpp"

struct vgk_haplo {
    vgk_ctx* ctx = nullptr;
    vgk::GIndex dev{};                  // device pointers
    std::vector<void*> held;
    uint32_t n_oriented = 0;
    std::vector<uint32_t> len;          // host copy, for validation
    // the index the gapless search walks when unary runs could be merged (gapless_device.hpp GMerge; owned: destroyed with this one), and the
    // tables that take seeds onto it and extension sets back
    vgk_haplo* merged = nullptr;
    vgk::GMerge merge{};
};

// What either index builder hands to vgk_haplo_from_tables (gapless_api.cpp): see there.
struct HaploTables { std::vector<uint32_t> count, body_off, body, edge_off, edge_base; std::vector<int32_t> edge_to; };
int vgk_haplo_strands(uint32_t n_nodes, const uint32_t* node_len, const char* fwd, std::vector<uint32_t>& len, std::vector<uint32_t>& seq_off, std::vector<char>& seq, uint64_t& total);
int vgk_haplo_from_tables(vgk_ctx* ctx, uint32_t n_oriented, const std::vector<uint32_t>& len, const std::vector<uint32_t>& seq_off, const std::vector<char>& seq, uint32_t total,
                          const HaploTables& T, vgk_haplo** out, bool merge_runs = true);
