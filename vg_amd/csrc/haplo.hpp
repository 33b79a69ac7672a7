// haplo.hpp — the haplotype index handle shared by gapless_api.cpp (which builds it) and wfa_api.cpp.
#pragma once
#include <memory>
#include <mutex>
#include <vector>
#include "ctx.hpp"
#include "gapless_device.hpp"

struct vgk_haplo {
    vgk_ctx* ctx = nullptr;
    vgk::GIndex dev{};                  // device pointers
    std::vector<void*> held;
    uint32_t n_oriented = 0;
    std::vector<uint32_t> len;          // host copy, for validation
    // the same index with its unary runs merged (gapless_device.hpp GMerge; owned: destroyed with this one), and the tables that take positions
    // onto it and paths back: what the WFA wavefront kernel walks (wfa_wave_device.hpp), and the gapless search when asked to
    vgk_haplo* merged = nullptr;
    vgk::GMerge merge{};
    // The merged form is built on the first call that walks it (vgk_wfa_extend; at index build when the gapless search is to walk it,
    // VGAMD_HAPLO_MERGE=1): a caller that only ever extends seeds gaplessly pays neither its HBM nor its build.  Until then the tables it is
    // made from wait here (host memory, released once it is built).
    struct PendingMerge;
    std::shared_ptr<PendingMerge> pending_merge; std::mutex merge_mu;
    bool search_merged = false;         // the gapless search walks `merged` (VGAMD_HAPLO_MERGE=1: measured, it does not pay there); the WFA wavefront kernel always does
};

// What either index builder hands to vgk_haplo_from_tables (gapless_api.cpp): see there.
struct HaploTables { std::vector<uint32_t> count, body_off, body, edge_off, edge_base; std::vector<int32_t> edge_to; };
// the merged form, now (a no-op once it exists or when nothing is pending); -> VGK_OK or the build's error
int vgk_haplo_ensure_merged(vgk_haplo* h);
int vgk_haplo_strands(uint32_t n_nodes, const uint32_t* node_len, const char* fwd, std::vector<uint32_t>& len, std::vector<uint32_t>& seq_off, std::vector<char>& seq, uint64_t& total);
int vgk_haplo_from_tables(vgk_ctx* ctx, uint32_t n_oriented, const std::vector<uint32_t>& len, const std::vector<uint32_t>& seq_off, const std::vector<char>& seq, uint32_t total,
                          const HaploTables& T, vgk_haplo** out, bool merge_runs = true);
