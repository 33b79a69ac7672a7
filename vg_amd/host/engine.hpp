// engine.hpp — binds the C ABI of include/vgk.h at run time.
//
// The product binds vg_amd/libvgamd.so (HIP kernels).  There is NO CPU fallback:
// if the HIP library cannot be loaded or no gfx950 device answers, construction
// throws and the caller fails loudly.  Tests may name another library exporting
// the same ABI (oracle/libvgoracle.so) explicitly — that is the only way the
// oracle is ever reached, and nothing in this directory names it.
#pragma once
#include <memory>
#include <string>
#include "../../include/vgk.h"

namespace vgamd {

struct EngineApi {
    void* dl = nullptr;
    decltype(&vgk_abi_version) abi_version = nullptr;
    decltype(&vgk_strerror) strerror = nullptr;
    decltype(&vgk_create) create = nullptr;
    decltype(&vgk_create_qual_adj) create_qual_adj = nullptr;
    decltype(&vgk_destroy) destroy = nullptr;
    decltype(&vgk_gssw_align) gssw_align = nullptr;
    decltype(&vgk_gssw_align_multi) gssw_align_multi = nullptr;
    decltype(&vgk_gssw_pack) gssw_pack = nullptr;
    decltype(&vgk_gssw_run) gssw_run = nullptr;
    decltype(&vgk_gssw_fetch) gssw_fetch = nullptr;
    decltype(&vgk_batch_free) batch_free = nullptr;
    decltype(&vgk_batch_kernel_ms) batch_kernel_ms = nullptr;
    decltype(&vgk_batch_alg_bytes) batch_alg_bytes = nullptr;
    decltype(&vgk_batch_cells) batch_cells = nullptr;
    decltype(&vgk_banded_align) banded_align = nullptr;
    decltype(&vgk_banded_align_multi) banded_align_multi = nullptr;
    decltype(&vgk_haplo_create) haplo_create = nullptr;
    decltype(&vgk_haplo_destroy) haplo_destroy = nullptr;
    decltype(&vgk_gapless_extend) gapless_extend = nullptr;
    decltype(&vgk_wfa_extend) wfa_extend = nullptr;
    decltype(&vgk_xdrop_band_align) xdrop_band_align = nullptr;
    decltype(&vgk_gssw_pack_windows) gssw_pack_windows = nullptr;
    decltype(&vgk_gssw_pack_extensions) gssw_pack_extensions = nullptr;
    decltype(&vgk_graph_create) graph_create = nullptr;
    decltype(&vgk_graph_destroy) graph_destroy = nullptr;
    decltype(&vgk_tail_forest) tail_forest = nullptr;
    decltype(&vgk_forest_fetch) forest_fetch = nullptr;
    decltype(&vgk_forest_graph) forest_graph = nullptr;
    decltype(&vgk_forest_size) forest_size = nullptr;
    decltype(&vgk_forest_destroy) forest_destroy = nullptr;
    decltype(&vgk_wfa_set_point_budgets) wfa_set_point_budgets = nullptr;
    decltype(&vgk_wfa_last_ms) wfa_last_ms = nullptr;
    decltype(&vgk_wfa_last_wave) wfa_last_wave = nullptr;
    decltype(&vgk_wfa_set_form) wfa_set_form = nullptr;
    decltype(&vgk_wfa_get_form) wfa_get_form = nullptr;
    decltype(&vgk_wfa_set_cost_hints) wfa_set_cost_hints = nullptr;
    decltype(&vgk_chain_stitch) chain_stitch = nullptr;
    decltype(&vgk_host_register) host_register = nullptr;
    decltype(&vgk_host_unregister) host_unregister = nullptr;
    decltype(&vgk_chain_stitch_last_ms) chain_stitch_last_ms = nullptr;
    ~EngineApi();
};

// Load the ABI from `path`; empty path = the product library next to this shim
// (libvgamd.so in the directory of libvgamd_host.so), or $VGAMD_ENGINE_LIB.
std::shared_ptr<EngineApi> load_engine(const std::string& path = "");

}  // namespace vgamd
