#include "engine.hpp"
#include <dlfcn.h>
#include <cstdlib>
#include <stdexcept>

namespace vgamd {

EngineApi::~EngineApi() { /* keep the library mapped: contexts may outlive us */ }

static std::string default_engine_path() {
    if (const char* e = std::getenv("VGAMD_ENGINE_LIB")) return e;
    Dl_info info;
    if (dladdr((void*)&default_engine_path, &info) && info.dli_fname) {
        std::string self = info.dli_fname;
        size_t slash = self.rfind('/');
        std::string dir = slash == std::string::npos ? "." : self.substr(0, slash);
        return dir + "/libvgamd.so";
    }
    return "libvgamd.so";
}

template <class F> static void bind(void* dl, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(dl, name));
    if (!fn) throw std::runtime_error(std::string("vgamd engine: missing symbol ") + name);
}

std::shared_ptr<EngineApi> load_engine(const std::string& path) {
    std::string p = path.empty() ? default_engine_path() : path;
    void* dl = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) throw std::runtime_error("vgamd engine: cannot load " + p + ": " + dlerror() +
                                      " (the HIP engine is required; there is no CPU fallback)");
    auto api = std::make_shared<EngineApi>();
    api->dl = dl;
    bind(dl, "vgk_abi_version", api->abi_version);
    bind(dl, "vgk_strerror", api->strerror);
    bind(dl, "vgk_create", api->create);
    bind(dl, "vgk_create_qual_adj", api->create_qual_adj);
    bind(dl, "vgk_destroy", api->destroy);
    bind(dl, "vgk_gssw_align", api->gssw_align);
    bind(dl, "vgk_gssw_align_multi", api->gssw_align_multi);
    bind(dl, "vgk_gssw_pack", api->gssw_pack);
    bind(dl, "vgk_gssw_run", api->gssw_run);
    bind(dl, "vgk_gssw_fetch", api->gssw_fetch);
    bind(dl, "vgk_batch_free", api->batch_free);
    bind(dl, "vgk_batch_kernel_ms", api->batch_kernel_ms);
    bind(dl, "vgk_batch_alg_bytes", api->batch_alg_bytes);
    bind(dl, "vgk_batch_cells", api->batch_cells);
    bind(dl, "vgk_banded_align", api->banded_align);
    bind(dl, "vgk_banded_align_multi", api->banded_align_multi);
    bind(dl, "vgk_haplo_create", api->haplo_create);
    bind(dl, "vgk_haplo_destroy", api->haplo_destroy);
    bind(dl, "vgk_gapless_extend", api->gapless_extend);
    bind(dl, "vgk_wfa_extend", api->wfa_extend);
    bind(dl, "vgk_xdrop_band_align", api->xdrop_band_align);
    bind(dl, "vgk_gssw_pack_windows", api->gssw_pack_windows);
    bind(dl, "vgk_gssw_pack_extensions", api->gssw_pack_extensions);
    bind(dl, "vgk_graph_create", api->graph_create);
    bind(dl, "vgk_graph_destroy", api->graph_destroy);
    bind(dl, "vgk_tail_forest", api->tail_forest);
    bind(dl, "vgk_forest_fetch", api->forest_fetch);
    bind(dl, "vgk_forest_graph", api->forest_graph);
    bind(dl, "vgk_forest_size", api->forest_size);
    bind(dl, "vgk_forest_destroy", api->forest_destroy);
    bind(dl, "vgk_wfa_set_point_budgets", api->wfa_set_point_budgets);
    bind(dl, "vgk_wfa_last_ms", api->wfa_last_ms);
    bind(dl, "vgk_wfa_last_wave", api->wfa_last_wave);
    bind(dl, "vgk_wfa_set_form", api->wfa_set_form);
    bind(dl, "vgk_wfa_get_form", api->wfa_get_form);
    bind(dl, "vgk_wfa_set_cost_hints", api->wfa_set_cost_hints);
    bind(dl, "vgk_chain_stitch", api->chain_stitch);
    bind(dl, "vgk_host_register", api->host_register);
    bind(dl, "vgk_host_unregister", api->host_unregister);
    bind(dl, "vgk_chain_stitch_last_ms", api->chain_stitch_last_ms);
    if (api->abi_version() != VGK_ABI_VERSION) throw std::runtime_error("vgamd engine: ABI version mismatch in " + p);
    return api;
}

}  // namespace vgamd
