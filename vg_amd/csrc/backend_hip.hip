// backend_hip.hip — HIP runtime plumbing + the gfx950 kernel entry points.
// There is deliberately no CPU fallback here: if no HIP device answers,
// make_backend() fails and vgk_create() returns VGK_ENODEV.
#include <hip/hip_runtime.h>
#include <string>
#include "backend.hpp"

namespace vgk {

// one DPP wave_shr:1 — lane l receives lane l-1's value (lane 0 receives 0)
static __device__ __forceinline__ uint32_t from_lane_above(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// Fill: 4 independent wavefronts per workgroup; each wavefront owns
// floor(64/G) read pairs for the whole skewed sweep.  No LDS, no barriers.
__global__ __launch_bounds__(256, 4) void gssw_fill_kernel(const GsswParams P) {
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= P.n_waves) return;
    const WaveDesc wd = P.waves[wave];
    Lane s;
    lane_init(s, P, wd, lane);
    asm volatile("" : "+v"(s.one));   // keep min(x,1) a packed min instead of cmp+cndmask
    uint32_t* tb = P.want_tb ? P.tb + (wd.tb_off + lane) * 4u : nullptr;
    for (uint32_t t = 0; t < wd.n_steps; ++t) {
        if ((t & 3u) == 0) lane_prefetch(s, P, t);
        const uint32_t rh = from_lane_above(s.out_h);
        const uint32_t rf = from_lane_above(s.out_f);
        const uint32_t ri = from_lane_above(s.info);
        lane_step(s, P, t, rh, rf, ri, tb ? tb + (size_t)t * 256u : nullptr);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t prob; unsigned long long key;
        if (lane_best(s, half, prob, key)) atomicMax(&P.best[prob], key);
    }
}

__global__ __launch_bounds__(256) void gssw_walk_kernel(const GsswParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_problems) walk_one(P, i);
}

class HipBackend final : public Backend {
public:
    int dev = 0; hipStream_t stream = nullptr; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipDeviceProp_t prop;
    float ms_fill = 0.f, ms_walk = 0.f; bool timed_walk = false, pending = false;
    ~HipBackend() override {
        hipSetDevice(dev);
        for (auto& e : ev) if (e) hipEventDestroy(e);
        if (stream) hipStreamDestroy(stream);
    }
    const char* name() const override { return prop.name; }
    int compute_units() const override { return prop.multiProcessorCount; }
    size_t memory_bytes() const override { return prop.totalGlobalMem; }
    void* alloc(size_t bytes) override {
        hipSetDevice(dev);
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        return p;
    }
    void release(void* p) override { if (p) { hipSetDevice(dev); hipFree(p); } }
    int upload(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int download(void* dst, const void* src, size_t bytes) override {
        hipSetDevice(dev);
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return VGK_ENODEV;
        return hipStreamSynchronize(stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int zero(void* dst, size_t bytes) override {
        hipSetDevice(dev);
        return hipMemsetAsync(dst, 0, bytes, stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int sync() override {
        hipSetDevice(dev);
        return hipStreamSynchronize(stream) == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    int run_gssw(const GsswParams& p, bool walk) override {
        hipSetDevice(dev);
        if (p.n_waves == 0) { ms_fill = ms_walk = 0; pending = false; return VGK_OK; }
        hipEventRecord(ev[0], stream);
        hipLaunchKernelGGL(gssw_fill_kernel, dim3((p.n_waves + 3) / 4), dim3(256), 0, stream, p);
        hipEventRecord(ev[1], stream);
        timed_walk = walk;
        if (walk) {
            hipLaunchKernelGGL(gssw_walk_kernel, dim3((p.n_problems + 255) / 256), dim3(256), 0, stream, p);
            hipEventRecord(ev[2], stream);
        }
        pending = true;
        return hipGetLastError() == hipSuccess ? VGK_OK : VGK_ENODEV;
    }
    double last_ms(int which) const override {
        HipBackend* self = const_cast<HipBackend*>(this);
        if (self->pending) {
            hipSetDevice(dev);
            hipStreamSynchronize(stream);
            hipEventElapsedTime(&self->ms_fill, ev[0], ev[1]);
            self->ms_walk = 0.f;
            if (timed_walk) hipEventElapsedTime(&self->ms_walk, ev[1], ev[2]);
            self->pending = false;
        }
        return which == 0 ? ms_fill : ms_walk;
    }
};

Backend* make_backend(int device, std::string& err) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { err = std::string("no HIP device: ") + hipGetErrorString(e); return nullptr; }
    if (device < 0 || device >= n) { err = "HIP device index out of range"; return nullptr; }
    auto* b = new HipBackend();
    b->dev = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&b->prop, device) != hipSuccess) {
        err = "cannot select HIP device"; delete b; return nullptr; }
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { err = "cannot create HIP stream"; delete b; return nullptr; }
    for (auto& ev : b->ev) if (hipEventCreate(&ev) != hipSuccess) { err = "cannot create HIP event"; delete b; return nullptr; }
    return b;
}

}  // namespace vgk
