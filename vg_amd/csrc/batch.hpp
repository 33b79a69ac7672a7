// batch.hpp — a packed gssw batch resident in HBM, shared by the translation units that build one (vgk_api.cpp: per-problem
// graphs packed on the host; window_api.cpp: windows of a resident graph packed on the device).
#pragma once
#include <cstdlib>
#include <vector>
#include "backend.hpp"
#include "ctx.hpp"

using namespace vgk;

struct vgk_batch {
    vgk_ctx* ctx = nullptr;
    uint32_t n = 0;
    bool want_tb = false, ran = false;
    bool probs_displaced = false;                     // a speculative run has rewritten descriptors of this batch (GsswParams::restore_probs)
    bool spec_probe = false;                           // the last run was a probe of the context's SpecPolicy (ctx.hpp)
    bool ran_spec = false, spec_observed = true;      // the last run speculated | its miss count has been handed to the context's SpecPolicy
    GsswParams P{};
    std::vector<vgk_ctx::Pooled> dev;   // every device allocation of this batch (back to the context's pool when the batch is freed)
    uint64_t cells = 0, tb_cells = 0, in_bytes = 0, dev_bytes = 0, alg_bytes = 0;
    uint64_t ops_total = 0;
    uint64_t wave_steps = 0;            // sum over wavefronts of their fill steps
    ProbDesc* probs = nullptr; uint64_t probs_bytes = 0;   // kept for fetch(): a page-locked block from the context's pool, back to it with the batch
    int lane = 0;                       // launch lane (stream) of this batch: consecutive batches of a context alternate (Backend::run_gssw_on)
    void* done = nullptr;               // recorded behind this batch's kernels by vgk_gssw_run (Backend::event_*)
    ~vgk_batch() { if (probs && ctx) ctx->host_give(probs, probs_bytes); if (done && ctx) ctx->be->event_destroy(done); }
    std::vector<FillLaunch> launches;   // one per length bucket
    // extension windows (vgk_gssw_pack_extensions): problem i's node k is window node ext_nodes[ext_off[i] + k]; ext_count[i] = WIN_EXT_DUMMY: nothing
    // lay in the extension's direction.  vgk_gssw_fetch hands results and ops back in the window's terms.
    std::vector<uint32_t> ext_count, ext_off, ext_nodes;
    struct Upload { void* dst; const void* src; size_t bytes; };
    std::vector<Upload> uploads;        // queued by to_device under the context lock, issued by vgk_gssw_pack outside it
};

inline int nt_read(char ch) {   // gssw_create_nt_table: case-insensitive ACGT, else N
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                  case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline int nt_ref(char ch) {    // after nonATGCNtoN (src/aligner.cpp:39): upper-case ACGT only
    switch (ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

// Which traceback a batch runs (gssw_device.hpp).  TB_CODES — the fill stores a 4-bit code per cell — is the default.  Measured on the MI355X
// (profiles/r04), per million 150 bp reads: the recomputing form TB_REWALK takes the fill from 19.3 to 14.8-15.6 ms, but then
// spends 5.2 ms recomputing the band (1.1 of it finding end cells and loading checkpoints), 4.3 ms walking it — no less than the walk over
// stored codes, which is bound by the latency of its scattered reads either way — and 1.3 ms on the handful of reads that leave their
// band (one on-demand walk's own latency): 26.4 ms against 24.3.  It is kept, exact and tested, behind VGAMD_TB_REWALK=1.
// Its band lies along the diagonal through the end cell in COLUMN space: it serves a path as long as walking back over a node boundary
// moves it at most a few columns beyond the previous node's end — chains, SNP bubbles, short insertions; a predecessor that ends more than
// TB_JUMP columns before its successor starts (the far side of a long bubble, a sibling subtree of a tail forest) throws the path onto the slow
// on-demand form — `near_chain` says the batch has no such edge (the packers work it out; it would gate the mode if it became the default).
constexpr uint32_t TB_JUMP = TB_SLACK / 2;
inline int32_t default_tb_mode(int fused, bool near_chain) {
    (void)near_chain;
    if (TB_TILE > 1 || fused) return TB_CODES;
    if (const char* e = std::getenv("VGAMD_TB_CODES")) if (std::atoi(e)) return TB_CODES;
    if (const char* e = std::getenv("VGAMD_TB_REWALK")) if (std::atoi(e)) return TB_REWALK;
    return TB_CODES;
}

template <class T>
inline int to_device(vgk_batch* b, const std::vector<T>& v, const T*& out, size_t extra = 0) {
    const size_t bytes = (v.size() + extra) * sizeof(T);
    uint64_t got = 0;
    void* p = b->ctx->dev_take(bytes, got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += bytes;
    if (!v.empty()) b->uploads.push_back({p, v.data(), v.size() * sizeof(T)});
    out = (const T*)p;
    return VGK_OK;
}

template <class T>
inline int to_device(vgk_batch* b, const T* v, size_t count, const T*& out, size_t extra = 0) {      // from a staging arena
    const size_t bytes = (count + extra) * sizeof(T);
    uint64_t got = 0;
    void* p = b->ctx->dev_take(bytes, got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += bytes;
    if (count) b->uploads.push_back({p, v, count * sizeof(T)});
    out = (const T*)p;
    return VGK_OK;
}
template <class T>
inline int dev_alloc(vgk_batch* b, size_t count, T*& out) {
    uint64_t got = 0;
    void* p = b->ctx->dev_take(count * sizeof(T), got);
    if (!p) return VGK_ENOMEM;
    b->dev.push_back({p, got}); b->dev_bytes += count * sizeof(T);
    out = (T*)p;
    return VGK_OK;
}

