// backend.hpp — the few device services the C-ABI layer needs.  The product
// implementation is backend_hip.hip (HIP runtime + the gfx950 kernels).  A
// second implementation exists only under tests/emu (CPU lock-step emulation of
// the same lane code) so the packing / kernel logic can be debugged without a GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>
#include "gssw_device.hpp"
#include "banded_device.hpp"
#include "banded_geom_device.hpp"
#include "gapless_device.hpp"
#include "wfa_device.hpp"
#include "wfa_wave_device.hpp"
#include "gssw_matrix_device.hpp"
#include "gssw_multi_device.hpp"
#include "banded_multi_device.hpp"
#include "gssw_pack_device.hpp"
#include "tail_device.hpp"
#include "minimizer_device.hpp"
#include "gssw_wide_device.hpp"
#include "rescue_requests_device.hpp"
#include "chain_device.hpp"

namespace vgk {

struct FillLaunch { uint32_t K, wave_begin, wave_count; };
struct BandedLaunch { uint32_t R, begin, count, lds_bytes; };     // problems order[begin, begin+count) with R band rows per lane;
                                                                  // lds_bytes = LDS staging area the largest of them needs (0: read from HBM)

class Backend {
public:
    virtual ~Backend() = default;
    virtual const char* name() const = 0;
    virtual int compute_units() const = 0;
    virtual size_t memory_bytes() const = 0;
    virtual int   device_index() const { return 0; }       // which device of the node this backend drives (two contexts on one device may share read-only tables)
    virtual void* alloc(size_t bytes) = 0;                 // device memory (nullptr on failure)
    virtual void  release(void* p) = 0;
    virtual void* host_alloc(size_t bytes) = 0;            // page-locked host staging memory (uninitialised; nullptr on failure)
    virtual void  host_release(void* p) = 0;
    virtual int   host_register(const void* p, size_t bytes) { (void)p; (void)bytes; return VGK_OK; }      // page-lock a caller's range (vgk_host_register)
    virtual int   host_unregister(const void* p) { (void)p; return VGK_OK; }
    virtual int   upload(void* dst, const void* src, size_t bytes) = 0;     // async on the stream
    virtual int   download(void* dst, const void* src, size_t bytes) = 0;   // synchronous
    // uploads that need not queue behind the kernels of another batch (vgk_gssw_pack while the previous batch runs): a copy
    // stream of their own; sync_side() waits for them only
    virtual int   upload_side(void* dst, const void* src, size_t bytes) { return upload(dst, src, bytes); }
    virtual int   sync_side() { return sync(); }
    virtual int   main_after_side() { return VGK_OK; }      // what the main stream is given from now on starts after everything the side stream holds so far (no host wait)
    virtual int   zero(void* dst, size_t bytes) = 0;                        // async on the stream
    // A batch's own completion event, and a third stream for the way back.  vgk_gssw_run records the batch's event behind its
    // kernels; vgk_gssw_fetch waits for THAT (polling: no blocking runtime call, no shared poll event), packs the CIGAR ops and
    // copies back on the fetch stream — so a caller may queue the next batch's kernels before it fetches this one and the GPU
    // never idles between batches.  (Backends without streams: everything is synchronous, the defaults do.)
    virtual void* event_create() { return nullptr; }
    virtual void  event_destroy(void* ev) { (void)ev; }
    virtual int   event_record(void* ev) { (void)ev; return VGK_OK; }               // on the main stream
    virtual int   event_wait(void* ev) { (void)ev; return sync(); }                 // host waits, polling
    virtual bool  event_done(void* ev) { (void)ev; return true; }                   // has everything before ev finished? (never waits; a synchronous backend: always)
    virtual int   fetch_after(void* ev) { (void)ev; return VGK_OK; }                // fetch-stream work queued from now on runs after ev
    virtual int   sync_fetch() { return sync(); }
    virtual int   download_fetch(void* dst, const void* src, size_t bytes) { return download(dst, src, bytes); }   // synchronous, fetch stream
    virtual int   download_fetch_async(void* dst, const void* src, size_t bytes) { return download(dst, src, bytes); }   // queued on the fetch stream (dst page-locked); sync_fetch waits
    // device-side packing of window problems (gssw_pack_device.hpp), asynchronous on the side (copy) stream like upload_side:
    // stage 1 = per-problem sizes + their prefix sums + totals, stage 2 = launch order, wavefronts, the arenas the kernels read.
    // win_tmp_bytes = device scratch both stages need (`tmp`); download_side / fill_side = synchronous copy back / async byte fill
    // on that stream.
    virtual size_t win_tmp_bytes(uint32_t n, uint32_t n_waves_cap) { (void)n; (void)n_waves_cap; return 16; }
    virtual int   win_stage1(const WinParams& P, void* tmp, size_t tmp_bytes) = 0;
    virtual int   win_stage2(const WinParams& P, void* tmp, size_t tmp_bytes) = 0;
    virtual int   download_side(void* dst, const void* src, size_t bytes) { return download(dst, src, bytes); }
    virtual int   fill_side(void* dst, int byte, size_t bytes) = 0;
    virtual int   sync() = 0;
    // gssw kernels: one fill launch per rows-per-lane instantiation (`launches`), then one traceback
    // launch over all reads; timings (ms, HIP events on the launch stream) of the last run
    virtual int   run_gssw(const GsswParams& p, const FillLaunch* launches, uint32_t n_launches, bool walk) = 0;
    // The same on launch lane `lane` (0 or 1), which also zeroes the batch's best-cell keys first and records `done` behind the
    // kernels.  Two lanes = two streams: consecutive batches of a streaming caller alternate between them, so the traceback of
    // batch k (bound by memory latency, few registers) runs under the fill of batch k + 1 (bound by VALU issue) and the fill's
    // last, partly empty round of wavefronts overlaps the next launch.  Backends with one stream ignore the lane.
    virtual int   run_gssw_on(int lane, const GsswParams& p, const FillLaunch* launches, uint32_t n_launches, bool walk, void* done) {
        (void)lane;
        int rc = zero(p.best, ((size_t)p.n_problems + 1) * sizeof(unsigned long long));
        if (!rc) rc = run_gssw(p, launches, n_launches, walk);
        if (!rc) rc = event_record(done);
        return rc;
    }
    virtual double last_ms_on(int lane, int which) const { (void)lane; return last_ms(which); }
    // Results on their way back (vgk_gssw_fetch): the CIGAR ops sit in per-problem slots of ops_per_problem entries, of which a
    // read uses a handful; they are packed behind each other on the device, in problem order, so that only what was written
    // crosses PCIe.  ops_offsets: offs[i] = position of problem i's first op inside its block of OPS_SCAN_BLOCK problems,
    // sums[b] = position of block b's first op, *total = all ops (synchronises).  ops_gather: out_res[i] = res[i] with
    // ops_begin rewritten (n_ops = 0 when the problem failed), out_ops = the packed ops (async on the stream).
    // A backend without them returns VGK_EUNSUPPORTED and the caller packs on the host.
    static constexpr uint32_t OPS_SCAN_BLOCK = 1024;
    virtual int   ops_offsets(const vgk_result*, uint32_t, uint32_t*, uint32_t*, uint64_t*) { return VGK_EUNSUPPORTED; }
    virtual int   ops_gather(const vgk_result*, const vgk_op*, uint32_t, const uint32_t*, const uint32_t*, vgk_result*, vgk_op*) { return VGK_EUNSUPPORTED; }
    virtual double last_ms(int which) const = 0;           // 0 = fill (all launches), 1 = traceback tail, 2 = number of fill launches,
                                                           // 3 / 4 = banded fill / banded traceback of the last run_banded
    // banded global alignment: one fill launch per rows-per-lane instantiation (one wavefront per problem), then one
    // traceback launch (one thread per problem) over p.n problems
    virtual int   run_banded(const BandedParams& p, const BandedLaunch* launches, uint32_t n_launches) = 0;
    // The same without waiting (the primary alignment only), bracketed by the timing events of `slot` (0 or 1): two sub-batches of a call in
    // flight.  banded_ms(slot, 0 | 1) = fill | traceback time, valid once the caller has waited for the launch.
    virtual int   run_banded_async(const BandedParams& p, const BandedLaunch* launches, uint32_t n_launches, int slot) { (void)slot; return run_banded(p, launches, n_launches); }
    virtual double banded_ms(int slot, int which) { (void)slot; return last_ms(which ? 4 : 3); }
    // the band geometry and kernel tables of p.n problems from their raw graph arrays, one lane per problem (banded_geom_device.hpp), on the
    // SIDE stream — behind upload_side()'s copies, beside whatever the main stream runs; download_side() of p.out waits for it
    virtual int   run_banded_geometry(const BGeomParams& p) = 0;
    // gapless extension: `threads` resident threads (one scratch slab each) stride over p.n reads; last_ms(5) = kernel ms
    virtual int   run_gapless(const GaplessParams& p, uint32_t threads) = 0;
    // the sets of a gapless batch in problem order (gapless_device.hpp): stage 1 = sizes per read, stage 2 = the gather (the prefix sums
    // in between are scan_u32's); mask_reads = ReadMasker over a byte range in place.  All asynchronous on the main stream.
    virtual int   gapless_order(const GOrderParams& p, int stage) = 0;
    virtual int   mask_reads(char* reads, size_t bytes) = 0;
    // clusters left on the device by the seeding: problem descriptors + hand-out keys (g_seeded_one), and a stable sort of (key, value)
    // pairs by the low `bits` bits of the key
    virtual int   gapless_seeded(const GSeededParams& p) = 0;
    virtual int   sort_pairs_u32(const uint32_t* key_in, uint32_t* key_out, const uint32_t* val_in, uint32_t* val_out, uint32_t n, int bits) = 0;
    // wavefront alignment: likewise, `threads` resident threads (one WScratch each, zeroed by the caller once) stride over
    // p.n problems; last_ms(6) = kernel ms
    virtual int   run_wfa(const WfaParams& p, uint32_t threads) = 0;
    virtual int   run_wfa_mask(const WProb* probs, const uint32_t* src_off, const char* raw, char* seqs, uint32_t n) = 0;     // wfa_mask_one for problems [0, n): asynchronous on the main stream
    virtual int   run_wfa_wave(const WwParams& p, uint32_t waves) = 0;    // one wavefront per problem (wfa_wave_device.hpp); the time adds to last_ms(6)
    // the hybrid's two kernels AT ONCE: the thread kernel on the main stream, the wavefront kernel beside it on a second one, polling the
    // hand-over list while it grows (a.producers_done / p.producers_done: wfa_wave_device.hpp); last_ms(6) = the pair's wall time on the
    // device.  A backend without a second stream says so and the caller launches them one after the other.
    virtual bool  wfa_concurrent() const { return false; }
    virtual int   run_wfa_hybrid(const WfaParams& p, uint32_t threads, const WwParams& a, uint32_t waves) { (void)p; (void)threads; (void)a; (void)waves; return VGK_EUNSUPPORTED; }
    virtual void  reset_wfa_ms() {}
    // pinned gssw fill that keeps H / E / F of every cell (k-best tracebacks): one thread per problem
    virtual int   run_gssw_matrix(const GsswMatrixParams& p) = 0;
    // the k-best tracebacks over those matrices, one lane per problem (gssw_multi_device.hpp); synchronises
    virtual int   run_gssw_multi(const GsswMultiParams& p) = 0;
    // the k-best banded alignments over the kept score matrices, one lane per problem (banded_multi_device.hpp); synchronises
    virtual int   run_banded_multi(const BandedMultiParams& q) = 0;
    // the wide route of vgk_gssw_align (gssw_wide_device.hpp): a workgroup of four wavefronts per problem, p.order[0 .. n8) with 8 rows
    // per lane, p.order[n8 .. n8 + n16) with 16; then the tracebacks, one lane per problem.  Asynchronous on the main stream.
    virtual int   run_gssw_wide(const WideParams& p, uint32_t n8, uint32_t n16) = 0;
    // X-drop with dozeu's band (vgk_xdrop_band_align): one wavefront per problem, the matrices stay for the host's traceback;
    // last_ms(7) = kernel ms
    virtual int   run_xdrop_band(const GsswMatrixParams& p) = 0;
    // The same without waiting: the kernels go onto the main stream, bracketed by the timing events of `slot` (0 or 1 — two sub-batches of a
    // call in flight: the host packs the next one while this one runs).  xdrop_band_ms(slot) is valid once the caller has waited for them
    // (an event recorded behind the launch, or sync()).  A backend without streams runs it on the spot.
    virtual int   run_xdrop_band_async(const GsswMatrixParams& p, int slot) { (void)slot; return run_xdrop_band(p); }
    virtual double xdrop_band_ms(int slot) { (void)slot; return last_ms(7); }
    // tail forests (tail_device.hpp), everything asynchronous on the main stream: one pass of the walks (p.pass; `threads` resident
    // lanes, one TScratch each, take the problems in turn); exclusive prefix sums of n 32-bit values (out[k] = in[0] + ... + in[k-1]);
    // the two per-node stages that turn the forest into the packer's tables; a byte fill; a stopwatch around all of it
    // (watch(0) ... watch(1), watch_ms() after a sync)
    virtual int   run_minimizer(const MinimizerParams& p) = 0;            // minimizer_device.hpp: one lane per read, pass p.pass
    virtual int   run_minimizer_list(const MzListParams& p) = 0;           // mz_list_one for reads [0, n] (pass p.pass), asynchronous on the main stream
    virtual int   run_minimizer_seeds_of(const MzSeedsOfParams& p) = 0;    // mz_seeds_of_one for minimizers [0, n]
    virtual int   run_tail(const TailParams& p, uint32_t threads) = 0;
    virtual int   run_tail_stage(const TStageParams& p, int what) = 0;     // one of the per-item stages of vgk_tail_stage (tail_device.hpp: TS_*)
    virtual int   run_rescue_requests(const RqParams& p, int what) = 0;    // one of the per-pair stages of vgk_rescue_requests (rescue_requests_device.hpp: RQ_*)
    virtual int   run_chain_stitch(const CsParams& p, int what) = 0;       // one of the stages of vgk_chain_stitch (chain_device.hpp: CS_*), asynchronous on the main stream
    virtual int   scan_u32(const uint32_t* in, uint32_t* out, uint32_t n) = 0;
    virtual int   forest_flags(const ForestParams& p) = 0;
    virtual int   forest_emit(const ForestParams& p) = 0;
    virtual int   fill(void* dst, int byte, size_t bytes) = 0;
    virtual void  watch(int which) { (void)which; }
    virtual double watch_ms() { return 0.0; }
};

// returns nullptr and sets err when the device cannot be used
Backend* make_backend(int device, std::string& err);

}  // namespace vgk
