// chain_api.cpp — vgk_chain_stitch: the host half of the per-read Path composition (chain_device.hpp).
//
// Per call: the caller's pieces and the arrays its ALIGNMENT / PATH pieces point into go up (page-locked staging, one copy each); the LINK
// pieces point into what the context's last vgk_wfa_extend call left in HBM (vgk_ctx::wfa_out) — nothing of that comes down.  Three
// kernels around two prefix sums (per-read bounds -> stretches of the work arrays -> exact sizes -> dense output); the results, the
// mappings and the edit runs of the whole batch come back in three copies.
#include <algorithm>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "haplo.hpp"
#include "chain_device.hpp"

using namespace vgk;

namespace {
struct ChainHost { PinnedBuf<char> up; PinnedBuf<vgk_chain_result> res; PinnedBuf<uint32_t> tails; };
}

extern "C" {

double vgk_chain_stitch_last_ms(vgk_ctx* ctx) { return ctx ? ctx->chain_stitch_ms : 0.0; }

int vgk_chain_stitch(vgk_ctx* ctx, const vgk_haplo* index, const vgk_chain_piece* pieces, const uint64_t* piece_off, uint32_t n_reads,
                     const uint32_t* nodes, size_t n_nodes, const vgk_chain_mapping* mappings, size_t n_mappings, const uint32_t* edits, size_t n_edits,
                     vgk_chain_result* results, vgk_chain_mapping* out_mappings, size_t mapping_cap, uint32_t* out_edits, size_t edit_cap, size_t written[2]) try {
    if (written) written[0] = written[1] = 0;
    if (!ctx || !index || !vgk_tables_usable(index->ctx, ctx) || !piece_off || (!results && n_reads) || (!nodes && n_nodes) || (!mappings && n_mappings) || (!edits && n_edits)
        || (!out_mappings && mapping_cap) || (!out_edits && edit_cap)) return VGK_EINVAL;
    if (!n_reads) return VGK_OK;
    for (uint32_t r = 0; r < n_reads; ++r) if (piece_off[r + 1] < piece_off[r]) return VGK_EINVAL;
    const uint64_t n_pieces = piece_off[n_reads];
    if (n_pieces && !pieces) return VGK_EINVAL;
    if (n_nodes > 0xfffffff0ull || n_mappings > 0xfffffff0ull || n_edits > 0xfffffff0ull) return VGK_ETOOBIG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    Backend* be = ctx->be.get();
    if (!ctx->chain_host) ctx->chain_host = std::make_shared<ChainHost>();
    ChainHost& H = *static_cast<ChainHost*>(ctx->chain_host.get());
    const uint32_t R = n_reads + 1;

    CsParams P{};
    P.index = index->dev; P.n_reads = n_reads; P.n_nodes = n_nodes; P.n_mappings = n_mappings; P.n_edits = n_edits;
    const bool links = ctx->wfa_out.valid && ctx->wfa_out.index == (const void*)index;
    P.link_res = links ? ctx->wfa_out.res : nullptr; P.link_paths = links ? ctx->wfa_out.paths : nullptr; P.link_edits = links ? ctx->wfa_out.edits : nullptr;
    P.n_links = links ? ctx->wfa_out.n : 0; P.link_path_cap = links ? ctx->wfa_out.path_cap : 0; P.link_edit_cap = links ? ctx->wfa_out.edit_cap : 0;

    // one staging block, one upload: piece offsets | pieces | nodes | mappings | edits (each 16-byte aligned)
    auto al = [](uint64_t b) { return (b + 15) & ~15ull; };
    const uint64_t b_off = al(sizeof(uint64_t) * (uint64_t)R), b_pc = al(sizeof(vgk_chain_piece) * n_pieces), b_nd = al(sizeof(uint32_t) * n_nodes),
                   b_mp = al(sizeof(vgk_chain_mapping) * n_mappings), b_ed = al(sizeof(uint32_t) * n_edits);
    const uint64_t up_bytes = b_off + b_pc + b_nd + b_mp + b_ed + 16;
    char* up = H.up.get(be, up_bytes);
    if (!up) return VGK_ENOMEM;
    std::memcpy(up, piece_off, sizeof(uint64_t) * (size_t)R);
    if (n_pieces) std::memcpy(up + b_off, pieces, sizeof(vgk_chain_piece) * n_pieces);
    if (n_nodes) std::memcpy(up + b_off + b_pc, nodes, sizeof(uint32_t) * n_nodes);
    if (n_mappings) std::memcpy(up + b_off + b_pc + b_nd, mappings, sizeof(vgk_chain_mapping) * n_mappings);
    if (n_edits) std::memcpy(up + b_off + b_pc + b_nd + b_mp, edits, sizeof(uint32_t) * n_edits);
    char* d_up = (char*)ctx->ensure_scratch(140, up_bytes);
    uint32_t* d_tab = (uint32_t*)ctx->ensure_scratch(141, sizeof(uint32_t) * 8 * (uint64_t)R);      // bound | slot | count | out_slot, two halves each
    vgk_chain_result* d_res = (vgk_chain_result*)ctx->ensure_scratch(142, sizeof(vgk_chain_result) * 2 * (uint64_t)n_reads);
    if (!d_up || !d_tab || !d_res) return VGK_ENOMEM;
    int rc;
    be->watch(0);
    if ((rc = be->upload(d_up, up, up_bytes))) return rc;
    P.piece_off = (const uint64_t*)d_up; P.pieces = (const vgk_chain_piece*)(d_up + b_off); P.nodes = (const uint32_t*)(d_up + b_off + b_pc);
    P.mappings = (const vgk_chain_mapping*)(d_up + b_off + b_pc + b_nd); P.edits = (const uint32_t*)(d_up + b_off + b_pc + b_nd + b_mp);
    P.bound = d_tab; P.slot = d_tab + 2 * (size_t)R; P.count = d_tab + 4 * (size_t)R; P.out_slot = d_tab + 6 * (size_t)R;
    P.res = d_res; P.out_res = d_res + n_reads;
    // 1. bounds per read (entry n_reads of either half: 0), their prefix sums, the totals
    if ((rc = be->run_chain_stitch(P, CS_BOUND))) return rc;
    if ((rc = be->scan_u32(P.bound, d_tab + 2 * (size_t)R, R))) return rc;
    if ((rc = be->scan_u32(P.bound + R, d_tab + 3 * (size_t)R, R))) return rc;
    uint32_t* tails = H.tails.get(be, 8);
    if (!tails) return VGK_ENOMEM;
    if ((rc = be->download(&tails[0], d_tab + 2 * (size_t)R + n_reads, sizeof(uint32_t)))) return rc;      // (the scans' last entries: all mappings | all edit runs at most)
    if ((rc = be->download(&tails[1], d_tab + 3 * (size_t)R + n_reads, sizeof(uint32_t)))) return rc;
    // (a 32-bit prefix sum that wrapped: the sum of per-read bounds each < 2^32 — caught by comparing with what the inputs can make at most)
    const uint64_t most_m = n_mappings + (uint64_t)n_nodes + n_pieces + (links ? ctx->wfa_out.path_cap + ctx->wfa_out.n : 0);
    const uint64_t most_e = n_edits + (uint64_t)n_nodes + n_pieces + (links ? ctx->wfa_out.path_cap + ctx->wfa_out.edit_cap : 0);
    if (most_m > 0xfffffff0ull || most_e > 0xfffffff0ull) return VGK_ETOOBIG;
    const uint64_t work_m = tails[0], work_e = tails[1];
    P.work_m = (vgk_chain_mapping*)ctx->ensure_scratch(143, sizeof(vgk_chain_mapping) * (work_m + 1));
    P.work_e = (uint32_t*)ctx->ensure_scratch(144, sizeof(uint32_t) * (work_e + 1));
    if (!P.work_m || !P.work_e) return VGK_ENOMEM;
    // 2. the composition; exact sizes and their prefix sums
    if ((rc = be->run_chain_stitch(P, CS_STITCH))) return rc;
    if ((rc = be->scan_u32(P.count, d_tab + 6 * (size_t)R, R))) return rc;
    if ((rc = be->scan_u32(P.count + R, d_tab + 7 * (size_t)R, R))) return rc;
    if ((rc = be->download(&tails[2], d_tab + 6 * (size_t)R + n_reads, sizeof(uint32_t)))) return rc;
    if ((rc = be->download(&tails[3], d_tab + 7 * (size_t)R + n_reads, sizeof(uint32_t)))) return rc;
    const uint64_t total_m = tails[2], total_e = tails[3];
    if (written) { written[0] = total_m; written[1] = total_e; }
    // 3. dense output in read order, as much of it as the caller has room for
    P.out_m_cap = std::min<uint64_t>(mapping_cap, total_m); P.out_e_cap = std::min<uint64_t>(edit_cap, total_e);
    P.out_m = (vgk_chain_mapping*)ctx->ensure_scratch(145, sizeof(vgk_chain_mapping) * (P.out_m_cap + 1));
    P.out_e = (uint32_t*)ctx->ensure_scratch(146, sizeof(uint32_t) * (P.out_e_cap + 1));
    if (!P.out_m || !P.out_e) return VGK_ENOMEM;
    if ((rc = be->run_chain_stitch(P, CS_GATHER))) return rc;
    be->watch(1);
    vgk_chain_result* hres = H.res.get(be, n_reads);
    if (!hres) return VGK_ENOMEM;
    if ((rc = be->download(hres, P.out_res, sizeof(vgk_chain_result) * (size_t)n_reads))) return rc;
    ctx->chain_stitch_ms = be->watch_ms();
    std::memcpy(results, hres, sizeof(vgk_chain_result) * (size_t)n_reads);
    // (the caller's arrays are not page-locked as a rule: the runtime stages them; a caller that wants full-rate DMA registers them — vgk_host_register)
    if (P.out_m_cap && (rc = be->download(out_mappings, P.out_m, sizeof(vgk_chain_mapping) * P.out_m_cap))) return rc;
    if (P.out_e_cap && (rc = be->download(out_edits, P.out_e, sizeof(uint32_t) * P.out_e_cap))) return rc;
    return total_m > mapping_cap || total_e > edit_cap ? VGK_EOPS : VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

}  // extern "C"
