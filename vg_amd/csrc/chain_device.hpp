// chain_device.hpp — one Path per read out of the pieces of its chain (vgk_chain_stitch): one lane per read walks the read's pieces in read
// order, turns each into mappings as WFAAlignment::to_path does (reference src/gbwt_extender.cpp:954-1070) — or takes a stated Path's mappings
// as they are —, and feeds them, one mapping at a time, through the rules of `simplify(path, false)` (src/path.cpp:1314-1497; Mapping simplify
// :1509-1563; concat_mappings :1499-1507), which only ever look at the mapping they are given and at the LAST mapping kept so far.
//
// MI355X-first: the reference composes a protobuf Path per read on the mapping thread (append_path + simplify copy every Mapping two or three
// times); here the WFA results of a whole batch stay in HBM as flat node paths and edit runs, a lane per read writes flat mappings and edit
// runs into the read's own stretch of a work array — the LAST mapping's edits are always the tail of that stretch, so "append to the last
// mapping", "join with the last mapping" and "merge its runs again" are all in place —, and a gather packs the reads behind each other, so
// that only the composed alignments cross PCIe.  HBM-bound byte work: ~12 B read and ~12 B written per mapping.
//
// Three stages around two prefix sums (Backend::scan_u32):  CS_BOUND  per read, how many mappings / edit runs its pieces can make at most;
// CS_STITCH  the composition into the stretch those bounds reserve, exact sizes out;  CS_GATHER  dense copy in read order.
#pragma once
#include <cstdint>
#include "../../include/vgk.h"
#include "gapless_device.hpp"

namespace vgk {

struct CsParams {
    GIndex index;                                               // node lengths (g_len)
    const vgk_chain_piece* pieces; const uint64_t* piece_off; uint32_t n_reads;
    const uint32_t* nodes; const vgk_chain_mapping* mappings; const uint32_t* edits;       // the caller's arrays, in HBM
    uint64_t n_nodes, n_mappings, n_edits;
    // what the last vgk_wfa_extend call left in HBM (results in problem order; paths and edit runs in completion order)
    const vgk_wfa_result* link_res; const uint32_t* link_paths; const uint32_t* link_edits; uint32_t n_links; uint64_t link_path_cap, link_edit_cap;
    uint32_t* bound;                                            // [2 (n_reads + 1)] CS_BOUND: mappings | edit runs per read at most (entry n_reads of either half = 0)
    const uint32_t* slot;                                       // [2 (n_reads + 1)] their exclusive prefix sums, half by half
    vgk_chain_mapping* work_m; uint32_t* work_e;                // the stretches
    vgk_chain_result* res;                                      // [n_reads] CS_STITCH: status, sizes; mapping_begin / edit_begin inside the work arrays
    uint32_t* count;                                            // [2 (n_reads + 1)] exact sizes
    const uint32_t* out_slot;                                   // [2 (n_reads + 1)] their exclusive prefix sums
    vgk_chain_mapping* out_m; uint32_t* out_e; vgk_chain_result* out_res;     // CS_GATHER: dense, in read order
    uint64_t out_m_cap, out_e_cap;
};
enum { CS_BOUND = 0, CS_STITCH = 1, CS_GATHER = 2 };

VGK_HD uint32_t cs_kind(uint32_t run) { return run & 3u; }
VGK_HD uint32_t cs_len(uint32_t run) { return run >> 2; }
VGK_HD bool cs_link_ok(const CsParams& P, const vgk_chain_piece& pc) {
    if (pc.link >= P.n_links) return false;
    const vgk_wfa_result& r = P.link_res[pc.link];
    return r.status == VGK_OK && r.ok != 0 && (uint64_t)r.path_begin + r.path_len <= P.link_path_cap && (uint64_t)r.edit_begin + r.n_edits <= P.link_edit_cap;
}

// ---- CS_BOUND --------------------------------------------------------------------------------------------------------------------------
// A WFAAlignment of p nodes and e runs makes at most max(p, 1) mappings and e + p edits (a run is cut at most once per node boundary);
// a stated Path makes what it holds.  Nothing in simplify makes more of either.
VGK_HD void cs_bound_one(const CsParams& P, uint32_t r) {
    uint64_t m = 0, e = 0;
    if (r < P.n_reads) {
        for (uint64_t k = P.piece_off[r]; k < P.piece_off[r + 1]; ++k) {
            const vgk_chain_piece pc = P.pieces[k];
            if (pc.kind == (uint32_t)VGK_PIECE_LINK) {
                if (!cs_link_ok(P, pc)) continue;
                const vgk_wfa_result& w = P.link_res[pc.link];
                m += w.path_len ? w.path_len : 1u; e += (uint64_t)w.n_edits + w.path_len;
            } else if (pc.kind == (uint32_t)VGK_PIECE_ALIGNMENT) {
                m += pc.path_len ? pc.path_len : 1u; e += (uint64_t)pc.n_edits + pc.path_len;
            } else if (pc.kind == (uint32_t)VGK_PIECE_PATH) {
                if ((uint64_t)pc.path_begin + pc.path_len > P.n_mappings) continue;
                m += pc.path_len;
                for (uint32_t q = 0; q < pc.path_len; ++q) e += P.mappings[pc.path_begin + q].n_edits;
            }
        }
    }
    P.bound[r] = m < 0xffffffffull ? (uint32_t)m : 0xffffffffu;
    P.bound[(size_t)P.n_reads + 1 + r] = e < 0xffffffffull ? (uint32_t)e : 0xffffffffu;
}

// ---- CS_STITCH -------------------------------------------------------------------------------------------------------------------------
// The composed path so far: mappings M[0 .. nm), the edits of M[k] at E[M[k].edit_begin .. + n_edits) (offsets inside the read's stretch),
// behind each other without gaps, the last mapping's edits ending at `tail`.
// A lane's time is its chain of dependent memory operations (a load that follows a store to the same stretch waits ~1 us at 125 wavefronts per
// launch), so what the rules look at lives in registers: the LAST mapping kept (l_*: node, offset, its runs' place, its from_length, its last run —
// M[nm - 1] itself is written when the next mapping is pushed, and at the end) and the mapping in the making (c_*: its last run is written only when
// the next run begins or the mapping closes).  Per run one store; per mapping the inputs' loads, a store or two, no load of anything written here.
struct CsState {
    vgk_chain_mapping* M; uint32_t* E; uint32_t nm, tail, cap_m, cap_e;
    uint32_t l_node, l_offset, l_begin, l_n, l_from, l_last;
    bool l_mixed;                 // an insertion was appended behind an insertion: two runs of one kind side by side until something joins (:1349: appended, not merged)
    uint32_t c_node, c_offset, c_begin, c_end, c_last, c_first, c_from;
    uint32_t total_to, total_from;
    int32_t status;
};
VGK_HD uint32_t cs_from_length(const uint32_t* E, uint32_t b, uint32_t n) { uint32_t f = 0; for (uint32_t k = 0; k < n; ++k) if (cs_kind(E[b + k]) != (uint32_t)VGK_WFA_INSERTION) f += cs_len(E[b + k]); return f; }
VGK_HD uint32_t cs_to_length(const uint32_t* E, uint32_t b, uint32_t n) { uint32_t t = 0; for (uint32_t k = 0; k < n; ++k) if (cs_kind(E[b + k]) != (uint32_t)VGK_WFA_DELETION) t += cs_len(E[b + k]); return t; }
// Mapping simplify (:1509-1563, trim_internal_deletions = false) over E[b, b + n): runs of one kind become one; -> the new n
VGK_HD uint32_t cs_merge_runs(uint32_t* E, uint32_t b, uint32_t n) {
    if (!n) return 0;
    uint32_t w = b;
    for (uint32_t k = b + 1; k < b + n; ++k) {
        if (cs_len(E[k]) == 0) continue;                                       // edit_is_empty(f): skipped (:1545)
        if (cs_kind(E[k]) == cs_kind(E[w])) E[w] += cs_len(E[k]) << 2;        // edits_are_compatible -> merge_edits_in_place
        else E[++w] = E[k];
    }
    return w - b + 1;
}
VGK_HD void cs_open(CsState& S, uint32_t node, uint32_t offset) { S.c_node = node; S.c_offset = offset; S.c_begin = S.c_end = S.tail; S.c_from = 0; S.c_last = S.c_first = 0; }
// one edit of the mapping in the making; Mapping simplify's merging happens here (a run of the kind of the mapping's last run extends it)
VGK_HD void cs_edit(CsState& S, uint32_t kind, uint32_t len) {
    if (!len) return;
    if (kind != (uint32_t)VGK_WFA_DELETION) S.total_to += len;
    if (kind != (uint32_t)VGK_WFA_INSERTION) { S.total_from += len; S.c_from += len; }
    if (S.c_end > S.c_begin && cs_kind(S.c_last) == kind) { S.c_last += len << 2; return; }
    if (S.c_end > S.c_begin) { S.E[S.c_end - 1] = S.c_last; if (S.c_end - S.c_begin == 1) S.c_first = S.c_last; }
    if (S.c_end >= S.cap_e) { S.status = VGK_EOPS; return; }
    S.c_last = len << 2 | kind; ++S.c_end;
}
VGK_HD void cs_flush_last(CsState& S) { vgk_chain_mapping m; m.node = S.l_node; m.offset = S.l_offset; m.edit_begin = S.l_begin; m.n_edits = S.l_n; S.M[S.nm - 1] = m; }
// the mapping in the making is complete: simplify's loop body for it (:1324-1406)
VGK_HD void cs_close(CsState& S) {
    uint32_t mb = S.c_begin; const uint32_t me = S.c_end;
    if (me == mb) return;                                                      // no edits: redundant (:1334)
    S.E[me - 1] = S.c_last; if (me - mb == 1) S.c_first = S.c_last;
    if (!S.nm) {
        if (S.nm >= S.cap_m) { S.status = VGK_EOPS; return; }
        S.l_node = S.c_node; S.l_offset = S.c_offset; S.l_begin = mb; S.l_n = me - mb; S.l_from = S.c_from; S.l_last = S.c_last; S.l_mixed = false;
        S.nm = 1; S.tail = me; return;
    }
    // an insertion at the start of this mapping belongs to the previous one: appended there as it is (:1345-1352; the mapping's runs are merged: at most one)
    const bool moved = cs_kind(S.c_first) == (uint32_t)VGK_WFA_INSERTION;
    if (moved) { if (cs_kind(S.l_last) == (uint32_t)VGK_WFA_INSERTION) S.l_mixed = true; ++S.l_n; S.l_last = S.c_first; ++mb; }
    uint32_t node = S.c_node, offset = S.c_offset;
    if (S.l_node == VGK_WFA_NO_NODE && node != VGK_WFA_NO_NODE) { S.l_node = node; S.l_offset = offset; }              // (:1361-1369)
    else if (node == VGK_WFA_NO_NODE && S.l_node != VGK_WFA_NO_NODE) { node = S.l_node; offset = S.l_from; }           // (:1371-1380: the offset is from_length(*l), as written there)
    const bool joins = (S.l_node == VGK_WFA_NO_NODE && node == VGK_WFA_NO_NODE) || (S.l_node != VGK_WFA_NO_NODE && node != VGK_WFA_NO_NODE && S.l_node == node && S.l_offset + S.l_from == offset);
    const uint32_t rem = me - mb;
    if (joins) {                                                               // concat_mappings: all edits, merged again (:1382-1394)
        if (S.l_mixed) {                                                       // runs of one kind side by side somewhere in l: the general merge, over memory
            S.l_n = cs_merge_runs(S.E, S.l_begin, S.l_n + rem); S.l_last = S.E[S.l_begin + S.l_n - 1]; S.l_mixed = false;
        } else if (rem) {                                                      // l and m are merged inside: only the run where they meet can merge
            const uint32_t first_rem = !moved ? S.c_first : rem == 1 ? S.c_last : S.E[mb];
            if (cs_kind(first_rem) == cs_kind(S.l_last)) {
                const uint32_t merged = S.l_last + (cs_len(first_rem) << 2);
                S.E[S.l_begin + S.l_n - 1] = merged;
                for (uint32_t x = mb + 1; x < me; ++x) S.E[x - 1] = S.E[x];
                S.l_n += rem - 1; S.l_last = rem == 1 ? merged : S.c_last;
            } else { S.l_n += rem; S.l_last = S.c_last; }
        }
        S.l_from += S.c_from;
        S.tail = S.l_begin + S.l_n;
    } else if (rem) {                                                          // from_length(m) || to_length(m) (:1396)
        if (S.nm >= S.cap_m) { S.status = VGK_EOPS; return; }
        cs_flush_last(S);
        S.l_node = node; S.l_offset = offset; S.l_begin = mb; S.l_n = rem; S.l_from = S.c_from; S.l_last = S.c_last; S.l_mixed = false;
        ++S.nm; S.tail = me;
    } else S.tail = mb;
}
// WFAAlignment::to_path over (node path, node_offset, edit runs): a mapping per node the edits reach
VGK_HD void cs_alignment(const CsParams& P, CsState& S, const uint32_t* path, uint32_t path_len, uint32_t node_offset, const uint32_t* runs, uint32_t n_runs) {
    if (!path_len) {
        if (n_runs == 1 && cs_kind(runs[0]) == (uint32_t)VGK_WFA_INSERTION) {   // unlocalized_insertion(): a mapping without a position (:964-970)
            cs_open(S, VGK_WFA_NO_NODE, 0); cs_edit(S, VGK_WFA_INSERTION, cs_len(runs[0])); cs_close(S);
        } else if (n_runs) S.status = VGK_EINVAL;
        return;                                                                // path empty: an empty Path (:972-974)
    }
    if (path[0] >= P.index.n_oriented) { S.status = VGK_EINVAL; return; }
    // The walk's inputs are read AHEAD of their use — the node after the current one and its length, the node after that, the run after the current
    // one — so that a mapping costs no load it has to wait a whole memory latency for: the loads of step s + 1 and s + 2 are in flight while step s
    // is written.  (A node index beyond the index is refused when the walk reaches it, as before: its "length" is then never loaded.)
    const uint32_t NO = P.index.n_oriented;
    uint32_t step = 0, node_at = node_offset, node_end = g_len(P.index, (int32_t)path[0]);
    uint32_t n1 = path_len > 1 ? path[1] : 0u, n2 = path_len > 2 ? path[2] : 0u;
    uint32_t l1 = path_len > 1 && n1 < NO ? g_len(P.index, (int32_t)n1) : 0u;
    if (node_offset >= node_end || !n_runs) { S.status = VGK_EINVAL; return; } // "offset to or past end of first node", "has no edits"
    cs_open(S, path[0], node_offset);
    uint32_t run_next = runs[0];
    for (uint32_t k = 0; k < n_runs && S.status == VGK_OK; ++k) {
        const uint32_t run = run_next;
        if (k + 1 < n_runs) run_next = runs[k + 1];
        const uint32_t kind = cs_kind(run);
        uint32_t left = cs_len(run);
        if (!left) { S.status = VGK_EINVAL; return; }                          // "has empty edit"
        const bool uses_graph = kind != (uint32_t)VGK_WFA_INSERTION;
        while (left && S.status == VGK_OK) {
            uint32_t take = left;
            if (uses_graph) {
                if (step == path_len || node_at == node_end) { S.status = VGK_EINVAL; return; }     // "tried to go past end of path / node"
                if (node_end - node_at < take) take = node_end - node_at;
            }
            cs_edit(S, kind, take);
            left -= take;
            if (uses_graph) {
                node_at += take;
                if (node_at == node_end) {
                    node_at = 0; ++step;
                    if (step != path_len) {
                        const uint32_t node = n1;
                        if (node >= NO) { S.status = VGK_EINVAL; return; }
                        node_end = l1;
                        if (!node_end) { S.status = VGK_EINVAL; return; }      // "has empty node"
                        n1 = n2;                                                // (loaded a step ago)
                        l1 = step + 1 < path_len && n1 < NO ? g_len(P.index, (int32_t)n1) : 0u;
                        n2 = step + 2 < path_len ? path[step + 2] : 0u;
                        cs_close(S); cs_open(S, node, 0);
                    } else node_end = 0;
                }
            }
        }
    }
    if (S.status == VGK_OK) cs_close(S);
}

VGK_HD void cs_stitch_one(const CsParams& P, uint32_t r) {
    const uint32_t R = P.n_reads + 1;
    CsState S;
    S.M = P.work_m + P.slot[r]; S.E = P.work_e + P.slot[R + r]; S.cap_m = P.bound[r]; S.cap_e = P.bound[R + r];
    S.nm = 0; S.tail = 0; S.status = VGK_OK; S.total_to = S.total_from = 0;
    S.l_node = VGK_WFA_NO_NODE; S.l_offset = S.l_begin = S.l_n = S.l_from = S.l_last = 0; S.l_mixed = false;
    cs_open(S, VGK_WFA_NO_NODE, 0);
    // (the next piece and, for a LINK, the result it names are read while the current piece is walked)
    const uint64_t k_end = P.piece_off[r + 1];
    vgk_chain_piece pc_next{}; vgk_wfa_result w_next{};
    auto fetch = [&](uint64_t k) { pc_next = P.pieces[k]; if (pc_next.kind == (uint32_t)VGK_PIECE_LINK && pc_next.link < P.n_links) w_next = P.link_res[pc_next.link]; };
    if (P.piece_off[r] < k_end) fetch(P.piece_off[r]);
    for (uint64_t k = P.piece_off[r]; k < k_end && S.status == VGK_OK; ++k) {
        const vgk_chain_piece pc = pc_next; const vgk_wfa_result w = w_next;
        if (k + 1 < k_end) fetch(k + 1);
        if (pc.kind == (uint32_t)VGK_PIECE_LINK) {
            if (pc.link >= P.n_links || w.status != VGK_OK || !w.ok || (uint64_t)w.path_begin + w.path_len > P.link_path_cap || (uint64_t)w.edit_begin + w.n_edits > P.link_edit_cap) { S.status = VGK_EINVAL; break; }          // "WFAAlignment is not OK and cannot become a path" (cs_link_ok over the result already read)
            cs_alignment(P, S, P.link_paths + w.path_begin, w.path_len, w.node_offset, P.link_edits + w.edit_begin, w.n_edits);
        } else if (pc.kind == (uint32_t)VGK_PIECE_ALIGNMENT) {
            if ((uint64_t)pc.path_begin + pc.path_len > P.n_nodes || (uint64_t)pc.edit_begin + pc.n_edits > P.n_edits) { S.status = VGK_EINVAL; break; }
            cs_alignment(P, S, P.nodes + pc.path_begin, pc.path_len, pc.node_offset, P.edits + pc.edit_begin, pc.n_edits);
        } else if (pc.kind == (uint32_t)VGK_PIECE_PATH) {
            if ((uint64_t)pc.path_begin + pc.path_len > P.n_mappings) { S.status = VGK_EINVAL; break; }
            for (uint32_t q = 0; q < pc.path_len && S.status == VGK_OK; ++q) {
                const vgk_chain_mapping m = P.mappings[pc.path_begin + q];
                if ((uint64_t)m.edit_begin + m.n_edits > P.n_edits || (m.node != VGK_WFA_NO_NODE && m.node >= P.index.n_oriented)) { S.status = VGK_EINVAL; break; }
                cs_open(S, m.node, m.node == VGK_WFA_NO_NODE ? 0u : m.offset);          // (a mapping without a position has no offset)
                for (uint32_t x = 0; x < m.n_edits; ++x) cs_edit(S, cs_kind(P.edits[m.edit_begin + x]), cs_len(P.edits[m.edit_begin + x]));
                cs_close(S);
            }
        } else S.status = VGK_EINVAL;
    }
    vgk_chain_result out; out.status = S.status; out.mapping_begin = P.slot[r]; out.edit_begin = P.slot[R + r];
    out.n_mappings = out.n_edits = out.from_length = out.to_length = 0; out.reserved = 0;
    if (S.status == VGK_OK && S.nm && S.total_to) {
        cs_flush_last(S);
        // Deletions before the first and after the last read base go (:1422-1475): the mappings without a read base at either end, the deletion at the
        // start of the first mapping that has one (its offset moves on), the deletion at the end of the last one — when that is another mapping (a path
        // whose one mapping holds every read base keeps its trailing deletion: the first-mapping branch at :1453-1470 does not look at the end).  Only
        // the ends are looked at; what lies between is where it is: the result names the stretch of mappings and of runs that is left.
        uint32_t f = 0, g = S.nm - 1, from = S.total_from;
        while (f < S.nm) { const vgk_chain_mapping m = S.M[f]; if (cs_to_length(S.E, m.edit_begin, m.n_edits)) break; from -= cs_from_length(S.E, m.edit_begin, m.n_edits); ++f; }
        while (g > f) { const vgk_chain_mapping m = S.M[g]; if (cs_to_length(S.E, m.edit_begin, m.n_edits)) break; from -= cs_from_length(S.E, m.edit_begin, m.n_edits); --g; }
        vgk_chain_mapping mf = S.M[f];
        if (mf.n_edits && cs_kind(S.E[mf.edit_begin]) == (uint32_t)VGK_WFA_DELETION) {      // (runs are merged: one run at most)
            const uint32_t d = cs_len(S.E[mf.edit_begin]); mf.offset += d; ++mf.edit_begin; --mf.n_edits; from -= d;
        }
        if (mf.node == VGK_WFA_NO_NODE) mf.offset = 0;                         // an empty position is cleared (:1484-1487)
        S.M[f] = mf;
        uint32_t end_e;
        if (g > f) {
            vgk_chain_mapping mg = S.M[g];
            if (mg.n_edits && cs_kind(S.E[mg.edit_begin + mg.n_edits - 1]) == (uint32_t)VGK_WFA_DELETION) { from -= cs_len(S.E[mg.edit_begin + mg.n_edits - 1]); --mg.n_edits; S.M[g] = mg; }
            end_e = mg.edit_begin + mg.n_edits;
        } else end_e = mf.edit_begin + mf.n_edits;
        out.mapping_begin = P.slot[r] + f; out.n_mappings = g - f + 1;
        out.edit_begin = P.slot[R + r] + mf.edit_begin; out.n_edits = end_e - mf.edit_begin;
        out.from_length = from; out.to_length = S.total_to;
    }
    P.res[r] = out;
    P.count[r] = out.n_mappings; P.count[R + r] = out.n_edits;
}

// ---- CS_GATHER: lane `lane` of `lanes` copies its share of read r's mappings and edit runs into the dense arrays -----------------------------
VGK_HD void cs_gather_one(const CsParams& P, uint32_t r, uint32_t lane, uint32_t lanes) {
    const uint32_t R = P.n_reads + 1;
    const vgk_chain_result w = P.res[r];
    const uint32_t mb = P.out_slot[r], eb = P.out_slot[R + r];
    const bool fits = (uint64_t)mb + w.n_mappings <= P.out_m_cap && (uint64_t)eb + w.n_edits <= P.out_e_cap;
    if (fits) {
        const uint32_t rebase = eb - (w.edit_begin - P.slot[R + r]);          // a mapping's runs are named by their place in the read's stretch; the result's first run sits at w.edit_begin
        for (uint32_t k = lane; k < w.n_mappings; k += lanes) { vgk_chain_mapping m = P.work_m[w.mapping_begin + k]; m.edit_begin += rebase; P.out_m[mb + k] = m; }
        for (uint32_t k = lane; k < w.n_edits; k += lanes) P.out_e[eb + k] = P.work_e[w.edit_begin + k];
    }
    if (lane == 0) {
        vgk_chain_result o = w; o.mapping_begin = mb; o.edit_begin = eb;
        if (!fits && o.status == VGK_OK) o.status = VGK_EOPS;
        P.out_res[r] = o;
    }
}

// what one lane does in stage `what`: item r of n_reads + 1 (the bounds' and sizes' last entries are 0: the prefix sums' totals land there)
VGK_HD void cs_one(const CsParams& P, int what, uint32_t r) {
    if (what == CS_BOUND) { if (r <= P.n_reads) cs_bound_one(P, r); }
    else if (what == CS_STITCH) { if (r < P.n_reads) cs_stitch_one(P, r); else if (r == P.n_reads) { P.count[r] = 0; P.count[(size_t)P.n_reads + 1 + r] = 0; } }
}

}  // namespace vgk
