// gapless_api.cpp — vgk_haplo_create / vgk_gapless_extend: the host half of haplotype-consistent gapless extension.
//
// The index: what the reference gets from GBWTGraph (node sequences in both orientations + GBWT records) is built
// here from the caller's threads in uncompressed form and kept in HBM.  The visits of an oriented node are laid down
// in GBWT order — by (predecessor node, rank in the predecessor's record), threads that start at the node first in
// thread order — by delivering every finished record to its successors when the threads are acyclic as oriented-node
// sequences, and by prefix doubling over the reversed prefixes otherwise.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.hpp"
#include "haplo.hpp"
#include "host_parallel.hpp"

using namespace vgk;

namespace {
struct GLap {                     // VGAMD_TIMING=1: where a vgk_gapless_extend call spends its time on the host
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); const bool on = std::getenv("VGAMD_TIMING") != nullptr;
    void operator()(const char* what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gapless_extend] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count()); t0 = t;
    }
};
}  // namespace


namespace {

struct GaplessHost { PinnedBuf<char> reads; PinnedBuf<vgk_seed> seeds; PinnedBuf<vgk_gapless_result> dres; PinnedBuf<vgk_extension> dext; PinnedBuf<uint32_t> dnodes, dmism; };

char complement(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}
template <class T> int put(vgk_haplo* h, const std::vector<T>& v, const T*& out) {
    void* d = h->ctx->be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!d) return VGK_ENOMEM;
    h->held.push_back(d);
    if (!v.empty()) { int rc = h->ctx->be->upload(d, v.data(), v.size() * sizeof(T)); if (rc) return rc; }
    out = (const T*)d;
    return VGK_OK;
}
struct Arrival { int32_t pred; uint32_t order, seq, pos; };

}  // namespace

// the sets of a VGK_GAPLESS_DEFER call: wait for the copies, then out of the staging buffers in slices on the host threads (context lock held)
int vgk_ctx::start_deferred() {
    if (!deferred.pending || deferred.queued) return VGK_OK;
    deferred.queued = true;
    if (!deferred.ev) deferred.ev = be->event_create();
    int rc = be->event_record(deferred.ev);
    if (!rc) rc = be->fetch_after(deferred.ev);
    for (int k = 0; k < 4 && !rc; ++k) if (deferred.spans[k].bytes) rc = be->download_fetch_async(const_cast<char*>(deferred.spans[k].src), deferred.from[k], deferred.spans[k].bytes);
    return rc;
}
int vgk_ctx::finish_deferred() {
    if (!deferred.pending) return VGK_OK;
    int rc = start_deferred();
    deferred.pending = false;
    if (!rc) rc = be->sync_fetch();
    if (rc) return rc;
    std::vector<DeferredSpan> slices;
    for (const DeferredSpan& sp : deferred.spans) for (size_t at = 0; at < sp.bytes; at += (size_t)8 << 20) slices.push_back({sp.dst + at, sp.src + at, std::min<size_t>((size_t)8 << 20, sp.bytes - at)});
    parallel_tasks((uint32_t)slices.size(), [&](uint32_t k) { std::memcpy(slices[k].dst, slices[k].src, slices[k].bytes); });
    return VGK_OK;
}

extern "C" {

// both strands of every node, as the kernels read them: forward strands in node order, then the reverse complements (8 bytes of padding at
// either end: the kernels compare eight bases per load)
extern "C++" int vgk_haplo_strands(uint32_t N, const uint32_t* node_len, const char* fwd, std::vector<uint32_t>& len, std::vector<uint32_t>& seq_off, std::vector<char>& seq, uint64_t& total) {
    const uint32_t O = 2 * N;
    len.assign(O, 0); seq_off.assign(O, 0);
    total = 0; for (uint32_t i = 0; i < N; ++i) total += node_len[i];
    if (2 * total > 0xfffffff0ull) return VGK_ETOOBIG;
    seq.assign(2 * total + 16, 0);
    uint32_t at = 8, rat = (uint32_t)total + 8;
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t L = node_len[i];
        len[2 * i] = len[2 * i + 1] = L; seq_off[2 * i] = at; seq_off[2 * i + 1] = rat;
        for (uint32_t k = 0; k < L; ++k) { seq[at + k] = fwd[at - 8 + k]; seq[rat + k] = complement(fwd[at - 8 + L - 1 - k]); }
        at += L; rat += L;
    }
    return VGK_OK;
}

int vgk_haplo_create(vgk_ctx* ctx, const vgk_haplotypes* d, vgk_haplo** out) try {
    if (!ctx || !d || !out || !d->n_nodes || !d->node_len || !d->seq || (d->n_threads && (!d->thread_off || !d->thread_nodes))) return VGK_EINVAL;
    *out = nullptr;
    const uint32_t N = d->n_nodes, O = 2 * N, S = 2 * d->n_threads;
    for (uint32_t t = 0; t < d->n_threads; ++t) for (uint32_t k = d->thread_off[t]; k < d->thread_off[t + 1]; ++k) if (d->thread_nodes[k] >= O) return VGK_EINVAL;
    // both strands of every node
    std::vector<uint32_t> len, seq_off; std::vector<char> seq; uint64_t total = 0;
    if (int rc0 = vgk_haplo_strands(N, d->node_len, d->seq, len, seq_off, seq, total)) return rc0;
    // sequences: thread t forward = 2t, reverse complement = 2t + 1
    std::vector<uint32_t> soff(S + 1, 0);
    for (uint32_t t = 0; t < d->n_threads; ++t) { const uint32_t n = d->thread_off[t + 1] - d->thread_off[t]; soff[2 * t + 1] = soff[2 * t] + n; soff[2 * t + 2] = soff[2 * t + 1] + n; }
    const uint32_t V = soff[S];
    std::vector<int32_t> sn(V);
    for (uint32_t t = 0; t < d->n_threads; ++t) {
        const uint32_t n = d->thread_off[t + 1] - d->thread_off[t];
        for (uint32_t k = 0; k < n; ++k) { const uint32_t o = d->thread_nodes[d->thread_off[t] + k]; sn[soff[2 * t] + k] = (int32_t)o; sn[soff[2 * t + 1] + (n - 1 - k)] = (int32_t)(o ^ 1u); }
    }
    std::vector<uint32_t> count(O, 0), body_off(O + 1, 0);
    for (uint32_t i = 0; i < V; ++i) ++count[sn[i]];
    for (uint32_t o = 0; o < O; ++o) body_off[o + 1] = body_off[o] + count[o];
    std::vector<Arrival> arr(V);
    std::vector<uint32_t> got(O, 0), queue;
    for (uint32_t s = 0; s < S; ++s) if (soff[s + 1] > soff[s]) { const int32_t o = sn[soff[s]]; arr[body_off[o] + got[o]++] = {-1, s, s, 0}; }
    uint32_t with_visits = 0;
    for (uint32_t o = 0; o < O; ++o) { if (count[o]) ++with_visits; if (count[o] && got[o] == count[o]) queue.push_back(o); }
    std::vector<int32_t> succ(V);
    for (size_t qh = 0; qh < queue.size(); ++qh) {
        const uint32_t o = queue[qh];
        Arrival* a = arr.data() + body_off[o];
        std::sort(a, a + count[o], [](const Arrival& x, const Arrival& y) { return x.pred != y.pred ? x.pred < y.pred : x.order < y.order; });
        for (uint32_t i = 0; i < count[o]; ++i) {
            const uint32_t slen = soff[a[i].seq + 1] - soff[a[i].seq];
            const int32_t w = a[i].pos + 1 < slen ? sn[soff[a[i].seq] + a[i].pos + 1] : -1;
            succ[body_off[o] + i] = w;
            if (w < 0) continue;
            arr[body_off[w] + got[w]++] = {(int32_t)o, i, a[i].seq, a[i].pos + 1};
            if (got[w] == count[w]) queue.push_back((uint32_t)w);
        }
    }
    if (queue.size() != with_visits) {
        // Threads that revisit a node leave no topological order of records.  The rule is the same — visits ordered by their
        // reversed prefixes (predecessor, its predecessor, ..., thread start; starts by thread number) — evaluated by prefix
        // doubling over all visits: rank by the first h symbols, then pair with the rank of the visit h steps back.
        std::vector<uint32_t> seq_of(V), rank(V), next(V), idx(V);
        uint32_t maxlen = 0;
        for (uint32_t s = 0; s < S; ++s) {
            maxlen = std::max(maxlen, soff[s + 1] - soff[s]);
            for (uint32_t v = soff[s]; v < soff[s + 1]; ++v) { seq_of[v] = s; rank[v] = v == soff[s] ? s : S + (uint32_t)sn[v - 1]; }
        }
        std::vector<uint64_t> key(V);
        for (uint32_t h = 1; h <= maxlen; h *= 2) {
            for (uint32_t v = 0; v < V; ++v) { const uint32_t k = v - soff[seq_of[v]]; key[v] = ((uint64_t)rank[v] << 32) | (k >= h ? (uint64_t)rank[v - h] + 1u : 0u); idx[v] = v; }
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
            uint32_t distinct = 0;
            for (uint32_t i = 0; i < V; ++i) { if (i && key[idx[i]] != key[idx[i - 1]]) ++distinct; next[idx[i]] = distinct; }
            rank.swap(next);
            if (distinct + 1 == V) break;
        }
        for (uint32_t v = 0; v < V; ++v) { key[v] = ((uint64_t)(uint32_t)sn[v] << 32) | rank[v]; idx[v] = v; }
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });   // record after record
        for (uint32_t i = 0; i < V; ++i) {
            const uint32_t v = idx[i], s = seq_of[v], k = v - soff[s];
            arr[i] = {k ? sn[v - 1] : -1, 0, s, k};
            succ[i] = v + 1 < soff[s + 1] ? sn[v + 1] : -1;
        }
    }
    HaploTables T;
    std::vector<uint32_t>& edge_off = T.edge_off; std::vector<uint32_t>& body = T.body; std::vector<uint32_t>& edge_base = T.edge_base; std::vector<int32_t>& edge_to = T.edge_to;
    edge_off.assign(O + 1, 0); body.assign(V, 0);
    for (uint32_t o = 0; o < O; ++o) {
        std::vector<int32_t> e(succ.begin() + body_off[o], succ.begin() + body_off[o + 1]);
        std::sort(e.begin(), e.end()); e.erase(std::unique(e.begin(), e.end()), e.end());
        for (uint32_t i = 0; i < count[o]; ++i) body[body_off[o] + i] = (uint32_t)(std::lower_bound(e.begin(), e.end(), succ[body_off[o] + i]) - e.begin());
        edge_to.insert(edge_to.end(), e.begin(), e.end());
        edge_off[o + 1] = (uint32_t)edge_to.size();
    }
    edge_base.assign(edge_to.size() + 1, 0);
    for (uint32_t w = 0; w < O; ++w) {
        const Arrival* a = arr.data() + body_off[w];
        for (uint32_t i = 0; i < count[w]; ++i) if (a[i].pred >= 0 && (!i || a[i - 1].pred != a[i].pred)) {
            const uint32_t o = (uint32_t)a[i].pred;
            const int32_t* first = edge_to.data() + edge_off[o]; const int32_t* last = edge_to.data() + edge_off[o + 1];
            edge_base[(uint32_t)(std::lower_bound(first, last, (int32_t)w) - edge_to.data())] = i;
        }
    }
    T.count.swap(count); T.body_off.swap(body_off);
    return vgk_haplo_from_tables(ctx, O, len, seq_off, seq, (uint32_t)total, T, out);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

struct vgk_haplo::PendingMerge { uint32_t O = 0, total = 0; std::vector<uint32_t> len; std::vector<char> seq; HaploTables T; };

// The records as the kernels read them, from the tables either builder makes (this file's, from threads; gbwt_file.cpp's, straight from
// a GBWT's own records): per oriented node its visits (count), per visit the edge it leaves through (body, from body_off), its edges in
// successor order (edge_to from edge_off; -1 = the path ends here, first) and per edge the rank of its first visit in the successor's record.
// Unary runs merged (gapless_device.hpp GMerge): runs of consecutive nodes v, v + 1, ... where every visit of 2 v leaves through the one edge to
// 2 (v + 1) and every visit of 2 (v + 1) arrives that way (rank offset 0, equal visit counts), the same on the other strand, at most 255 bases
// together.  The merged node takes the visits of its first node and the edges and visit bodies of its last (in either orientation); its bases
// are the run's, which already lie behind each other.  -> VGK_OK with h->merged set, or with nothing set when no two nodes merge.
static int merge_unary_runs(vgk_ctx* ctx, vgk_haplo* h, uint32_t O, const std::vector<uint32_t>& len, const std::vector<char>& seq, uint32_t total, const HaploTables& T) {
    const uint32_t N = O / 2;
    // For the gapless search: built, exact (tie for tie), measured on the MI355X (profiles/r05/NOTES.md) — and OFF unless VGAMD_HAPLO_MERGE=1: the search kernel's
    // time is the bytes it moves, not its hops (6.48 against 6.62 ms per million reads at chr22-scale variant density, 13.2 against 9.5 ms where runs
    // are two or three nodes), and the merged build's extra live values are scratch traffic of their own.
    // The WFA wavefront kernel walks the merged index always (its trie walk is one lane's chain of record fetches: hops are its time).
    if (N < 2 || std::getenv("VGAMD_HAPLO_NO_MERGE")) return VGK_OK;
    auto unary = [&](uint32_t o, uint32_t p) {
        return T.count[o] > 0 && T.edge_off[o + 1] - T.edge_off[o] == 1 && T.edge_to[T.edge_off[o]] == (int32_t)p && T.edge_base[T.edge_off[o]] == 0 && T.count[p] == T.count[o];
    };
    constexpr uint32_t MAX_RUN_BASES = 255;                 // the fast kernel's entries hold 8-bit offsets (gapless_device.hpp GStoreLds)
    std::vector<uint32_t> run_first; std::vector<uint32_t> run_of(N);
    uint32_t bases = 0;
    for (uint32_t v = 0; v < N; ++v) {
        const bool joins = v > 0 && unary(2 * (v - 1), 2 * v) && unary(2 * v + 1, 2 * (v - 1) + 1) && bases + len[2 * v] <= MAX_RUN_BASES;
        if (!joins) { run_first.push_back(v); bases = 0; }
        bases += len[2 * v];
        run_of[v] = (uint32_t)run_first.size() - 1;
    }
    const uint32_t M = (uint32_t)run_first.size();
    if (M == N) return VGK_OK;

    run_first.push_back(N);
    std::vector<uint32_t> ocol((size_t)N + 1, 0);
    for (uint32_t v = 0; v < N; ++v) ocol[v + 1] = ocol[v] + len[2 * v];
    std::vector<uint32_t> mlen(M);
    for (uint32_t m = 0; m < M; ++m) mlen[m] = ocol[run_first[m + 1]] - ocol[run_first[m]];
    HaploTables U;
    U.count.assign(2 * M, 0); U.body_off.assign(2 * (size_t)M + 1, 0); U.edge_off.assign(2 * (size_t)M + 1, 0);
    for (uint32_t mo = 0; mo < 2 * M; ++mo) {
        const uint32_t m = mo >> 1, v0 = run_first[m], v1 = run_first[m + 1];
        const uint32_t first = (mo & 1u) ? 2 * (v1 - 1) + 1 : 2 * v0, last = (mo & 1u) ? 2 * v0 + 1 : 2 * (v1 - 1);
        U.count[mo] = T.count[first];
        if (T.count[last] != T.count[first]) return VGK_OK;                              // (cannot happen; the original index is complete either way)
        U.body.insert(U.body.end(), T.body.begin() + T.body_off[last], T.body.begin() + T.body_off[last + 1]);
        U.body_off[mo + 1] = (uint32_t)U.body.size();
        for (uint32_t e = T.edge_off[last]; e < T.edge_off[last + 1]; ++e) {
            const int32_t to = T.edge_to[e];
            int32_t mt = -1;
            if (to >= 0) {
                const uint32_t tv = (uint32_t)to >> 1, tm = run_of[tv];
                if (((uint32_t)to & 1u) ? tv + 1 != run_first[tm + 1] : tv != run_first[tm]) return VGK_OK;      // a successor inside a run: not a unary run after all
                mt = (int32_t)(2 * tm + ((uint32_t)to & 1u));
            }
            U.edge_to.push_back(mt); U.edge_base.push_back(T.edge_base[e]);
        }
        U.edge_off[mo + 1] = (uint32_t)U.edge_to.size();
    }
    U.edge_base.push_back(0);
    std::vector<uint32_t> mlen2, mseq_off; std::vector<char> mseq; uint64_t mtotal = 0;
    int rc = vgk_haplo_strands(M, mlen.data(), seq.data() + 8, mlen2, mseq_off, mseq, mtotal);      // (the forward strands of the runs ARE the forward strands of the nodes, in a row)
    if (rc || mtotal != total) return rc;
    vgk_haplo* mh = nullptr;
    if ((rc = vgk_haplo_from_tables(ctx, 2 * M, mlen2, mseq_off, mseq, total, U, &mh, false))) return rc;
    std::vector<uint64_t> seed_map(O);
    for (uint32_t o = 0; o < O; ++o) {
        const uint32_t v = o >> 1, m = run_of[v], v0 = run_first[m], v1 = run_first[m + 1];
        const uint32_t off = (o & 1u) ? ocol[v1] - ocol[v + 1] : ocol[v] - ocol[v0];
        seed_map[o] = (uint64_t)(2 * m + (o & 1u)) | ((uint64_t)off << 32);
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    if ((rc = put(h, seed_map, h->merge.seed_map)) || (rc = put(h, run_first, h->merge.run_first)) || (rc = put(h, ocol, h->merge.ocol)) || (rc = ctx->be->sync())) { vgk_haplo_destroy(mh); return rc; }
    h->merge.on = 1; h->merge.n_orig_oriented = O; h->merged = mh;
    h->search_merged = std::getenv("VGAMD_HAPLO_MERGE") != nullptr;              // (only with a merged form to search)
    return VGK_OK;
}

extern "C++" int vgk_haplo_from_tables(vgk_ctx* ctx, uint32_t O, const std::vector<uint32_t>& len, const std::vector<uint32_t>& seq_off, const std::vector<char>& seq, uint32_t total,
                          const HaploTables& T, vgk_haplo** out, bool merge_runs) {
    const std::vector<uint32_t>& count = T.count; const std::vector<uint32_t>& body_off = T.body_off; const std::vector<uint32_t>& body = T.body;
    const std::vector<uint32_t>& edge_off = T.edge_off; const std::vector<int32_t>& edge_to = T.edge_to; const std::vector<uint32_t>& edge_base = T.edge_base;
    // one padded record per oriented node (layout in gapless_device.hpp): sizes first, so that every edge can name its successor's
    // record.  The visit body of a record is stored as bytes or run-length encoded (one word per run), whichever is smaller.
    std::vector<uint32_t> rec_off(O + 1, 0), rec, runs_of(O, 0);
    const bool allow_rle = std::getenv("VGAMD_HAPLO_NO_RLE") == nullptr;
    { uint64_t at = 0;
      for (uint32_t o = 0; o < O; ++o) {
          const uint32_t ne = edge_off[o + 1] - edge_off[o];
          if (ne > 255 || count[o] > 65535 || len[o] > 65535) return VGK_ETOOBIG;      // edge numbers are bytes, ranks and lengths 16 bits
          uint32_t runs = 0;
          for (uint32_t i = 0; i < count[o]; ++i) if (!i || body[body_off[o] + i] != body[body_off[o] + i - 1]) ++runs;
          const bool rle = allow_rle && runs && 4ull * runs < (count[o] + 3) / 4 * 4;
          runs_of[o] = rle ? runs : 0;
          rec_off[o] = (uint32_t)at;
          at += (4 + 4ull * ne + (rle ? runs : (count[o] + 3) / 4) + 15) / 16 * 16;
          if (at > 0xfffffff0ull) return VGK_ETOOBIG;
      }
      rec_off[O] = (uint32_t)at; rec.assign(at, 0); }
    for (uint32_t o = 0; o < O; ++o) {
        uint32_t* r = rec.data() + rec_off[o];
        const uint32_t ne = edge_off[o + 1] - edge_off[o];
        r[0] = count[o]; r[1] = ne | (runs_of[o] ? 0x80000000u : 0u); r[2] = len[o]; r[3] = seq_off[o];
        for (uint32_t k = 0; k < ne; ++k) {
            const int32_t to = edge_to[edge_off[o] + k];
            r[4 + 4 * k] = (uint32_t)to;
            r[5 + 4 * k] = edge_base[edge_off[o] + k] | (to >= 0 ? len[(uint32_t)to] << 16 : 0u);
            r[6 + 4 * k] = to >= 0 ? seq_off[(uint32_t)to] : 0u;
            r[7 + 4 * k] = to >= 0 ? rec_off[(uint32_t)to] : 0u;
        }
        uint32_t* visits = r + 4 + 4 * ne;
        if (runs_of[o]) {
            uint32_t k = 0;
            for (uint32_t i = 0; i < count[o];) {
                uint32_t j = i; while (j < count[o] && body[body_off[o] + j] == body[body_off[o] + i]) ++j;
                visits[k++] = body[body_off[o] + i] | ((j - i) << 8);
                i = j;
            }
        } else for (uint32_t i = 0; i < count[o]; ++i) visits[i >> 2] |= body[body_off[o] + i] << (8 * (i & 3u));
    }
    rec_off[O] = (uint32_t)rec.size();
    vgk_haplo* h = new vgk_haplo();
    h->ctx = ctx; h->n_oriented = O; h->len = len;
    std::unique_lock<std::mutex> lock(ctx->mu);
    int rc;
    h->dev.n_oriented = O; h->dev.strand_shift = (uint32_t)total;
    h->dev.max_node_len = 0; h->dev.max_visits = 0;          // what the fast kernel's compact entries have to hold (gapless_device.hpp)
    for (uint32_t o = 0; o < O; ++o) { h->dev.max_node_len = std::max(h->dev.max_node_len, len[o]); h->dev.max_visits = std::max(h->dev.max_visits, count[o]); }
    std::vector<uint64_t> node_tab(O);
    for (uint32_t o = 0; o < O; ++o) node_tab[o] = (uint64_t)seq_off[o] | ((uint64_t)len[o] << 32);
    if ((rc = put(h, rec_off, h->dev.rec_off)) || (rc = put(h, rec, h->dev.rec)) || (rc = put(h, seq, h->dev.seq)) || (rc = put(h, node_tab, h->dev.node_tab)) || (rc = ctx->be->sync())) {
        for (void* p : h->held) ctx->be->release(p);
        delete h; return rc;
    }
    lock.unlock();
    if (merge_runs && !std::getenv("VGAMD_HAPLO_NO_MERGE")) {
        if (std::getenv("VGAMD_HAPLO_MERGE")) { if ((rc = merge_unary_runs(ctx, h, O, len, seq, total, T))) { vgk_haplo_destroy(h); return rc; } }      // the gapless search walks it: now
        else { auto pm = std::make_shared<vgk_haplo::PendingMerge>(); pm->O = O; pm->len = len; pm->seq = seq; pm->total = total; pm->T = T; h->pending_merge = pm; }   // the first vgk_wfa_extend builds it
    }
    *out = h;
    return VGK_OK;
}

extern "C++" int vgk_haplo_ensure_merged(vgk_haplo* h) {
    if (!h) return VGK_EINVAL;
    std::lock_guard<std::mutex> once(h->merge_mu);
    if (!h->pending_merge) return VGK_OK;
    std::shared_ptr<vgk_haplo::PendingMerge> pm = h->pending_merge;
    const int rc = merge_unary_runs(h->ctx, h, pm->O, pm->len, pm->seq, pm->total, pm->T);
    h->pending_merge.reset();                                                   // (built, or not buildable: either way the tables go)
    return rc;
}

void vgk_haplo_destroy(vgk_haplo* h) {
    if (!h) return;
    if (h->merged) vgk_haplo_destroy(h->merged);
    { std::lock_guard<std::mutex> lock(h->ctx->mu); for (void* p : h->held) h->ctx->be->release(p); }
    delete h;
}

// Everything behind "the inputs are on the device": the three kernels, the sets put in problem order on the device, the way back.
// P carries index, probs, reads, seeds, order, n; n_seed = seeds of the whole batch; `slot` = the next free scratch slot.
static int gapless_run_and_fetch(vgk_ctx* ctx, GaplessParams& P, uint32_t n, uint64_t n_seed, int next_slot, GaplessHost& H, GLap& lap,
                                 vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                                 uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3], const bool defer = false) {
    Backend* be = ctx->be.get();
    auto cleanup = [&](int rc) { return rc; };
    { const int rc0 = ctx->finish_deferred(); if (rc0) return rc0; }         // (an earlier call's sets still on their way use the same staging)
    auto dev = [&](const void* src, size_t bytes) -> void* {
        void* d = ctx->ensure_scratch(next_slot++, std::max<size_t>(bytes, 16)); if (!d) return nullptr;
        if (src && bytes && be->upload(d, src, bytes)) return nullptr;
        return d;
    };
    P.match = ctx->sc.matrix[0]; P.mismatch = -ctx->sc.matrix[1]; P.bonus = ctx->sc.full_length_bonus;
    // dense outputs: at most one extension per seed; nodes / mismatches sized generously and checked on the device
    const uint64_t cap_e = n_seed + 1, cap_n = std::min<uint64_t>(n_seed * G_PATH, std::max<uint64_t>(n_seed * 16 + 1024, nodes_cap)) + 1,
                   cap_m = std::min<uint64_t>(n_seed * G_MISM, std::max<uint64_t>(n_seed * 8 + 1024, mism_cap)) + 1;
    P.flat_min_idle = G_FLAT_MIN_IDLE;
    if (const char* e = std::getenv("VGAMD_GAPLESS_MIN_IDLE")) P.flat_min_idle = (uint32_t)std::max(1, std::atoi(e));
    P.caps[0] = cap_e; P.caps[1] = cap_n; P.caps[2] = cap_m;
    // resident threads = scratch slabs: as many as the kernel's register footprint lets the device hold
    uint64_t per_cu = 1024;          // 16 wavefronts per CU: the kernel is built for at most 128 VGPRs (__launch_bounds__(64, 4))
    if (const char* e = std::getenv("VGAMD_GAPLESS_THREADS_PER_CU")) per_cu = (uint64_t)std::max(64, std::atoi(e));
    const uint32_t threads = (uint32_t)std::min<uint64_t>(n, (uint64_t)std::max(1, be->compute_units()) * per_cu);
    { vgk_ctx::DevBuf& b = ctx->scratch[15];
      const uint64_t want = sizeof(GScratch) * (uint64_t)threads;
      (void)b; P.scratch = (GScratch*)ctx->ensure_scratch(15, want);
      P.cold = (GCold*)ctx->ensure_scratch(30, sizeof(GCold) * (uint64_t)threads); }
    P.results = (vgk_gapless_result*)dev(nullptr, sizeof(vgk_gapless_result) * n);
    P.ext = (vgk_extension*)dev(nullptr, sizeof(vgk_extension) * cap_e);
    P.nodes = (uint32_t*)dev(nullptr, sizeof(uint32_t) * cap_n);
    P.mism = (uint32_t*)dev(nullptr, sizeof(uint32_t) * cap_m);
    P.counters = (unsigned long long*)dev(nullptr, 256);
    P.retry = (uint8_t*)ctx->ensure_scratch(60, (size_t)n + 16);        // (a slot of its own: 31 is the banded k-best's score planes)
    P.winners = (GExt*)dev(nullptr, sizeof(GExt) * (n_seed + 1));          // the searches' winners wait here for the rules kernel (248 B each; only the used ones are touched)
    if (!P.winners || !P.probs || !P.reads || !P.seeds || !P.order || !P.scratch || !P.cold || !P.results || !P.ext || !P.nodes || !P.mism || !P.counters) return cleanup(VGK_ENOMEM);
    int rc;
    if ((rc = be->zero(P.counters, 256))) return cleanup(rc);
    if (!P.retry || (rc = be->zero(P.retry, (size_t)n + 16))) return cleanup(rc ? rc : VGK_ENOMEM);
    lap("order, uploads queued");
    if ((rc = be->run_gapless(P, threads))) return cleanup(rc);
    lap("uploads + kernels");
    ctx->gapless_last = P; ctx->gapless_last_threads = threads; ctx->gapless_last_valid = true;
    unsigned long long counters[6] = {0, 0, 0, 0, 0, 0};
    if ((rc = be->download(counters, P.counters, sizeof counters))) return cleanup(rc);
    ctx->gapless_ms = be->last_ms(5); ctx->gapless_retried = counters[3]; ctx->gapless_redone = counters[5];
#if defined(VGAMD_GAPLESS_PROF)
    { unsigned long long sec[20] = {0}; be->download(sec, P.counters + 8, sizeof(unsigned long long) * 12);
      std::fprintf(stderr, "gapless sections (wave cycles):"); for (int i = 0; i < 12; ++i) std::fprintf(stderr, " [%d]=%llu", i, sec[i]); std::fprintf(stderr, "\n"); }
#endif
    const uint64_t ne = std::min<uint64_t>(counters[0], cap_e), nn = std::min<uint64_t>(counters[1], cap_n), nm = std::min<uint64_t>(counters[2], cap_m);
    // the device packed the sets in completion order; they go back in problem order: sizes, prefix sums and the gather on the device
    // (gapless_device.hpp), then three contiguous arrays come down through page-locked staging
    GOrderParams O{};
    O.n = n; O.res = P.results; O.ext = P.ext; O.nodes = P.nodes; O.mism = P.mism;
    uint32_t* tab = (uint32_t*)ctx->ensure_scratch(26, sizeof(uint32_t) * 6 * ((size_t)n + 1));
    O.res_out = (vgk_gapless_result*)ctx->ensure_scratch(27, sizeof(vgk_gapless_result) * (size_t)n);
    O.ext_out = (vgk_extension*)ctx->ensure_scratch(28, sizeof(vgk_extension) * (ne + 1));
    O.nodes_out = (uint32_t*)ctx->ensure_scratch(29, sizeof(uint32_t) * (nn + nm + 2));
    O.read_of = (uint32_t*)ctx->ensure_scratch(59, sizeof(uint32_t) * (ne + 1));
    if (!tab || !O.res_out || !O.ext_out || !O.nodes_out || !O.read_of) return cleanup(VGK_ENOMEM);
    O.mism_out = O.nodes_out + nn + 1;
    const size_t n1 = (size_t)n + 1;
    O.size_e = tab; O.size_n = tab + n1; O.size_m = tab + 2 * n1; O.off_e = tab + 3 * n1; O.off_n = tab + 4 * n1; O.off_m = tab + 5 * n1;
    if ((rc = be->zero(tab, sizeof(uint32_t) * 3 * n1))) return cleanup(rc);
    if ((rc = be->gapless_order(O, 1))) return cleanup(rc);
    for (int k = 0; k < 3; ++k) if ((rc = be->scan_u32(tab + k * n1, tab + (3 + k) * n1, (uint32_t)n1))) return cleanup(rc);
    if ((rc = be->gapless_order(O, 2))) return cleanup(rc);
    uint32_t tot[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) if ((rc = be->download(&tot[k], tab + (3 + k) * n1 + n, sizeof(uint32_t)))) return cleanup(rc);
    const size_t we = tot[0], wn = tot[1], wm = tot[2];
    ctx->sets.valid = true; ctx->sets.n = n; ctx->sets.n_ext = tot[0]; ctx->sets.probs = P.probs; ctx->sets.reads = P.reads;
    ctx->sets.res = O.res_out; ctx->sets.ext = O.ext_out; ctx->sets.nodes = O.nodes_out; ctx->sets.index = nullptr; ctx->sets.read_of = O.read_of;
    int rc_all = VGK_OK;
    vgk_gapless_result* dres = H.dres.get(be, n);
    vgk_extension* dext = H.dext.get(be, we + 1); uint32_t* dnodes = H.dnodes.get(be, wn + 1); uint32_t* dmism = H.dmism.get(be, wm + 1);
    if (!dres || !dext || !dnodes || !dmism) return cleanup(VGK_ENOMEM);
    const bool fits = we <= ext_cap && wn <= nodes_cap && wm <= mism_cap && (we == 0 || (extensions && nodes && mismatches));
    if (fits && defer) {
        // VGK_GAPLESS_DEFER: the copies down are queued on the fetch stream behind the gather kernels and run while the caller's next
        // call (vgk_tail_stage*) keeps the device busy; that call copies them out of the staging buffers when it is done
        // (queued by vgk_tail_stage just before its fill kernels — VALU-bound, indifferent to the DMA; the small kernels before them ran
        // twice as long beside the copies — or by whoever finishes the deferral first)
        ctx->deferred.from[0] = O.res_out; ctx->deferred.from[1] = O.ext_out; ctx->deferred.from[2] = O.nodes_out; ctx->deferred.from[3] = O.mism_out;
        ctx->deferred.queued = false;
        ctx->deferred.spans[0] = {(char*)results, (const char*)dres, sizeof(vgk_gapless_result) * n};
        ctx->deferred.spans[1] = {(char*)extensions, (const char*)dext, sizeof(vgk_extension) * we};
        ctx->deferred.spans[2] = {(char*)nodes, (const char*)dnodes, sizeof(uint32_t) * wn};
        ctx->deferred.spans[3] = {(char*)mismatches, (const char*)dmism, sizeof(uint32_t) * wm};
        ctx->deferred.pending = true;
        lap("ordered on the device, downloads queued");
        if (written) { written[0] = we; written[1] = wn; written[2] = wm; }
        return cleanup(VGK_OK);
    }
    if ((rc = be->download(dres, O.res_out, sizeof(vgk_gapless_result) * n))) return cleanup(rc);
    if (fits) {
        if (we && (rc = be->download(dext, O.ext_out, sizeof(vgk_extension) * we))) return cleanup(rc);
        if (wn && (rc = be->download(dnodes, O.nodes_out, sizeof(uint32_t) * wn))) return cleanup(rc);
        if (wm && (rc = be->download(dmism, O.mism_out, sizeof(uint32_t) * wm))) return cleanup(rc);
        lap("ordered on the device, downloads");
        // out of the staging buffers in slices, on the host threads
        struct Span { char* dst; const char* src; size_t bytes; };
        const Span spans[4] = {{(char*)results, (const char*)dres, sizeof(vgk_gapless_result) * n}, {(char*)extensions, (const char*)dext, sizeof(vgk_extension) * we},
                               {(char*)nodes, (const char*)dnodes, sizeof(uint32_t) * wn}, {(char*)mismatches, (const char*)dmism, sizeof(uint32_t) * wm}};
        std::vector<Span> slices;
        for (const Span& sp : spans) for (size_t at = 0; at < sp.bytes; at += (size_t)8 << 20) slices.push_back({sp.dst + at, sp.src + at, std::min<size_t>((size_t)8 << 20, sp.bytes - at)});
        parallel_tasks((uint32_t)slices.size(), [&](uint32_t k) { std::memcpy(slices[k].dst, slices[k].src, slices[k].bytes); });
    } else {
        // the caller's arrays are too small: the reads whose sets fit behind each other get theirs, the others VGK_EOPS (rare; on the host)
        if (we && (rc = be->download(dext, O.ext_out, sizeof(vgk_extension) * we))) return cleanup(rc);
        if (wn && (rc = be->download(dnodes, O.nodes_out, sizeof(uint32_t) * wn))) return cleanup(rc);
        if (wm && (rc = be->download(dmism, O.mism_out, sizeof(uint32_t) * wm))) return cleanup(rc);
        uint64_t ae = 0, an = 0, am = 0;
        for (uint32_t i = 0; i < n; ++i) {
            vgk_gapless_result r = dres[i];
            if (r.status == VGK_OK) {
                uint64_t need_n = 0, need_m = 0;
                for (uint32_t k = 0; k < r.n_ext; ++k) { need_n += dext[r.ext_begin + k].path_len; need_m += dext[r.ext_begin + k].n_mismatches; }
                if (ae + r.n_ext > ext_cap || an + need_n > nodes_cap || am + need_m > mism_cap || !extensions || !nodes || !mismatches) { r.status = VGK_EOPS; r.n_ext = 0; rc_all = VGK_EOPS; }
                else {
                    for (uint32_t k = 0; k < r.n_ext; ++k) {
                        vgk_extension x = dext[r.ext_begin + k];
                        std::memcpy(nodes + an, dnodes + x.path_begin, sizeof(uint32_t) * x.path_len);
                        std::memcpy(mismatches + am, dmism + x.mism_begin, sizeof(uint32_t) * x.n_mismatches);
                        x.path_begin = (uint32_t)an; x.mism_begin = (uint32_t)am; an += x.path_len; am += x.n_mismatches;
                        extensions[ae + k] = x;
                    }
                    const uint32_t begin = (uint32_t)ae; ae += r.n_ext; r.ext_begin = begin;
                }
            }
            if (r.status != VGK_OK) r.ext_begin = (uint32_t)ae;
            results[i] = r;
        }
        if (written) { written[0] = ae; written[1] = an; written[2] = am; }
        return cleanup(rc_all);
    }
    lap("sets copied out");
    if (written) { written[0] = we; written[1] = wn; written[2] = wm; }
    return cleanup(rc_all);
}


int vgk_gapless_fetch_deferred(vgk_ctx* ctx) try {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ctx->finish_deferred();
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_gapless_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_gapless_problem* problems, uint32_t n,
                       vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                       uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3]) try {
    if (!ctx || !index || !vgk_tables_usable(index->ctx, ctx) || (!problems && n) || (!results && n)) return VGK_EINVAL;
    if (written) written[0] = written[1] = written[2] = 0;
    if (!n) return VGK_OK;
    std::lock_guard<std::mutex> stage(ctx->stage_mu);
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->gapless_last_valid = false; ctx->sets.valid = false;
    Backend* be = ctx->be.get();
    GLap lap;
    // pack: masked reads (ReadMasker, src/gbwt_extender.cpp:160-176), seeds
    std::vector<GProb> probs(n);
    uint64_t n_read = 0, n_seed = 0;
    // a batch caller's reads (and seeds) usually lie behind each other in one buffer, in problem order: then they go up as they are,
    // straight from the caller's memory; otherwise they are gathered into page-locked staging first
    bool reads_in_a_row = true, seeds_in_a_row = true; const char* read0 = nullptr; const vgk_seed* seed0 = nullptr;
    {   // in slices on the host threads: sizes per slice, their prefix sums, then the descriptors
        const uint32_t slices = std::min<uint32_t>(64u, (n + 16383u) / 16384u);
        std::vector<uint64_t> sr((size_t)slices + 1, 0), ss((size_t)slices + 1, 0); std::vector<int> bad(slices, 0);
        auto lo_of = [&](uint32_t c) { return (uint32_t)((uint64_t)n * c / slices); };
        parallel_tasks(slices, [&](uint32_t c) {
            uint64_t r = 0, s = 0;
            for (uint32_t i = lo_of(c); i < lo_of(c + 1); ++i) { const vgk_gapless_problem& p = problems[i]; if ((p.read_len && !p.read) || (p.n_seeds && !p.seeds)) bad[c] = 1; r += p.read_len; s += p.n_seeds; }
            sr[c + 1] = r; ss[c + 1] = s;
        });
        for (uint32_t c = 0; c < slices; ++c) { if (bad[c]) return VGK_EINVAL; sr[c + 1] += sr[c]; ss[c + 1] += ss[c]; }
        n_read = sr[slices]; n_seed = ss[slices];
        if (n_read > 0xfffffff0ull || n_seed > 0xfffffff0ull) return VGK_ETOOBIG;
        for (uint32_t i = 0; i < n && (!read0 || !seed0); ++i) {               // where the first read / seed lies (usually problem 0)
            if (!read0 && problems[i].read_len) read0 = problems[i].read;      // (valid as the common base only if everything before it is empty: checked below)
            if (!seed0 && problems[i].n_seeds) seed0 = problems[i].seeds;
        }
        std::vector<int> gap_r(slices, 0), gap_s(slices, 0);
        parallel_tasks(slices, [&](uint32_t c) {
            uint64_t r = sr[c], s = ss[c];
            for (uint32_t i = lo_of(c); i < lo_of(c + 1); ++i) {
                const vgk_gapless_problem& p = problems[i];
                probs[i] = {(uint32_t)r + 8, p.read_len, (uint32_t)s, p.n_seeds, p.max_mismatches, p.flags & ~(uint32_t)VGK_GAPLESS_DEFER, p.overlap_threshold};
                if (p.read_len && p.read != read0 + r) gap_r[c] = 1;
                if (p.n_seeds && p.seeds != seed0 + s) gap_s[c] = 1;
                r += p.read_len; s += p.n_seeds;
            }
        });
        for (uint32_t c = 0; c < slices; ++c) { if (gap_r[c]) reads_in_a_row = false; if (gap_s[c]) seeds_in_a_row = false; }
    }
    if (std::getenv("VGAMD_GAPLESS_GATHER")) reads_in_a_row = seeds_in_a_row = false;
    if (!ctx->gapless_host) ctx->gapless_host = std::make_shared<GaplessHost>();
    GaplessHost& H = *static_cast<GaplessHost*>(ctx->gapless_host.get());
    char* reads = reads_in_a_row ? nullptr : H.reads.get(be, n_read + 16); vgk_seed* seeds = seeds_in_a_row ? nullptr : H.seeds.get(be, n_seed + 1);
    if ((!reads_in_a_row && !reads) || (!seeds_in_a_row && !seeds)) return VGK_ENOMEM;
    if (!reads_in_a_row || !seeds_in_a_row) parallel_for(n, [&](uint32_t i, unsigned) {      // (the masking itself happens on the device, over the uploaded bytes)
        const vgk_gapless_problem& p = problems[i];
        if (reads && p.read_len) std::memcpy(reads + probs[i].read_off, p.read, p.read_len);
        if (seeds && p.n_seeds) std::memcpy(seeds + probs[i].seed_off, p.seeds, sizeof(vgk_seed) * p.n_seeds);
    });
    lap("descriptors; reads and seeds gathered unless they lie in a row");
    // device buffers are kept on the context between calls (grow-only)
    int next_slot = 16;
    auto cleanup = [&](int rc) { return rc; };
    auto dev = [&](const void* src, size_t bytes) -> void* {
        void* d = ctx->ensure_scratch(next_slot++, std::max<size_t>(bytes, 16)); if (!d) return nullptr;
        if (src && bytes && be->upload(d, src, bytes)) return nullptr;
        return d;
    };
    GaplessParams P{};
    { const bool runs = index->merged && index->search_merged;
      P.index = runs ? index->merged->dev : index->dev; P.orig = index->dev; P.merge = runs ? index->merge : GMerge{}; P.n = n; }
    P.probs = (const GProb*)dev(probs.data(), sizeof(GProb) * n);
    P.reads = (const char*)dev(nullptr, n_read + 16);                  // 8 bytes of padding at either end
    if (P.reads && n_read && be->upload(const_cast<char*>(P.reads) + 8, reads_in_a_row ? read0 : reads + 8, n_read)) return VGK_ENODEV;
    if (P.reads && (be->mask_reads(const_cast<char*>(P.reads), n_read + 16) || be->zero(const_cast<char*>(P.reads), 8) || be->zero(const_cast<char*>(P.reads) + 8 + n_read, 8))) return VGK_ENODEV;
    P.seeds = (const vgk_seed*)dev(nullptr, sizeof(vgk_seed) * (n_seed + 1));
    if (P.seeds && n_seed && be->upload(const_cast<vgk_seed*>(P.seeds), seeds_in_a_row ? seed0 : seeds, sizeof(vgk_seed) * n_seed)) return VGK_ENODEV;
    // processing order: by the node of the first seed (a counting sort; reads without seeds last).  Results do not depend on it —
    // problems are independent and the sets are handed back in problem order below — but reads that sit next to each other in a
    // wavefront now walk the same records and bases, which the L2 then serves (FETCH_SIZE per million reads: see DESIGN.md §11)
    // (made on the device from the uploaded descriptors and seeds: a kernel for the keys and a stable radix sort, as for the seeded form)
    if (!P.probs || !P.reads || !P.seeds) return VGK_ENOMEM;
    uint32_t* d_sort = (uint32_t*)ctx->ensure_scratch(next_slot++, sizeof(uint32_t) * 4 * (size_t)n);       // key, index, sorted key, order
    if (!d_sort) return VGK_ENOMEM;
    if (std::getenv("VGAMD_GAPLESS_UNSORTED")) {
        std::vector<uint32_t> order(n); for (uint32_t i = 0; i < n; ++i) order[i] = i;
        if (be->upload(d_sort + 3 * (size_t)n, order.data(), sizeof(uint32_t) * n) || be->sync()) return VGK_ENODEV;
    } else {
        const uint32_t buckets = index->n_oriented / 2 + 2;
        int bits = 1; while ((1u << bits) < buckets && bits < 32) ++bits;
        GSeededParams S{};
        S.n = n; S.read_off = nullptr; S.seeds = P.seeds; S.buckets = buckets; S.probs = const_cast<GProb*>(P.probs); S.key = d_sort; S.idx = d_sort + n;
        int rc = be->gapless_seeded(S);
        if (!rc) rc = be->sort_pairs_u32(d_sort, d_sort + 2 * (size_t)n, d_sort + n, d_sort + 3 * (size_t)n, n, bits);
        if (rc) return rc;
    }
    P.order = d_sort + 3 * (size_t)n;
    return gapless_run_and_fetch(ctx, P, n, n_seed, next_slot, H, lap, results, extensions, ext_cap, nodes, nodes_cap, mismatches, mism_cap, written,
                                 (problems[0].flags & VGK_GAPLESS_DEFER) != 0);        // (a property of the call, carried by its first problem)
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

// The clusters of the last vgk_minimizer_seeds call, extended without leaving the device in between: the reads it uploaded (masked,
// padded), the seeds it found and their offsets per read are still in HBM; the problem descriptors and the hand-out order are made
// there too (a kernel + a radix sort).  One setting of max_mismatches / overlap_threshold / flags for the whole batch.
int vgk_gapless_extend_seeded(vgk_ctx* ctx, const vgk_haplo* index, uint32_t max_mismatches, double overlap_threshold, uint32_t flags,
                              vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                              uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3]) try {
    if (!ctx || !index || !vgk_tables_usable(index->ctx, ctx)) return VGK_EINVAL;
    if (written) written[0] = written[1] = written[2] = 0;
    std::lock_guard<std::mutex> stage(ctx->stage_mu);
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->seeded.valid || ctx->seeded.graph != index) return VGK_EINVAL;
    const uint32_t n = ctx->seeded.n;
    if (!n) return VGK_OK;
    if (!results) return VGK_EINVAL;
    ctx->gapless_last_valid = false; ctx->sets.valid = false;
    Backend* be = ctx->be.get();
    GLap lap;
    if (!ctx->gapless_host) ctx->gapless_host = std::make_shared<GaplessHost>();
    GaplessHost& H = *static_cast<GaplessHost*>(ctx->gapless_host.get());
    const uint64_t n_seed = ctx->seeded.n_seeds;
    int next_slot = 16;
    GProb* d_probs = (GProb*)ctx->ensure_scratch(next_slot++, sizeof(GProb) * (size_t)n);
    uint32_t* d_sort = (uint32_t*)ctx->ensure_scratch(next_slot++, sizeof(uint32_t) * 4 * (size_t)n);      // key, index, sorted key, order
    if (!d_probs || !d_sort) return VGK_ENOMEM;
    const uint32_t buckets = index->n_oriented / 2 + 2;
    int bits = 1; while ((1u << bits) < buckets && bits < 32) ++bits;
    GSeededParams S{};
    S.n = n; S.read_off = ctx->seeded.read_off; S.seed_off = ctx->seeded.seed_off; S.seeds = ctx->seeded.seeds;
    S.max_mm = max_mismatches; S.flags = flags & ~(uint32_t)VGK_GAPLESS_DEFER; S.overlap = overlap_threshold; S.buckets = buckets;
    S.probs = d_probs; S.key = d_sort; S.idx = d_sort + n;
    int rc = be->gapless_seeded(S);
    if (!rc) rc = be->sort_pairs_u32(d_sort, d_sort + 2 * (size_t)n, d_sort + n, d_sort + 3 * (size_t)n, n, bits);
    if (rc) return rc;
    GaplessParams P{};
    { const bool runs = index->merged && index->search_merged;
      P.index = runs ? index->merged->dev : index->dev; P.orig = index->dev; P.merge = runs ? index->merge : GMerge{}; P.n = n; }
    P.probs = d_probs; P.reads = ctx->seeded.reads; P.seeds = ctx->seeded.seeds; P.order = d_sort + 3 * (size_t)n;
    lap("descriptors and order on the device");
    return gapless_run_and_fetch(ctx, P, n, n_seed, next_slot, H, lap, results, extensions, ext_cap, nodes, nodes_cap, mismatches, mism_cap, written, (flags & VGK_GAPLESS_DEFER) != 0);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_gapless_rerun(vgk_ctx* ctx) try {
    if (!ctx) return VGK_EINVAL;
    std::lock_guard<std::mutex> stage(ctx->stage_mu);
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->gapless_last_valid) return VGK_EINVAL;
    int rc;
    if ((rc = ctx->be->zero(ctx->gapless_last.counters, 256))) return rc;
    if ((rc = ctx->be->run_gapless(ctx->gapless_last, ctx->gapless_last_threads))) return rc;
    ctx->gapless_ms = ctx->be->last_ms(5);
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_gapless_last_ms(vgk_ctx* ctx) { return ctx ? ctx->gapless_ms : 0.0; }
uint64_t vgk_gapless_last_retried(vgk_ctx* ctx) { return ctx ? ctx->gapless_retried : 0; }
uint64_t vgk_gapless_last_redone(vgk_ctx* ctx) { return ctx ? ctx->gapless_redone : 0; }
uint64_t vgk_haplo_run_nodes(const vgk_haplo* index) {      // (asks about the merged form: builds it when it is still pending)
    if (!index) return 0;
    if (index->pending_merge) vgk_haplo_ensure_merged(const_cast<vgk_haplo*>(index));
    return index->merged ? index->merged->n_oriented / 2 : index->n_oriented / 2;
}
uint64_t vgk_haplo_search_nodes(const vgk_haplo* index) { return index ? (index->merged && index->search_merged ? index->merged->n_oriented / 2 : index->n_oriented / 2) : 0; }

}  // extern "C"
