// minimizer_api.cpp — vgk_minimizer_index_create / vgk_minimizer_seeds (minimizer_device.hpp): the minimizer index of the haplotype
// threads built on host threads and kept in HBM; the reads' minimizers, lookups and seeds on the device.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>
#include "ctx.hpp"
#include "haplo.hpp"
#include "host_parallel.hpp"
#include "minimizer_device.hpp"

using namespace vgk;

struct vgk_minimizer_index {
    vgk_ctx* ctx = nullptr;
    MzIndex dev{};
    std::vector<void*> held;
    uint64_t n_keys = 0, n_pos = 0;
    std::vector<vgk_minimizer_hit> hits;      // what the index holds (vgk_minimizer_index_fetch)
    MzPolicy policy{};                        // vgk_minimizer_set_policy (on == 0: none); its score table in `policy_tab` (device)
    void* policy_tab = nullptr;
};

namespace {
char mz_comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }
struct Entry { uint64_t key, hash; uint32_t node, offset; };
}

extern "C" {

int vgk_minimizer_index_create(vgk_ctx* ctx, const vgk_haplotypes* d, uint32_t k, uint32_t w, vgk_minimizer_index** out) try {
    if (!ctx || !d || !out || !d->n_nodes || !d->node_len || !d->seq || (d->n_threads && (!d->thread_off || !d->thread_nodes))) return VGK_EINVAL;
    if (k == 0 || k > MZ_MAX_K || w == 0 || w > MZ_MAX_W) return VGK_EINVAL;
    *out = nullptr;
    const uint32_t N = d->n_nodes;
    std::vector<uint64_t> node_at((size_t)N + 1, 0);
    for (uint32_t i = 0; i < N; ++i) node_at[i + 1] = node_at[i] + d->node_len[i];
    for (uint32_t t = 0; t < d->n_threads; ++t) for (uint32_t x = d->thread_off[t]; x < d->thread_off[t + 1]; ++x) if (d->thread_nodes[x] >= 2 * N) return VGK_EINVAL;
    for (uint32_t i = 0; i < N; ++i) if (d->node_len[i] > 65535u) return VGK_ETOOBIG;      // (a position's offset word holds the offset from either end in 16 bits each)
    // the minimizers of every thread, read along the thread; a reverse-canonical one is filed under the position its
    // reverse complement starts at: the k-mer's last base, seen from the other strand
    std::vector<std::vector<Entry>> per_thread(d->n_threads);
    parallel_tasks(d->n_threads, [&](uint32_t t) {
        const uint32_t* tn = d->thread_nodes + d->thread_off[t]; const uint32_t len = d->thread_off[t + 1] - d->thread_off[t];
        size_t bases = 0;
        for (uint32_t x = 0; x < len; ++x) bases += d->node_len[tn[x] >> 1];
        std::vector<char> seq(bases); std::vector<uint32_t> step(bases);          // step[b] = index in the thread of the node that holds base b
        std::vector<uint64_t> start((size_t)len + 1, 0);
        size_t at = 0;
        for (uint32_t x = 0; x < len; ++x) {
            const uint32_t o = tn[x], v = o >> 1, L = d->node_len[v]; const char* s = d->seq + node_at[v];
            start[x] = at;
            for (uint32_t b = 0; b < L; ++b) { seq[at] = (o & 1) ? mz_comp(s[L - 1 - b]) : s[b]; step[at] = x; ++at; }
        }
        start[len] = at;
        std::vector<Entry>& outv = per_thread[t];
        mz_minimizers(seq.data(), (uint32_t)std::min<size_t>(bases, 0xffffffffu), k, w, [&](uint32_t p, const MzKmer& m) {
            Entry e; e.key = m.key; e.hash = m.hash;
            if (!m.reverse) { const uint32_t x = step[p]; e.node = tn[x]; e.offset = (uint32_t)(p - start[x]); }
            else { const size_t q = (size_t)p + k - 1; const uint32_t x = step[q]; const uint32_t L = d->node_len[tn[x] >> 1]; e.node = tn[x] ^ 1u; e.offset = L - 1 - (uint32_t)(q - start[x]); }
            outv.push_back(e);
        });
    });
    std::vector<Entry> all;
    for (auto& v : per_thread) { all.insert(all.end(), v.begin(), v.end()); std::vector<Entry>().swap(v); }
    std::sort(all.begin(), all.end(), [](const Entry& a, const Entry& b) { return a.key != b.key ? a.key < b.key : a.node != b.node ? a.node < b.node : a.offset < b.offset; });
    all.erase(std::unique(all.begin(), all.end(), [](const Entry& a, const Entry& b) { return a.key == b.key && a.node == b.node && a.offset == b.offset; }), all.end());
    uint64_t n_keys = 0;
    for (size_t i = 0; i < all.size(); ++i) if (i == 0 || all[i].key != all[i - 1].key) ++n_keys;
    uint64_t cap = 16; while (cap < 2 * n_keys + 2) cap <<= 1;
    if (cap > (1ull << 31)) return VGK_ETOOBIG;
    std::vector<MzSlot> slots(cap, MzSlot{0, 0, 0});
    std::vector<MzPos> pos(all.size() + 1);
    for (size_t i = 0; i < all.size();) {
        size_t j = i; while (j < all.size() && all[j].key == all[i].key) { pos[j] = MzPos{all[j].node, mz_pack_offset(all[j].offset, d->node_len[all[j].node >> 1])}; ++j; }
        uint32_t s = (uint32_t)all[i].hash & (uint32_t)(cap - 1);
        while (slots[s].count) s = (s + 1) & (uint32_t)(cap - 1);
        // a key with one position keeps it in its slot (one request per lookup); its offset word must leave the flag bit free
        if (j - i == 1 && !(pos[i].offset & MZ_INLINE)) slots[s] = MzSlot{all[i].key, pos[i].node, pos[i].offset | MZ_INLINE};
        else slots[s] = MzSlot{all[i].key, (uint32_t)i, (uint32_t)(j - i)};
        i = j;
    }
    std::unique_ptr<vgk_minimizer_index> ix(new (std::nothrow) vgk_minimizer_index());
    if (!ix) return VGK_ENOMEM;
    ix->ctx = ctx; ix->n_keys = n_keys; ix->n_pos = all.size();
    ix->hits.resize(all.size());
    for (size_t i = 0; i < all.size(); ++i) ix->hits[i] = vgk_minimizer_hit{all[i].key, all[i].node, all[i].offset};
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    void* ds = be->alloc(sizeof(MzSlot) * cap); void* dp = be->alloc(sizeof(MzPos) * pos.size());
    if (ds) ix->held.push_back(ds);
    if (dp) ix->held.push_back(dp);
    int rc = (ds && dp) ? VGK_OK : VGK_ENOMEM;
    if (!rc) rc = be->upload(ds, slots.data(), sizeof(MzSlot) * cap);
    if (!rc) rc = be->upload(dp, pos.data(), sizeof(MzPos) * pos.size());
    if (!rc) rc = be->sync();
    if (rc) { for (void* p : ix->held) be->release(p); return rc; }
    ix->dev.slots = (const MzSlot*)ds; ix->dev.mask = (uint32_t)(cap - 1); ix->dev.pos = (const MzPos*)dp; ix->dev.k = k; ix->dev.w = w;
    *out = ix.release();
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

void vgk_minimizer_index_destroy(vgk_minimizer_index* ix) {
    if (!ix) return;
    { std::lock_guard<std::mutex> lk(ix->ctx->mu); ix->ctx->be->sync(); for (void* p : ix->held) ix->ctx->be->release(p); if (ix->policy_tab) ix->ctx->be->release(ix->policy_tab); }
    delete ix;
}
int vgk_minimizer_set_policy(vgk_minimizer_index* ix, const vgk_seed_policy* policy) try {
    if (!ix) return VGK_EINVAL;
    std::lock_guard<std::mutex> lk(ix->ctx->mu);
    Backend* be = ix->ctx->be.get();
    if (!policy) { ix->policy = MzPolicy{}; return VGK_OK; }
    if (!policy->hard_hit_cap || policy->hard_hit_cap > 65535u || !(policy->minimizer_score_fraction >= 0.0 && policy->minimizer_score_fraction <= 1.0)) return VGK_EINVAL;
    // find_minimizers' scores by hit count, with the host's logarithm (src/minimizer_mapper.cpp:3927-3937)
    std::vector<double> tab((size_t)policy->hard_hit_cap + 1, 0.0);
    const double base = 1.0 + std::log((double)policy->hard_hit_cap);
    for (uint32_t h = 1; h <= policy->hard_hit_cap; ++h) tab[h] = base - std::log((double)h);
    int rc = be->sync();
    if (rc) return rc;
    if (ix->policy_tab) { be->release(ix->policy_tab); ix->policy_tab = nullptr; ix->policy = MzPolicy{}; }
    void* d = be->alloc(sizeof(double) * tab.size());
    if (!d) return VGK_ENOMEM;
    if ((rc = be->upload(d, tab.data(), sizeof(double) * tab.size())) || (rc = be->sync())) { be->release(d); return rc; }
    ix->policy_tab = d;
    ix->policy.on = 1; ix->policy.hit_cap = policy->hit_cap; ix->policy.hard_hit_cap = policy->hard_hit_cap; ix->policy.fraction = policy->minimizer_score_fraction; ix->policy.paired = policy->paired;
    ix->policy.tab = (const double*)d;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }
uint64_t vgk_minimizer_index_keys(const vgk_minimizer_index* ix) { return ix ? ix->n_keys : 0; }
uint64_t vgk_minimizer_index_hits(const vgk_minimizer_index* ix) { return ix ? ix->hits.size() : 0; }
int vgk_minimizer_index_fetch(const vgk_minimizer_index* ix, vgk_minimizer_hit* hits, size_t cap) try {
    if (!ix || (!hits && cap)) return VGK_EINVAL;
    if (cap < ix->hits.size()) return VGK_EOPS;
    if (!ix->hits.empty()) std::memcpy(hits, ix->hits.data(), sizeof(vgk_minimizer_hit) * ix->hits.size());
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

int vgk_minimizer_seeds(vgk_ctx* ctx, const vgk_minimizer_index* ix, const vgk_haplo* graph, const char* reads, const uint64_t* read_off, uint32_t n,
                        uint32_t hit_cap, uint32_t* seed_off, uint32_t* minimizers, vgk_seed* seeds, size_t seeds_cap, size_t* written) try {
    if (!ctx || !ix || !graph || !vgk_tables_usable(ix->ctx, ctx) || !vgk_tables_usable(graph->ctx, ctx) || (n && (!reads || !read_off || !seed_off))) return VGK_EINVAL;
    if (written) *written = 0;
    if (!n) { if (seed_off) seed_off[0] = 0; return VGK_OK; }
    for (uint32_t i = 0; i < n; ++i) if (read_off[i + 1] < read_off[i]) return VGK_EINVAL;
    const uint64_t bytes = read_off[n] - read_off[0];
    if (bytes > 0xfffffff0ull) return VGK_ETOOBIG;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> stage(ctx->stage_mu);
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->seeded.valid = false;
    char* d_reads = (char*)ctx->ensure_scratch(55, bytes + 32);              // 8 bytes of padding at either end, as the extension kernels want them
    uint64_t* d_off = (uint64_t*)ctx->ensure_scratch(56, sizeof(uint64_t) * ((size_t)n + 1));
    uint32_t* d_tab = (uint32_t*)ctx->ensure_scratch(57, sizeof(uint32_t) * 3 * ((size_t)n + 1));
    if (!d_reads || !d_off || !d_tab) return VGK_ENOMEM;
    const size_t n1 = (size_t)n + 1;
    std::vector<uint64_t> rel(n1);
    for (size_t i = 0; i < n1; ++i) rel[i] = read_off[i] - read_off[0];
    MinimizerParams P{};
    P.index = ix->dev; P.graph = graph->dev; P.reads = d_reads + 8; P.read_off = d_off; P.n = n; P.hit_cap = hit_cap ? hit_cap : 0xffffffffu;
    P.policy = ix->policy;
    if (P.policy.on) P.hit_cap = P.policy.hard_hit_cap;                     // (what a read beyond the selection's 64 minimizers is held to)
    P.counts = d_tab; P.mins = d_tab + n1; P.first = d_tab + 2 * n1;
    int rc = be->upload(d_off, rel.data(), sizeof(uint64_t) * n1);
    if (!rc) rc = be->zero(d_tab, sizeof(uint32_t) * 3 * n1);
    if (!rc) rc = be->zero(d_reads, 8);
    if (!rc) rc = be->zero(d_reads + 8 + bytes, 8);
    be->watch(0);                                                          // (the stopwatch covers copies, kernels and scans from here on)
    // The reads go up in slices on the copy stream; a slice is masked (ReadMasker once, for the seeding — anything but ACGT is no k-mer —
    // and the extension) and its minimizers found on the main stream while the next slice travels.
    P.pass = 1;
    const uint32_t slices = n >= (1u << 16) ? 4u : 1u;
    for (uint32_t c = 0; c < slices && !rc; ++c) {
        const uint32_t lo = (uint32_t)((uint64_t)n * c / slices), hi = (uint32_t)((uint64_t)n * (c + 1) / slices);
        const uint64_t a = rel[lo], b = rel[hi];
        if (b > a) { rc = be->upload_side(d_reads + 8 + a, reads + read_off[0] + a, b - a); if (!rc) rc = be->sync_side(); if (!rc) { const uint64_t from = (8 + a) & ~15ull; rc = be->mask_reads(d_reads + from, 8 + b - from); } }      // (from a 16-byte boundary: the few bytes before `a` are the previous slice's, masked already — masking is idempotent)
        P.lo = lo; P.hi = hi;
        if (!rc) rc = be->run_minimizer(P);
    }
    if (!rc) rc = be->scan_u32(d_tab, d_tab + 2 * n1, (uint32_t)n1);
    if (!rc) rc = be->download(seed_off, d_tab + 2 * n1, sizeof(uint32_t) * n1);      // synchronises (rel[] may go)
    if (!rc && minimizers) rc = be->download(minimizers, d_tab + n1, sizeof(uint32_t) * n);
    if (rc) return rc;
    const size_t total = seed_off[n];
    if (written) *written = total;
    const bool keep_on_device = !seeds && !seeds_cap;                         // the caller goes on with vgk_gapless_extend_seeded: the seeds need not come down
    if (!keep_on_device && (total > seeds_cap || (total && !seeds))) return VGK_EOPS;
    // (the seed buffer exists even for a batch without a single seed: the seeded extension call tells "no seeds" from "no memory" by it)
    vgk_seed* d_seeds = (vgk_seed*)ctx->ensure_scratch(58, sizeof(vgk_seed) * std::max<size_t>(total, 1));
    if (!d_seeds) return VGK_ENOMEM;
    P.seeds = d_seeds;
    if (total) {
        P.pass = 2;
        rc = be->run_minimizer(P);
        be->watch(1);
        if (!rc) rc = keep_on_device ? be->sync() : be->download(seeds, d_seeds, sizeof(vgk_seed) * total);
        if (rc) return rc;
    } else { be->watch(1); be->sync(); }
    ctx->minimizer_ms = be->watch_ms();
    ctx->seeded.valid = true; ctx->seeded.n = n; ctx->seeded.reads = d_reads; ctx->seeded.bytes = bytes; ctx->seeded.read_off = d_off;
    ctx->seeded.seed_off = d_tab + 2 * n1; ctx->seeded.seeds = P.seeds; ctx->seeded.n_seeds = total; ctx->seeded.graph = graph;
    return VGK_OK;
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }      // (no exception leaves the C ABI)

double vgk_minimizer_last_ms(vgk_ctx* ctx) { return ctx ? ctx->minimizer_ms : 0.0; }

// ---- reads of any length: every minimizer listed; the seeds of those the caller takes (include/vgk.h) ----------------------------------------
int vgk_minimizer_list(vgk_ctx* ctx, const vgk_minimizer_index* ix, const char* reads, const uint64_t* read_off, uint32_t n,
                       uint64_t* minimizer_off, vgk_read_minimizer* minimizers, size_t cap, size_t* written) try {
    if (!ctx || !ix || !vgk_tables_usable(ix->ctx, ctx) || !minimizer_off || (n && (!reads || !read_off)) || (!minimizers && cap)) return VGK_EINVAL;
    if (written) *written = 0;
    minimizer_off[0] = 0;
    if (!n) return VGK_OK;
    for (uint32_t i = 0; i < n; ++i) if (read_off[i + 1] < read_off[i]) return VGK_EINVAL;
    const uint64_t bytes = read_off[n] - read_off[0];
    if (bytes > 0xfffffff0ull) return VGK_ETOOBIG;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    // a read is cut into stretches of MZ_LIST_WINDOWS windows, a lane each (a 15 kbp read alone would keep one lane busy for 15 000 steps)
    std::vector<MzListItem> items; std::vector<uint64_t> item_first((size_t)n + 1, 0);
    const uint32_t k = ix->dev.k, w = ix->dev.w;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t L = read_off[i + 1] - read_off[i];
        if (L > 0xfffffff0ull) return VGK_ETOOBIG;
        const uint32_t windows = L >= (uint64_t)k + w - 1 ? (uint32_t)(L - k - w + 2) : 0u;
        for (uint32_t s = 0; s < windows; s += MZ_LIST_WINDOWS) items.push_back(MzListItem{i, s});
        item_first[i + 1] = items.size();
    }
    if (items.size() > 0xfffffff0ull) return VGK_ETOOBIG;
    const uint32_t n_items = (uint32_t)items.size();
    const size_t n1 = (size_t)n + 1, m1 = (size_t)n_items + 1;
    char* d_reads = (char*)ctx->ensure_scratch(150, bytes + 32);
    uint64_t* d_off = (uint64_t*)ctx->ensure_scratch(151, sizeof(uint64_t) * n1);
    uint32_t* d_tab = (uint32_t*)ctx->ensure_scratch(152, sizeof(uint32_t) * 2 * m1);
    MzListItem* d_items = (MzListItem*)ctx->ensure_scratch(157, sizeof(MzListItem) * std::max<size_t>(n_items, 1));
    if (!d_reads || !d_off || !d_tab || !d_items) return VGK_ENOMEM;
    std::vector<uint64_t> rel(n1);
    for (size_t i = 0; i < n1; ++i) rel[i] = read_off[i] - read_off[0];
    MzListParams P{};
    P.index = ix->dev; P.reads = d_reads + 8; P.read_off = d_off; P.n = n_items; P.items = d_items; P.counts = d_tab; P.first = d_tab + m1; P.pass = 1;
    int rc = be->upload(d_off, rel.data(), sizeof(uint64_t) * n1);
    if (!rc && n_items) rc = be->upload(d_items, items.data(), sizeof(MzListItem) * n_items);
    if (!rc) rc = be->zero(d_reads, 8);
    if (!rc) rc = be->zero(d_reads + 8 + bytes, 16);
    if (!rc && bytes) rc = be->upload(d_reads + 8, reads + read_off[0], bytes);
    if (!rc) rc = be->run_minimizer_list(P);
    if (!rc) rc = be->scan_u32(d_tab, d_tab + m1, (uint32_t)m1);
    std::vector<uint32_t> first(m1);
    if (!rc) rc = be->download(first.data(), d_tab + m1, sizeof(uint32_t) * m1);      // synchronises (rel[], items may go)
    if (rc) return rc;
    for (size_t i = 0; i < n1; ++i) minimizer_off[i] = first[item_first[i]];
    const size_t total = first[n_items];
    if (written) *written = total;
    if (total > cap) return VGK_EOPS;
    if (!total) return VGK_OK;
    vgk_read_minimizer* d_out = (vgk_read_minimizer*)ctx->ensure_scratch(153, sizeof(vgk_read_minimizer) * total);
    if (!d_out) return VGK_ENOMEM;
    P.out = d_out; P.pass = 2;
    if ((rc = be->run_minimizer_list(P))) return rc;
    return be->download(minimizers, d_out, sizeof(vgk_read_minimizer) * total);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }

int vgk_minimizer_seeds_of(vgk_ctx* ctx, const vgk_minimizer_index* ix, const vgk_read_minimizer* minimizers, const uint8_t* take, size_t n_minimizers,
                           uint64_t* seed_off, vgk_seed* seeds, size_t cap, size_t* written) try {
    if (!ctx || !ix || !vgk_tables_usable(ix->ctx, ctx) || !seed_off || (n_minimizers && (!minimizers || !take)) || (!seeds && cap)) return VGK_EINVAL;
    if (written) *written = 0;
    seed_off[0] = 0;
    if (!n_minimizers) return VGK_OK;
    if (n_minimizers > 0xfffffff0ull) return VGK_ETOOBIG;
    Backend* be = ctx->be.get();
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint32_t n = (uint32_t)n_minimizers; const size_t n1 = (size_t)n + 1;
    vgk_read_minimizer* d_min = (vgk_read_minimizer*)ctx->ensure_scratch(153, sizeof(vgk_read_minimizer) * n_minimizers);
    uint8_t* d_take = (uint8_t*)ctx->ensure_scratch(154, n_minimizers + 16);
    uint32_t* d_tab = (uint32_t*)ctx->ensure_scratch(155, sizeof(uint32_t) * 2 * n1);
    if (!d_min || !d_take || !d_tab) return VGK_ENOMEM;
    MzSeedsOfParams P{};
    P.index = ix->dev; P.mins = d_min; P.take = d_take; P.n = n; P.counts = d_tab; P.first = d_tab + n1; P.pass = 1;
    int rc = be->upload(d_min, minimizers, sizeof(vgk_read_minimizer) * n_minimizers);
    if (!rc) rc = be->upload(d_take, take, n_minimizers);
    if (!rc) rc = be->run_minimizer_seeds_of(P);
    // (a 32-bit prefix sum: the hits of the taken minimizers of one call stay below 2^32 — checked against the index's own size below)
    if (!rc) rc = be->scan_u32(d_tab, d_tab + n1, (uint32_t)n1);
    std::vector<uint32_t> first(n1);
    if (!rc) rc = be->download(first.data(), d_tab + n1, sizeof(uint32_t) * n1);
    if (rc) return rc;
    uint64_t check = 0;
    for (size_t j = 0; j < n_minimizers; ++j) if (take[j]) check += minimizers[j].hits;
    if (check > 0xfffffff0ull) return VGK_ETOOBIG;
    for (size_t j = 0; j < n1; ++j) seed_off[j] = first[j];
    const size_t total = first[n];
    if (written) *written = total;
    if (total > cap) return VGK_EOPS;
    if (!total) return VGK_OK;
    vgk_seed* d_out = (vgk_seed*)ctx->ensure_scratch(156, sizeof(vgk_seed) * total);
    if (!d_out) return VGK_ENOMEM;
    P.out = d_out; P.pass = 2;
    if ((rc = be->run_minimizer_seeds_of(P))) return rc;
    return be->download(seeds, d_out, sizeof(vgk_seed) * total);
} catch (const std::bad_alloc&) { return VGK_ENOMEM; } catch (...) { return VGK_EINVAL; }

}  // extern "C"
