// rescue_requests.cpp — see rescue_requests.hpp.
#include "rescue_requests.hpp"
#include <algorithm>
#include <cmath>
#include <exception>
#include <mutex>
#include <thread>

namespace vgamd {

namespace {
inline char complement(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return c; } }

template <class F> void chunks(size_t n, unsigned threads, F body) {
    if (!threads) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(n / 4096, 1));
    if (threads < 2) { body((size_t)0, n); return; }
    std::vector<std::thread> ts;
    const size_t per = (n + threads - 1) / threads;
    std::mutex first_mutex; std::exception_ptr first;                    // (a worker's exception is rethrown on the caller after the join, never left to std::terminate)
    auto guarded = [&](size_t lo, size_t hi) { try { if (lo < hi) body(lo, hi); } catch (...) { std::lock_guard<std::mutex> hold(first_mutex); if (!first) first = std::current_exception(); } };
    for (unsigned t = 1; t < threads; ++t) ts.emplace_back([&, t]() { const size_t lo = std::min(n, t * per); guarded(lo, std::min(n, lo + per)); });
    guarded((size_t)0, std::min(n, per));
    for (auto& t : ts) t.join();
    if (first) std::rethrow_exception(first);
}
}  // namespace

void build_rescue_requests(uint32_t n_pairs, const vgk_gapless_result* res, const vgk_extension* ext, const uint32_t* nodes, uint32_t n_nodes, const int64_t* col,
                           const char* reads, uint32_t L, double mean, double sd, double stdevs, unsigned host_threads, RescueRequestTable& out) {
    auto full = [&](uint32_t i) { return res[i].status == 0 && res[i].full_length != 0; };
    // which pairs: exactly one mate with a full-length extension set
    std::vector<uint32_t> slot((size_t)n_pairs + 1, 0);
    chunks(n_pairs, host_threads, [&](size_t lo, size_t hi) { for (size_t p = lo; p < hi; ++p) slot[p + 1] = full(2 * (uint32_t)p) != full(2 * (uint32_t)p + 1) ? 1u : 0u; });
    for (uint32_t p = 0; p < n_pairs; ++p) slot[p + 1] += slot[p];
    const size_t m = slot[n_pairs];
    out.mapped.assign(m, 0); out.lost.assign(m, 0); out.requests.assign(6 * m, 0); out.reads.assign(m * (size_t)L, 'N');
    // a forward-mapped mate starting at column s: its partner lies downstream on the other strand, within [s + mean - k sd - L, s + (mean + k sd) 1.1 + 40];
    // a reverse-mapped mate ending at column e: upstream on the forward strand
    const double lo_d = std::max(0.0, mean - stdevs * sd - (double)L), hi_d = (mean + stdevs * sd) * 1.1 + 40.0;
    const int64_t* col_end = col + n_nodes + 1;
    // numpy searchsorted(col, x, side = "right") = the first node whose first column lies beyond x — looked for from `near` outwards: the rescue window
    // lies a fragment's length from the mapped mate, a few dozen nodes, and a binary search over the whole table is seventeen cache misses
    auto upper = [&](double x, int64_t near) {
        int64_t lo = near, hi = near, step = 16;
        while (lo > 0 && !((double)col[lo] <= x)) { lo = std::max<int64_t>(0, lo - step); step *= 2; }                 // col[lo] <= x (or lo = 0)
        step = 16;
        while (hi < (int64_t)n_nodes + 1 && !(x < (double)col[hi])) { hi = std::min<int64_t>((int64_t)n_nodes + 1, hi + step); step *= 2; }   // x < col[hi] (or hi = past the end)
        return (int64_t)(std::upper_bound(col + lo, std::min(col + hi + 1, col_end), x, [](double v, int64_t c) { return v < (double)c; }) - col);
    };
    auto olen = [&](uint32_t oriented) { return col[(oriented >> 1) + 1] - col[oriented >> 1]; };
    chunks(n_pairs, host_threads, [&](size_t plo, size_t phi) {
        for (size_t p = plo; p < phi; ++p) {
            if (slot[p + 1] == slot[p]) continue;
            const size_t k = slot[p];
            const uint32_t mapped = full(2 * (uint32_t)p) ? 2 * (uint32_t)p : 2 * (uint32_t)p + 1, lost = mapped ^ 1u;
            out.mapped[k] = mapped; out.lost[k] = lost;
            const vgk_extension& e0 = ext[res[mapped].ext_begin];
            const uint32_t first = nodes[e0.path_begin];
            const bool fwd = (first & 1u) == 0;
            const double s_col = (double)(col[first >> 1] + (int64_t)e0.offset), e_col = (double)(col[(first >> 1) + 1] - (int64_t)e0.offset);
            const double c_lo = fwd ? s_col + lo_d : e_col - hi_d, c_hi = fwd ? s_col + hi_d : e_col - lo_d;
            const int64_t node_lo = std::min<int64_t>(std::max<int64_t>(upper(std::max(c_lo, 0.0), first >> 1) - 1, 0), (int64_t)n_nodes - 1);
            const int64_t node_hi = std::min<int64_t>(std::max<int64_t>(upper(std::min(c_hi, (double)(col[n_nodes] - 1)), first >> 1), 1), (int64_t)n_nodes);
            int64_t* rq = out.requests.data() + 6 * k;
            rq[0] = node_lo; rq[1] = node_hi; rq[2] = 0; rq[3] = 0; rq[4] = -1; rq[5] = 0;
            // the mate as it reads along the FORWARD strand of that subgraph: reverse-complemented when its partner maps forward
            const bool rc = fwd;
            const char* src = reads + (size_t)lost * L; char* dst = out.reads.data() + k * (size_t)L;
            if (rc) for (uint32_t t = 0; t < L; ++t) dst[t] = complement(src[L - 1 - t]); else std::copy(src, src + L, dst);
            // dozeu's seed: the best extension of the lost mate inside the subgraph on the strand it is rescued on (best score, the earlier among equals)
            const uint32_t ne = res[lost].status == 0 ? res[lost].n_ext : 0u;
            const vgk_extension* best = nullptr;
            for (uint32_t x = 0; x < ne; ++x) {
                const vgk_extension& e = ext[res[lost].ext_begin + x];
                if (!e.path_len) continue;
                const uint32_t* pn = nodes + e.path_begin;
                if (((pn[0] & 1u) == 1u) != rc) continue;
                uint32_t pmin = pn[0] >> 1, pmax = pn[0] >> 1;
                for (uint32_t t = 1; t < e.path_len; ++t) { pmin = std::min(pmin, pn[t] >> 1); pmax = std::max(pmax, pn[t] >> 1); }
                if ((int64_t)pmin < node_lo || (int64_t)pmax >= node_hi) continue;
                if (!best || e.score > best->score) best = &e;
            }
            if (best) {
                const vgk_extension& e = *best; const uint32_t* pn = nodes + e.path_begin;
                int64_t path_bases = 0;
                for (uint32_t t = 0; t < e.path_len; ++t) path_bases += olen(pn[t]);
                const uint32_t last_o = pn[e.path_len - 1], first_o = pn[0];
                const int64_t lastlen = olen(last_o), matched = (int64_t)e.read_end - (int64_t)e.read_begin, off = e.offset;
                // seen from the forward strand (a mate rescued as its reverse complement): the path backwards, the read interval mirrored, the offset
                // counted from the last node's other end
                const int64_t end_in_last = e.path_len > 1 ? matched - (path_bases - off - lastlen) : off + matched;
                rq[2] = rc ? (int64_t)L - e.read_end : e.read_begin; rq[3] = rc ? (int64_t)L - e.read_begin : e.read_end;
                rq[4] = rc ? (int64_t)(last_o >> 1) : (int64_t)(first_o >> 1); rq[5] = rc ? lastlen - end_in_last : off;
            }
        }
    });
}

}  // namespace vgamd
