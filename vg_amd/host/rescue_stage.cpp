// rescue_stage.cpp — see rescue_stage.hpp.
#include "rescue_stage.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <mutex>
#include <thread>
#include "rescue_fixups.hpp"

namespace vgamd {

namespace {
// body(k) for k in [0, n) on up to `threads` host threads.  A body that throws (an engine status inside fix_dozeu_score's re-alignment, bad_alloc
// building a subgraph) stops the handing-out of work; the first exception is kept and rethrown on the caller after the join — an exception that
// left a std::thread's function would be std::terminate (Aligner::xdrop_align_many's `each` has the same rule).
template <class F> void on_threads(size_t n, unsigned threads, F body) {
    if (!threads) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(n, 1));
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::mutex first_mutex; std::exception_ptr first;
    auto work = [&]() {
        try {
            for (size_t i; !failed.load(std::memory_order_relaxed) && (i = next.fetch_add(64)) < n;)
                for (size_t k = i; k < std::min(n, i + 64); ++k) body(k);
        } catch (...) {
            std::lock_guard<std::mutex> hold(first_mutex);
            if (!first) first = std::current_exception();
            failed.store(true, std::memory_order_relaxed);
        }
    };
    std::vector<std::thread> ts;
    for (unsigned t = 1; t < threads; ++t) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
    if (first) std::rethrow_exception(first);
}
}  // namespace

void run_rescue_stage(const Aligner& aligner, const RescueGraph& G, const std::vector<RescueRequest>& requests, uint64_t max_cells,
                      unsigned host_threads, std::vector<RescueResult>& results, std::vector<vgk_op>* out_ops, std::vector<uint64_t>* out_ops_begin) {
    const size_t n = requests.size();
    results.assign(n, RescueResult{});
    std::vector<std::vector<vgk_op>> op_runs(out_ops && out_ops_begin ? n : 0);
    auto lap_t0 = std::chrono::steady_clock::now(); const bool lap_on = std::getenv("VGAMD_TIMING") != nullptr;
    auto lap = [&](const char* what) { if (!lap_on) return; const auto t = std::chrono::steady_clock::now(); std::fprintf(stderr, "[rescue_stage] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - lap_t0).count()); lap_t0 = t; };
    std::deque<HashGraph> graphs(n);
    std::deque<Alignment> alns(n);
    std::vector<Aligner::XdropRequest> all(n);
    std::vector<uint8_t> runs(n, 0);
    // 1. the subgraphs (nodes [lo, hi) and the edges among them), their order, the guard, dozeu's seed
    on_threads(n, host_threads, [&](size_t k) {
        const RescueRequest& rq = requests[k]; RescueResult& out = results[k];
        if (rq.node_lo >= rq.node_hi || rq.node_hi > G.n_nodes || !rq.read_len) { out.status = 2; return; }
        uint64_t bases = 0;
        for (uint32_t v = rq.node_lo; v < rq.node_hi; ++v) bases += G.node_len[v];
        if (bases * rq.read_len > max_cells) { out.status = 1; return; }                   // (:3372-3381: refused, the pair keeps what it has)
        HashGraph& g = graphs[k];
        Aligner::XdropRequest& x = all[k];
        x.order.reserve(rq.node_hi - rq.node_lo);
        for (uint32_t v = rq.node_lo; v < rq.node_hi; ++v) x.order.push_back(g.create_handle(std::string(G.seq + G.seq_off[v], G.node_len[v]), (nid_t)v + 1));
        for (uint32_t v = rq.node_lo; v < rq.node_hi; ++v)
            for (uint32_t e = G.succ_off[v]; e < G.succ_off[v + 1]; ++e)
                if (G.succ[e] >= rq.node_lo && G.succ[e] < rq.node_hi) g.create_edge(x.order[v - rq.node_lo], x.order[G.succ[e] - rq.node_lo]);
        Alignment& aln = alns[k];
        aln.sequence.assign(rq.read, rq.read_len);
        x.alignment = &aln; x.graph = &g; x.reverse_complemented = false;
        x.max_gap_length = (uint16_t)std::min<size_t>(aligner.scorer->longest_detectable_gap(rq.read_len, rq.read_len / 2), 65535);      // (:3383)
        if (rq.seed_node >= (int64_t)rq.node_lo && rq.seed_node < (int64_t)rq.node_hi && rq.seed_end > rq.seed_begin) {
            MaximalExactMatch m; m.begin = (size_t)rq.seed_begin; m.end = (size_t)rq.seed_end;
            m.nodes.push_back({(nid_t)rq.seed_node + 1, (size_t)rq.seed_offset, false});
            x.mems.push_back(m);
        }
        runs[k] = 1;
    });
    lap("subgraphs");
    // 2. every request's X-drop passes side by side
    std::vector<Aligner::XdropRequest> todo; std::vector<size_t> owner;
    for (size_t k = 0; k < n; ++k) if (runs[k]) { todo.push_back(std::move(all[k])); owner.push_back(k); }
    if (!todo.empty()) aligner.align_xdrop_many(todo);
    lap("align_xdrop_many");
    // 3. the fix-ups, the answers
    on_threads(todo.size(), host_threads, [&](size_t a) {
        const size_t k = owner[a]; Alignment& aln = alns[k]; RescueResult& out = results[k];
        fix_dozeu_score(aln, aligner, graphs[k], todo[a].order);
        fix_dozeu_end_deletions(aln);
        out.score = aln.score; out.n_mappings = (uint32_t)aln.path.mapping.size();
        if (!aln.path.mapping.empty()) { out.first_node = aln.path.mapping.front().position.node_id - 1; out.first_offset = aln.path.mapping.front().position.offset; }
        uint32_t to = 0;
        for (const Mapping& m : aln.path.mapping) for (const Edit& e : m.edit) if (e.from_length) to += (uint32_t)e.to_length;
        out.aligned_read_bases = to;
        if (!op_runs.empty()) for (const Mapping& m : aln.path.mapping) {
            int prev = -1;
            for (const Edit& e : m.edit) {
                const int op = (edit_is_match(e) || edit_is_sub(e)) ? VGK_OP_M : edit_is_deletion(e) ? VGK_OP_D : VGK_OP_I;
                if (op == VGK_OP_M && prev == VGK_OP_M) op_runs[k].back().len = (uint16_t)(op_runs[k].back().len + e.to_length);
                else { vgk_op o{}; o.node = (uint32_t)(m.position.node_id - 1); o.op = (uint8_t)op; o.len = (uint16_t)(op == VGK_OP_D ? e.from_length : e.to_length); op_runs[k].push_back(o); }
                prev = op;
            }
        }
        graphs[k] = HashGraph(); aln = Alignment(); todo[a] = Aligner::XdropRequest();      // released where they were built: on the threads
    });
    lap("fix-ups");
    if (!op_runs.empty()) {
        out_ops_begin->assign(n + 1, 0);
        for (size_t k = 0; k < n; ++k) (*out_ops_begin)[k + 1] = (*out_ops_begin)[k] + op_runs[k].size();
        out_ops->clear(); out_ops->reserve((size_t)(*out_ops_begin)[n]);
        for (size_t k = 0; k < n; ++k) out_ops->insert(out_ops->end(), op_runs[k].begin(), op_runs[k].end());
    }
}

}  // namespace vgamd
