// STAND-IN (vg_amd/host/vg_standin/): restates a slice of vg / libhandlegraph / libvgio that stays vg's own in a real
// integration; present only so the reference's unit tests can be driven without vg.  Excluded from size / originality claims.
#include "handle_graph.hpp"
#include <algorithm>
#include <stdexcept>

namespace vgamd {

handle_t HashGraph::create_handle(const std::string& seq) { return create_handle(seq, next_id_); }

handle_t HashGraph::create_handle(const std::string& seq, nid_t id) {
    if (index_.count(id)) throw std::runtime_error("HashGraph: duplicate node id");
    index_[id] = ids_.size();
    ids_.push_back(id); seqs_.push_back(seq); out_.emplace_back(); in_.emplace_back();
    if (ids_.size() == 1) { min_id_ = max_id_ = id; }
    min_id_ = std::min(min_id_, id); max_id_ = std::max(max_id_, id);
    next_id_ = std::max(next_id_, id + 1);
    return get_handle(id, false);
}

void HashGraph::create_edge(const handle_t& from, const handle_t& to) {
    if (get_is_reverse(from) || get_is_reverse(to)) throw std::runtime_error("HashGraph: forward-strand edges only");
    nid_t a = get_id(from), b = get_id(to);
    auto& o = out_[index_.at(a)];
    if (std::find(o.begin(), o.end(), b) != o.end()) return;
    o.push_back(b); in_[index_.at(b)].push_back(a);
}

static char comp(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return c; }
}

std::string HashGraph::get_sequence(const handle_t& h) const {
    const std::string& s = seqs_[index_.at(get_id(h))];
    if (!get_is_reverse(h)) return s;
    std::string r(s.rbegin(), s.rend());
    for (auto& c : r) c = comp(c);
    return r;
}

bool HashGraph::follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    size_t i = index_.at(get_id(h));
    bool rev = get_is_reverse(h);
    // on the reverse strand, "right" neighbours are the forward predecessors, flipped
    const auto& adj = (go_left != rev) ? in_[i] : out_[i];
    for (nid_t n : adj) if (!it(get_handle(n, rev))) return false;
    return true;
}

bool HashGraph::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    for (nid_t id : ids_) if (!it(get_handle(id, false))) return false;
    return true;
}

std::string ReverseGraph::get_sequence(const handle_t& h) const {
    std::string s = g_->get_sequence(h);
    std::reverse(s.begin(), s.end());
    if (complement_) for (auto& c : s) c = comp(c);
    return s;
}

NullMaskingGraph::NullMaskingGraph(const HandleGraph* g) : g_(g) {
    g->for_each_handle_v([&](const handle_t& h) { if (g->get_length(h) == 0) ++nulls_; });
}

bool NullMaskingGraph::follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    return g_->follow_edges(h, go_left, [&](const handle_t& n) { return g_->get_length(n) > 0 ? it(n) : true; });
}

bool NullMaskingGraph::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    return g_->for_each_handle([&](const handle_t& n) { return g_->get_length(n) > 0 ? it(n) : true; });
}

DozeuPinningOverlay::DozeuPinningOverlay(const HandleGraph* g, bool sinks) : graph(g), preserve_sinks(sinks) {
    std::vector<handle_t> empty_nodes;
    uint64_t min_handle = ~0ull;
    graph->for_each_handle_v([&](const handle_t& h) {
        for (handle_t x : {h, graph->flip(h)}) { min_handle = std::min<uint64_t>(min_handle, (uint64_t)x.v); max_handle = std::max<uint64_t>(max_handle, (uint64_t)x.v); }
        if (graph->get_length(h) == 0) empty_nodes.push_back(h);
    });
    num_null_nodes = empty_nodes.size();
    handle_val_range = graph->get_node_count() ? max_handle - min_handle + 1 : 0;
    for (const handle_t& empty : empty_nodes) {
        // an empty node with nothing on its pinning side is a pinning tip; its neighbours inherit that role
        bool should_preserve = graph->follow_edges(empty, !preserve_sinks, [&](const handle_t&) { return false; });
        if (!should_preserve) continue;
        graph->follow_edges_v(empty, preserve_sinks, [&](const handle_t& next) {
            bool must_duplicate = !graph->follow_edges(next, !preserve_sinks, [&](const handle_t& prev) { return prev == empty; });
            if (must_duplicate) duplicated_handles.insert(next);
        });
    }
}

bool DozeuPinningOverlay::has_node(nid_t id) const {
    if (is_a_duplicate_id(id)) {
        nid_t under = get_underlying_id(id);
        return graph->has_node(under) && duplicated_handles.count(graph->get_handle(under));
    }
    return graph->has_node(id) && graph->get_length(graph->get_handle(id)) != 0;
}

handle_t DozeuPinningOverlay::get_handle(nid_t id, bool is_reverse) const {
    if (is_a_duplicate_id(id)) return get_duplicate_handle(graph->get_handle(get_underlying_id(id), is_reverse));
    return graph->get_handle(id, is_reverse);
}

nid_t DozeuPinningOverlay::get_id(const handle_t& h) const {
    if (is_a_duplicate_handle(h)) return graph->get_id(get_underlying_handle(h)) + (graph->max_node_id() - graph->min_node_id() + 1);
    return graph->get_id(h);
}

bool DozeuPinningOverlay::get_is_reverse(const handle_t& h) const { return graph->get_is_reverse(get_underlying_handle(h)); }

handle_t DozeuPinningOverlay::flip(const handle_t& h) const {
    if (is_a_duplicate_handle(h)) return get_duplicate_handle(graph->flip(get_underlying_handle(h)));
    return graph->flip(h);
}

handle_t DozeuPinningOverlay::get_underlying_handle(const handle_t& h) const {
    return is_a_duplicate_handle(h) ? handle_t{(int64_t)((uint64_t)h.v - handle_val_range)} : h;
}

bool DozeuPinningOverlay::follow_edges(const handle_t& handle, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    handle_t to_iterate = handle;
    if (is_a_duplicate_handle(handle)) {
        if (preserve_sinks != (go_left != get_is_reverse(handle))) return true;     // the duplicate is a tip on its pinning side
        to_iterate = get_underlying_handle(handle);
    }
    return graph->follow_edges(to_iterate, go_left, [&](const handle_t& next) {
        bool keep_going = true;
        if (graph->get_length(next) > 0) {
            keep_going = it(next);
            handle_t fwd = graph->get_is_reverse(next) ? graph->flip(next) : next;
            if (keep_going && duplicated_handles.count(fwd)) {
                if (preserve_sinks != (go_left != graph->get_is_reverse(next))) keep_going = it(get_duplicate_handle(next));
            }
        }
        return keep_going;
    });
}

bool DozeuPinningOverlay::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    bool keep_going = graph->for_each_handle([&](const handle_t& h) { return graph->get_length(h) > 0 ? it(h) : true; });
    for (auto i = duplicated_handles.begin(); i != duplicated_handles.end() && keep_going; ++i) keep_going = it(get_duplicate_handle(*i));
    return keep_going;
}

nid_t DozeuPinningOverlay::max_node_id() const {
    nid_t m = graph->max_node_id();
    for (const handle_t& h : duplicated_handles) m = std::max(m, get_id(get_duplicate_handle(h)));
    return m;
}

namespace handlealgs {

std::vector<handle_t> lazier_topological_order(const HandleGraph* g) {
    std::unordered_map<handle_t, size_t, handle_hash> indeg;
    std::set<handle_t> ready;
    g->for_each_handle_v([&](const handle_t& h) {
        size_t d = g->get_degree(h, true);
        indeg[h] = d;
        if (d == 0) ready.insert(h);
    });
    std::vector<handle_t> order;
    order.reserve(indeg.size());
    while (!ready.empty()) {
        handle_t n = *ready.begin();
        ready.erase(ready.begin());
        order.push_back(n);
        g->follow_edges_v(n, false, [&](const handle_t& nx) {
            auto it = indeg.find(nx);
            if (it != indeg.end() && it->second > 0 && --it->second == 0) ready.insert(nx);
        });
    }
    return order;
}

std::vector<handle_t> head_nodes(const HandleGraph* g) {
    std::vector<handle_t> out;
    g->for_each_handle_v([&](const handle_t& h) { if (g->get_degree(h, true) == 0) out.push_back(h); });
    return out;
}

std::vector<handle_t> tail_nodes(const HandleGraph* g) {
    std::vector<handle_t> out;
    g->for_each_handle_v([&](const handle_t& h) { if (g->get_degree(h, false) == 0) out.push_back(h); });
    return out;
}

}  // namespace handlealgs
}  // namespace vgamd
