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
