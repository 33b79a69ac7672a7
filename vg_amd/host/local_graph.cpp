// local_graph.cpp — see local_graph.hpp.
#include "local_graph.hpp"
#include <algorithm>
#include <deque>
#include <queue>
#include <set>
#include <stdexcept>
#include <unordered_set>

namespace vgamd {

namespace {
char complement_base(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;
    }
}
template <class V> bool holds(const V& v, const handle_t& x) { return std::find(v.begin(), v.end(), x) != v.end(); }
template <class V> void drop(V& v, const handle_t& x) { auto it = std::find(v.begin(), v.end(), x); if (it != v.end()) v.erase(it); }
}  // namespace

// ---- LocalGraph ---------------------------------------------------------------------------------------------------------------------
handle_t LocalGraph::create_handle(const std::string& seq) { return create_handle(seq, max_node_id() + 1); }

handle_t LocalGraph::create_handle(const std::string& seq, nid_t id) {
    if (id <= 0) throw std::runtime_error("LocalGraph: node ids are positive");
    if (!nodes_.emplace(id, Node{seq, {}, {}}).second) throw std::runtime_error("LocalGraph: duplicate node id " + std::to_string(id));
    return get_handle(id, false);
}

bool LocalGraph::has_edge(const handle_t& from, const handle_t& to) const {
    const Node& n = nodes_.at(get_id(from));
    return get_is_reverse(from) ? holds(n.left, flip(to)) : holds(n.right, to);
}

void LocalGraph::create_edge(const handle_t& from, const handle_t& to) {
    if (!has_node(get_id(from)) || !has_node(get_id(to))) throw std::runtime_error("LocalGraph: edge to a node that is not there");
    if (has_edge(from, to)) return;
    // the edge is listed at either end, seen from that end's forward strand; an edge from a strand to its own reverse has one end only
    Node& a = nodes_.at(get_id(from));
    if (get_is_reverse(from)) a.left.push_back(flip(to)); else a.right.push_back(to);
    if (from == flip(to)) return;
    Node& b = nodes_.at(get_id(to));
    if (get_is_reverse(to)) b.right.push_back(flip(from)); else b.left.push_back(from);
}

void LocalGraph::destroy_edge(const handle_t& from, const handle_t& to) {
    Node& a = nodes_.at(get_id(from));
    if (get_is_reverse(from)) drop(a.left, flip(to)); else drop(a.right, to);
    if (from == flip(to)) return;
    Node& b = nodes_.at(get_id(to));
    if (get_is_reverse(to)) drop(b.right, flip(from)); else drop(b.left, from);
}

void LocalGraph::edges_of(nid_t id, std::vector<edge_t>& out) const {
    const Node& n = nodes_.at(id);
    const handle_t fwd = get_handle(id, false);
    for (const handle_t& x : n.right) out.emplace_back(fwd, x);
    for (const handle_t& x : n.left) if (x != fwd) out.emplace_back(x, fwd);       // (a loop fwd -> fwd was listed through `right`)
}

void LocalGraph::destroy_handle(const handle_t& h) {
    const nid_t id = get_id(h);
    std::vector<edge_t> touching;
    edges_of(id, touching);
    for (const edge_t& e : touching) destroy_edge(e.first, e.second);
    nodes_.erase(id);
}

std::pair<handle_t, handle_t> LocalGraph::divide_handle(const handle_t& h, size_t offset) {
    const nid_t id = get_id(h);
    const size_t len = nodes_.at(id).seq.size();
    if (offset > len) throw std::runtime_error("LocalGraph: cut behind the end of a node");
    const size_t at = get_is_reverse(h) ? len - offset : offset;               // on the forward strand
    std::vector<edge_t> touching;
    edges_of(id, touching);
    for (const edge_t& e : touching) destroy_edge(e.first, e.second);
    const std::string tail = nodes_.at(id).seq.substr(at);
    nodes_.at(id).seq.resize(at);
    const handle_t head = get_handle(id, false), rest = create_handle(tail);
    for (const edge_t& e : touching) {
        // what left through the node's right side now leaves the new piece's; what arrived at its left side still does
        const handle_t from = e.first == head ? rest : e.first;
        const handle_t to = e.second == flip(head) ? flip(rest) : e.second;
        create_edge(from, to);
    }
    create_edge(head, rest);
    return get_is_reverse(h) ? std::make_pair(flip(rest), flip(head)) : std::make_pair(head, rest);
}

handle_t LocalGraph::truncate_handle(const handle_t& h, bool trunc_left, size_t offset) {
    const auto pieces = divide_handle(h, offset);
    destroy_handle(trunc_left ? pieces.first : pieces.second);
    return trunc_left ? pieces.second : pieces.first;
}

size_t LocalGraph::get_total_length() const {
    size_t total = 0;
    for (const auto& kv : nodes_) total += kv.second.seq.size();
    return total;
}

std::string LocalGraph::get_sequence(const handle_t& h) const {
    const std::string& s = nodes_.at(get_id(h)).seq;
    if (!get_is_reverse(h)) return s;
    std::string r(s.rbegin(), s.rend());
    for (char& c : r) c = complement_base(c);
    return r;
}

bool LocalGraph::follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    const Node& n = nodes_.at(get_id(h));
    // on the reverse strand it is the forward strand's other side, every neighbour turned around
    const bool rev = get_is_reverse(h);
    const std::vector<handle_t> copy = (go_left != rev) ? n.left : n.right;    // (a copy: the callback may edit the graph)
    for (const handle_t& x : copy) if (!it(rev ? flip(x) : x)) return false;
    return true;
}

bool LocalGraph::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    std::vector<nid_t> ids;
    ids.reserve(nodes_.size());
    for (const auto& kv : nodes_) ids.push_back(kv.first);
    for (nid_t id : ids) if (nodes_.count(id) && !it(get_handle(id, false))) return false;
    return true;
}

// ---- StrandSplitView ----------------------------------------------------------------------------------------------------------------
bool StrandSplitView::follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const {
    // Walk the underlying graph from the strand this handle reads.  A forward handle of ours lands on the forward handle of the
    // strand it reaches; a reverse handle of ours reads strand s backwards, i.e. walks the underlying strand !s, and lands on the
    // reverse handle of the node that stands for the strand OPPOSITE to the one reached.
    const bool backwards = get_is_reverse(h);
    return g_->follow_edges(get_underlying_handle(h), go_left, [&](const handle_t& reached) {
        const bool strand = g_->get_is_reverse(reached) != backwards;
        return it(get_handle((g_->get_id(reached) << 1) | (strand ? 1 : 0), backwards));
    });
}

bool StrandSplitView::for_each_handle(const std::function<bool(const handle_t&)>& it) const {
    return g_->for_each_handle([&](const handle_t& u) {
        const nid_t id = g_->get_id(u);
        return it(get_handle(id << 1, false)) && it(get_handle((id << 1) | 1, false));
    });
}

// ---- algorithms ---------------------------------------------------------------------------------------------------------------------
namespace handlealgs {

std::vector<handle_t> find_tips(const HandleGraph* g) {
    std::vector<handle_t> tips;
    g->for_each_handle_v([&](const handle_t& h) {
        if (g->get_degree(h, true) == 0) tips.push_back(h);
        if (g->get_degree(h, false) == 0) tips.push_back(g->flip(h));
    });
    return tips;
}

std::unordered_map<handle_t, size_t, handle_hash> find_shortest_paths(const HandleGraph* g, const handle_t& start, bool traverse_leftward) {
    std::unordered_map<handle_t, size_t, handle_hash> dist;
    using Item = std::pair<size_t, int64_t>;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> todo;
    todo.emplace(0, start.v);
    while (!todo.empty()) {
        const Item top = todo.top(); todo.pop();
        const handle_t here{top.second};
        if (!dist.emplace(here, top.first).second) continue;
        const size_t beyond = top.first + (here == start ? 0 : g->get_length(here));
        g->follow_edges_v(here, traverse_leftward, [&](const handle_t& next) { if (!dist.count(next)) todo.emplace(beyond, next.v); });
    }
    return dist;
}

bool is_single_stranded(const HandleGraph* g) {
    bool single = true;
    g->for_each_handle_v([&](const handle_t& h) {
        for (int left = 0; left < 2; ++left)
            g->follow_edges_v(h, left == 1, [&](const handle_t& next) { if (g->get_is_reverse(next)) single = false; });
    });
    return single;
}

bool is_acyclic(const HandleGraph* g) {
    // depth-first over oriented nodes; a walk that meets an oriented node still on its own stack closes a cycle
    std::unordered_map<int64_t, uint8_t> colour;                               // 1 = on the stack, 2 = done
    bool acyclic = true;
    g->for_each_handle_v([&](const handle_t& fwd) {
        for (int rev = 0; rev < 2 && acyclic; ++rev) {
            const handle_t root = rev ? g->flip(fwd) : fwd;
            if (colour.count(root.v)) continue;
            struct Frame { handle_t h; std::vector<handle_t> next; size_t at; };
            std::vector<Frame> stack;
            auto open = [&](const handle_t& h) {
                colour[h.v] = 1; stack.push_back({h, {}, 0});
                g->follow_edges_v(h, false, [&](const handle_t& n) { stack.back().next.push_back(n); });
            };
            open(root);
            while (!stack.empty() && acyclic) {
                Frame& f = stack.back();
                if (f.at == f.next.size()) { colour[f.h.v] = 2; stack.pop_back(); continue; }
                const handle_t n = f.next[f.at++];
                auto found = colour.find(n.v);
                if (found == colour.end()) open(n);
                else if (found->second == 1) acyclic = false;
            }
        }
    });
    return acyclic;
}

std::unordered_map<nid_t, std::pair<nid_t, bool>> split_strands(const HandleGraph* g, LocalGraph* into) {
    if (into->get_node_count()) throw std::invalid_argument("split_strands: the output graph must be empty");
    std::unordered_map<nid_t, std::pair<nid_t, bool>> to_source;
    std::unordered_map<int64_t, handle_t> copy;                                // oriented node of g -> forward node of `into`
    g->for_each_handle_v([&](const handle_t& h) {
        for (int rev = 0; rev < 2; ++rev) {
            const handle_t o = rev ? g->flip(h) : h;
            const handle_t c = into->create_handle(g->get_sequence(o));
            copy[o.v] = c; to_source[into->get_id(c)] = {g->get_id(h), rev == 1};
        }
    });
    g->for_each_handle_v([&](const handle_t& h) {
        for (int rev = 0; rev < 2; ++rev) {
            const handle_t o = rev ? g->flip(h) : h;
            g->follow_edges_v(o, false, [&](const handle_t& next) { into->create_edge(copy.at(o.v), copy.at(next.v)); });
        }
    });
    return to_source;
}

namespace {
Dagified dagify_impl(const HandleGraph* g, const std::vector<handle_t>& starts, bool whole_graph, LocalGraph* into, size_t min_preserved_path_length);
}
Dagified dagify_from(const HandleGraph* g, const std::vector<handle_t>& starts, LocalGraph* into, size_t min_preserved_path_length) {
    return dagify_impl(g, starts, false, into, min_preserved_path_length);
}
std::unordered_map<nid_t, nid_t> dagify(const HandleGraph* g, LocalGraph* into, size_t min_preserved_path_length) {
    return dagify_impl(g, {}, true, into, min_preserved_path_length).to_source;
}
namespace {
Dagified dagify_impl(const HandleGraph* g, const std::vector<handle_t>& starts, bool whole_graph, LocalGraph* into, size_t min_preserved_path_length) {
    if (into->get_node_count()) throw std::invalid_argument("dagify: the output graph must be empty");
    // 1. What the walks reach: a breadth-first search over (node, direction); direction 0 follows the edges, 1 runs against them.
    //    Nodes are numbered in the order they are first met; edges are kept as (index, index) along the forward strands.
    //    A node is walked on ONE strand (the graph is single-stranded, though a node's strand may be its reverse).  The strands of a
    //    whole connected component are fixed when a walk first touches it — the handle it is touched through runs WITH the edges —
    //    by a search that ignores edge direction; a node that would need both strands is an error.  A later start on the other
    //    orientation of a node of that component then names walks that run AGAINST the edges.
    std::unordered_map<nid_t, uint8_t> strand_of;
    auto orient_component = [&](const handle_t& first) {
        std::deque<handle_t> q{first};
        strand_of[g->get_id(first)] = g->get_is_reverse(first) ? 1 : 0;
        while (!q.empty()) {
            const handle_t h = q.front(); q.pop_front();
            for (int left = 0; left < 2; ++left)
                g->follow_edges_v(h, left == 1, [&](const handle_t& next) {
                    const uint8_t along = g->get_is_reverse(next) ? 1 : 0;
                    auto found = strand_of.find(g->get_id(next));
                    if (found == strand_of.end()) { strand_of.emplace(g->get_id(next), along); q.push_back(next); }
                    else if (found->second != along) throw std::runtime_error("dagify: node " + std::to_string(g->get_id(next)) + " is walked on both strands; split the strands first");
                });
        }
    };
    std::unordered_map<nid_t, uint32_t> index_of;
    std::vector<nid_t> node_id;
    std::vector<uint8_t> met;                                                  // bit d: met in direction d (0 with the edges, 1 against them)
    std::vector<uint8_t> strand;
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    std::set<std::pair<uint32_t, uint32_t>> have_edge;
    std::deque<std::pair<uint32_t, int>> todo;
    auto meet = [&](nid_t id, int dir) {
        auto found = index_of.find(id);
        uint32_t k;
        if (found == index_of.end()) { k = (uint32_t)node_id.size(); index_of.emplace(id, k); node_id.push_back(id); met.push_back(0); strand.push_back(strand_of.at(id)); }
        else k = found->second;
        if (!(met[k] & (1 << dir))) { met[k] |= (uint8_t)(1 << dir); todo.emplace_back(k, dir); }
        return k;
    };
    auto walk_out = [&]() {
        while (!todo.empty()) {
            const uint32_t k = todo.front().first; const int dir = todo.front().second;
            todo.pop_front();
            g->follow_edges_v(g->get_handle(node_id[k], strand[k] != 0), dir == 1, [&](const handle_t& next) {
                const uint32_t j = meet(g->get_id(next), dir);
                const auto e = dir == 0 ? std::make_pair(k, j) : std::make_pair(j, k);
                if (have_edge.insert(e).second) edges.push_back(e);
            });
        }
    };
    if (whole_graph) {          // dagify(): every node — in the graph's order, what a node's walks reach numbered before the next unmet node
        g->for_each_handle_v([&](const handle_t& h) {
            if (!strand_of.count(g->get_id(h))) orient_component(h);
            meet(g->get_id(h), 0);
            walk_out();
        });
    } else {
        for (const handle_t& s : starts) {
            if (!strand_of.count(g->get_id(s))) orient_component(s);
            meet(g->get_id(s), (g->get_is_reverse(s) ? 1 : 0) != strand_of.at(g->get_id(s)) ? 1 : 0);
        }
        walk_out();
    }
    const uint32_t n = (uint32_t)node_id.size();
    std::vector<std::vector<uint32_t>> out(n), in(n);
    for (const auto& e : edges) { out[e.first].push_back(e.second); in[e.second].push_back(e.first); }
    std::vector<size_t> len(n);
    for (uint32_t k = 0; k < n; ++k) len[k] = g->get_length(g->get_handle(node_id[k], false));

    // 2. Strongly connected components (Tarjan, explicit stack).
    std::vector<uint32_t> comp(n, UINT32_MAX), low(n, 0), num(n, UINT32_MAX);
    std::vector<std::vector<uint32_t>> members;
    {
        std::vector<uint32_t> stack; std::vector<uint8_t> on_stack(n, 0);
        uint32_t counter = 0;
        struct Frame { uint32_t v; size_t next; };
        for (uint32_t root = 0; root < n; ++root) {
            if (num[root] != UINT32_MAX) continue;
            std::vector<Frame> call{{root, 0}};
            num[root] = low[root] = counter++; stack.push_back(root); on_stack[root] = 1;
            while (!call.empty()) {
                Frame& f = call.back();
                if (f.next < out[f.v].size()) {
                    const uint32_t w = out[f.v][f.next++];
                    if (num[w] == UINT32_MAX) { num[w] = low[w] = counter++; stack.push_back(w); on_stack[w] = 1; call.push_back({w, 0}); }
                    else if (on_stack[w]) low[f.v] = std::min(low[f.v], num[w]);
                } else {
                    const uint32_t v = f.v;
                    if (low[v] == num[v]) {
                        members.emplace_back();
                        for (;;) { const uint32_t w = stack.back(); stack.pop_back(); on_stack[w] = 0; comp[w] = (uint32_t)members.size() - 1; members.back().push_back(w); if (w == v) break; }
                        std::sort(members.back().begin(), members.back().end());
                    }
                    call.pop_back();
                    if (!call.empty()) low[call.back().v] = std::min(low[call.back().v], low[v]);
                }
            }
        }
    }
    // 3. A component with a cycle is laid out in layers: inside a layer only the edges that run from an earlier-met node to a later-met
    //    one; every other edge of the component (a loop on one node included) climbs to the next layer.  Walks enter at layer 0 and
    //    may leave from any layer.  Layers are added until no walk of fewer than min_preserved_path_length bases can reach the next one
    //    (counting from anywhere in layer 0 — walks may enter the component anywhere).
    std::vector<uint32_t> layers(members.size(), 1);
    uint64_t copies = n;
    for (size_t c = 0; c < members.size(); ++c) {
        const std::vector<uint32_t>& m = members[c];
        bool cyclic = m.size() > 1;
        for (uint32_t w : out[m[0]]) cyclic = cyclic || w == m[0];
        if (!cyclic) continue;
        // below / here: bases walked from the END of the node a walk started on to the END of this copy (0 in layer 0: a walk may
        // start on any of its nodes, and the node it starts on does not count — the contract of find_shortest_paths too).  The
        // reference's known answers fix this reading: its 2-node loop of 2 + 2 bases takes 2 layers for 1 preserved base and 3 for 5
        // (src/unittest/dagify.cpp:22-139, :140-268).
        std::unordered_map<uint32_t, size_t> below, here;
        for (uint32_t v : m) below[v] = 0;
        const size_t unreachable = SIZE_MAX / 4;
        for (;;) {
            size_t nearest = unreachable;
            for (uint32_t v : m) {                                             // ascending: in-layer predecessors are final
                size_t best = unreachable;                                     // ... to the START of this copy
                for (uint32_t u : in[v]) {
                    if (comp[u] != c) continue;
                    const size_t from = u < v ? (here.count(u) ? here[u] : unreachable) : below[u];
                    if (from < unreachable) best = std::min(best, from);
                }
                here[v] = best < unreachable ? best + len[v] : unreachable; nearest = std::min(nearest, best);
            }
            if (nearest >= min_preserved_path_length) break;
            ++layers[c]; copies += m.size();
            if (copies > 4000000u || layers[c] > min_preserved_path_length * m.size() + 1)
                throw std::runtime_error("dagify: unrolling a cycle to " + std::to_string(min_preserved_path_length) + " bases takes too many copies");
            below.swap(here); here.clear();
        }
    }
    // 4. The copies: layer 0 of every node in meeting order (so the starts come first), then the upper layers component by component.
    std::vector<std::vector<handle_t>> copy(n);
    Dagified result;
    auto add = [&](uint32_t k) {                                               // (a copy keeps the node's forward strand; copy[] holds the handle walks pass it in)
        const handle_t h = into->create_handle(g->get_sequence(g->get_handle(node_id[k], false)));
        result.to_source[into->get_id(h)] = node_id[k];
        copy[k].push_back(strand[k] ? into->flip(h) : h);
    };
    for (uint32_t k = 0; k < n; ++k) add(k);
    for (size_t c = 0; c < members.size(); ++c) for (uint32_t layer = 1; layer < layers[c]; ++layer) for (uint32_t v : members[c]) add(v);
    for (const auto& e : edges) {
        const uint32_t u = e.first, v = e.second;
        if (comp[u] != comp[v]) { for (const handle_t& from : copy[u]) into->create_edge(from, copy[v][0]); continue; }
        const uint32_t depth = layers[comp[u]];
        if (u < v) for (uint32_t layer = 0; layer < depth; ++layer) into->create_edge(copy[u][layer], copy[v][layer]);
        else for (uint32_t layer = 0; layer + 1 < depth; ++layer) into->create_edge(copy[u][layer], copy[v][layer + 1]);
    }
    for (const handle_t& s : starts) {
        const uint32_t k = index_of.at(g->get_id(s));
        const handle_t first = copy[k][0];                                     // in the strand's orientation
        result.starts.push_back((g->get_is_reverse(s) ? 1 : 0) != strand[k] ? into->flip(first) : first);
    }
    return result;
}
}  // namespace

}  // namespace handlealgs

// ---- extraction ---------------------------------------------------------------------------------------------------------------------
namespace {

// nodes in the order a search meets them + every edge it crosses (duplicates welcome: LocalGraph keeps one)
struct Harvest {
    std::vector<nid_t> nodes; std::unordered_set<nid_t> have;
    std::vector<edge_t> edges;
    void node(nid_t id) { if (have.insert(id).second) nodes.push_back(id); }
};

// shortest-first traversals, each oriented node handed out once (the reference's UpdateablePriorityQueue keyed by handle)
class Frontier {
public:
    void push(const handle_t& h, int64_t dist) { if (!done_.count(h.v)) todo_.emplace(dist, h.v); }
    bool pop(handle_t& h, int64_t& dist) {
        while (!todo_.empty()) {
            const auto top = todo_.top(); todo_.pop();
            if (!done_.insert(top.second).second) continue;
            dist = top.first; h = handle_t{top.second};
            return true;
        }
        return false;
    }
private:
    using Item = std::pair<int64_t, int64_t>;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> todo_;
    std::unordered_set<int64_t> done_;
};

handle_t forward_of(const HandleGraph* g, const handle_t& h) { return g->get_is_reverse(h) ? g->flip(h) : h; }
handle_t same_in(const LocalGraph* into, const HandleGraph* source, const handle_t& h) { return into->get_handle(source->get_id(h), source->get_is_reverse(h)); }

}  // namespace

ConnectingGraph extract_connecting_graph(const HandleGraph* source, LocalGraph* into, int64_t max_len, const Position& pos_1, const Position& pos_2,
                                         bool strict_max_len) {
    if (into->get_node_count()) throw std::invalid_argument("extract_connecting_graph: the output graph must be empty");
    ConnectingGraph result;
    const handle_t h1 = source->get_handle(pos_1.node_id, pos_1.is_reverse), h2 = source->get_handle(pos_2.node_id, pos_2.is_reverse);
    const bool same_node = pos_1.node_id == pos_2.node_id;
    const bool same_strand = same_node && pos_1.is_reverse == pos_2.is_reverse;
    const bool inside_one_node = same_strand && pos_1.offset <= pos_2.offset;   // pos_2 lies ahead of pos_1 on their node: no search needed
    // bases a walk spends before it enters pos_2's node, at most:
    const int64_t reach = max_len - pos_2.offset;
    const int64_t out_of_first = (int64_t)source->get_length(h1) - pos_1.offset;

    Harvest seen;
    bool connected = false;
    if (inside_one_node) connected = pos_2.offset - pos_1.offset <= max_len;
    else {
        Frontier frontier;
        if (out_of_first <= reach) frontier.push(h1, out_of_first);
        handle_t here; int64_t dist;
        while (frontier.pop(here, dist)) {
            source->follow_edges_v(here, false, [&](const handle_t& next) {
                connected = connected || next == h2;
                seen.node(source->get_id(next));
                seen.edges.emplace_back(here, next);
                // neither anchor is walked through a second time in its own orientation: the walk ends at pos_2, and one that comes
                // back to pos_1's node from behind adds nothing
                const int64_t through = dist + (int64_t)source->get_length(next);
                if (next != h1 && next != h2 && through <= reach) frontier.push(next, through);
            });
        }
    }
    if (!connected) return result;

    // the nodes under their own ids, forward strands; the anchors' nodes first
    auto copy_node = [&](nid_t id) { into->create_handle(source->get_sequence(source->get_handle(id, false)), id); result.to_source[id] = id; };
    copy_node(pos_1.node_id);
    if (!same_node) copy_node(pos_2.node_id);
    for (nid_t id : seen.nodes) if (!result.to_source.count(id)) copy_node(id);
    for (const edge_t& e : seen.edges) into->create_edge(same_in(into, source, e.first), same_in(into, source, e.second));

    // the anchors become tips: cut their nodes at the positions, dropping the outer pieces with their edges
    const handle_t a1 = into->get_handle(pos_1.node_id, pos_1.is_reverse), a2 = into->get_handle(pos_2.node_id, pos_2.is_reverse);
    auto renamed = [&](nid_t was, const handle_t& now, nid_t stands_for) { if (was) result.to_source.erase(was); result.to_source[into->get_id(now)] = stands_for; };
    handle_t cut_1, cut_2;
    if (inside_one_node) {
        cut_1 = cut_2 = into->truncate_handle(into->truncate_handle(a2, false, (size_t)pos_2.offset), true, (size_t)pos_1.offset);
        renamed(pos_1.node_id, cut_1, pos_1.node_id);
    } else if (!same_node) {
        cut_1 = into->truncate_handle(a1, true, (size_t)pos_1.offset); renamed(pos_1.node_id, cut_1, pos_1.node_id);
        cut_2 = into->truncate_handle(a2, false, (size_t)pos_2.offset); renamed(pos_2.node_id, cut_2, pos_2.node_id);
    } else {
        // One node, and the walk has to leave it and come back (from behind on the same strand, or onto the other strand).  A second copy of
        // the node takes over what arrives at pos_2's side; the node itself keeps what leaves pos_1's.
        handle_t twin = into->create_handle(into->get_sequence(forward_of(into, a2)));
        if (into->get_is_reverse(a2)) twin = into->flip(twin);
        std::vector<edge_t> arriving;
        into->follow_edges_v(a2, true, [&](const handle_t& prev) {
            arriving.emplace_back(prev, twin);
            if (into->get_id(prev) == into->get_id(a2)) {                       // a loop: it also runs from the twin to the node and to the twin itself
                const handle_t twin_prev = prev == a2 ? twin : into->flip(twin);
                arriving.emplace_back(twin_prev, a2);
                arriving.emplace_back(twin_prev, twin);
            }
        });
        for (const edge_t& e : arriving) into->create_edge(e.first, e.second);
        cut_2 = into->truncate_handle(twin, false, (size_t)pos_2.offset); renamed(0, cut_2, pos_2.node_id);
        cut_1 = into->truncate_handle(a1, true, (size_t)pos_1.offset); renamed(pos_1.node_id, cut_1, pos_1.node_id);
    }

    // prune what no acceptable walk uses
    std::vector<nid_t> dead_nodes; std::vector<edge_t> dead_edges;
    if (strict_max_len) {
        // a node or an edge stays if the shortest walk through it, in either orientation, fits max_len
        const auto from_left = handlealgs::find_shortest_paths(into, cut_1, false), to_right = handlealgs::find_shortest_paths(into, cut_2, true);
        auto fits = [&](const handle_t& enter, const handle_t& leave, size_t inside) {
            const auto a = from_left.find(enter); const auto b = to_right.find(leave);
            return a != from_left.end() && b != to_right.end() && (int64_t)(a->second + inside + b->second) <= max_len;
        };
        into->for_each_handle_v([&](const handle_t& h) {
            const handle_t r = into->flip(h); const size_t l = into->get_length(h);
            if (!fits(h, h, l) && !fits(r, r, l)) { dead_nodes.push_back(into->get_id(h)); return; }
            auto check = [&](const handle_t& prev, const handle_t& next) {
                const size_t both = into->get_length(prev) + into->get_length(next);
                if (!fits(prev, next, both) && !fits(into->flip(next), into->flip(prev), both)) dead_edges.emplace_back(prev, next);
            };
            into->follow_edges_v(h, false, [&](const handle_t& next) { check(h, next); });
            into->follow_edges_v(h, true, [&](const handle_t& prev) { check(prev, h); });
        });
    } else {
        // forward reachability is what the search established; keep what also reaches the right anchor
        std::unordered_set<int64_t> reaches{cut_2.v};
        std::vector<handle_t> stack{cut_2};
        while (!stack.empty()) {
            const handle_t h = stack.back(); stack.pop_back();
            into->follow_edges_v(h, true, [&](const handle_t& prev) { if (reaches.insert(prev.v).second) stack.push_back(prev); });
        }
        into->for_each_handle_v([&](const handle_t& h) { if (!reaches.count(h.v) && !reaches.count(into->flip(h).v)) dead_nodes.push_back(into->get_id(h)); });
    }
    for (const edge_t& e : dead_edges)
        if (!std::count(dead_nodes.begin(), dead_nodes.end(), into->get_id(e.first)) && !std::count(dead_nodes.begin(), dead_nodes.end(), into->get_id(e.second)))
            into->destroy_edge(e.first, e.second);
    for (nid_t id : dead_nodes) { into->destroy_handle(into->get_handle(id, false)); result.to_source.erase(id); }
    result.left_id = into->has_node(into->get_id(cut_1)) ? into->get_id(cut_1) : 0;
    result.right_id = into->has_node(into->get_id(cut_2)) ? into->get_id(cut_2) : 0;
    if (result.to_source.size() != into->get_node_count()) throw std::logic_error("extract_connecting_graph: translation and graph disagree");
    return result;
}

ExtendingGraph extract_extending_graph(const HandleGraph* source, LocalGraph* into, int64_t max_dist, const Position& pos, bool backward,
                                       bool preserve_cycles_on_src_node) {
    if (into->get_node_count()) throw std::invalid_argument("extract_extending_graph: the output graph must be empty");
    ExtendingGraph result;
    const handle_t origin = source->get_handle(pos.node_id, false);
    const int64_t origin_len = (int64_t)source->get_length(origin);
    // the traversal of the start node that faces the way the search goes, and the bases left on it in that direction
    const handle_t outward = source->get_handle(pos.node_id, pos.is_reverse != backward);
    const int64_t left_on_node = backward ? pos.offset : origin_len - pos.offset;

    Harvest seen;
    bool came_back = false;
    Frontier frontier;
    if (left_on_node < max_dist) frontier.push(outward, left_on_node);        // otherwise the node alone holds enough sequence
    handle_t here; int64_t dist;
    while (frontier.pop(here, dist)) {
        source->follow_edges_v(here, false, [&](const handle_t& next) {
            came_back = came_back || source->get_id(next) == pos.node_id;
            seen.node(source->get_id(next));
            seen.edges.emplace_back(here, next);
            const int64_t through = dist + (int64_t)source->get_length(next);
            if (through < max_dist) frontier.push(next, through);
        });
    }
    const handle_t src_node = into->create_handle(source->get_sequence(origin), pos.node_id);
    result.to_source[pos.node_id] = pos.node_id;
    nid_t max_id = pos.node_id;
    for (nid_t id : seen.nodes) if (id != pos.node_id) {
        into->create_handle(source->get_sequence(source->get_handle(id, false)), id);
        result.to_source[id] = id; max_id = std::max(max_id, id);
    }
    // An edge on the side of the start node that the cut removes is left out: it arrives at `outward` or leaves its reverse.
    std::vector<edge_t> at_start;
    for (const edge_t& e : seen.edges) {
        const bool touches = source->get_id(e.first) == pos.node_id || source->get_id(e.second) == pos.node_id;
        if (touches) at_start.push_back(e);
        if (touches && (e.first == source->flip(outward) || e.second == outward)) continue;
        into->create_edge(same_in(into, source, e.first), same_in(into, source, e.second));
    }
    if (came_back && preserve_cycles_on_src_node) {
        // a whole second copy of the start node keeps the walks that return to it alive behind the cut
        const handle_t twin = into->create_handle(source->get_sequence(origin), max_id + 1);
        result.to_source[into->get_id(twin)] = pos.node_id;
        const bool outward_rev = source->get_is_reverse(outward);
        auto on = [&](const handle_t& node, const handle_t& like) { return source->get_is_reverse(like) ? into->flip(node) : node; };
        for (const edge_t& e : at_start) {
            const bool first_here = source->get_id(e.first) == pos.node_id, second_here = source->get_id(e.second) == pos.node_id;
            if (first_here && second_here) {
                into->create_edge(on(twin, e.first), on(twin, e.second));
                // an end on the side that survives the cut also links the node with its twin
                if (source->get_is_reverse(e.first) == outward_rev) into->create_edge(on(src_node, e.first), on(twin, e.second));
                else if (source->get_is_reverse(e.second) != outward_rev) into->create_edge(on(twin, e.first), on(src_node, e.second));
            } else if (first_here) into->create_edge(on(twin, e.first), same_in(into, source, e.second));
            else into->create_edge(same_in(into, source, e.first), on(twin, e.second));
        }
    }
    // cut the start node at the position; the half the search did not leave through goes
    const size_t at = (size_t)(pos.is_reverse ? origin_len - pos.offset : pos.offset);
    const auto halves = into->divide_handle(src_node, at);
    result.to_source.erase(pos.node_id);
    const bool keep_second = pos.is_reverse == backward;
    into->destroy_handle(keep_second ? halves.first : halves.second);
    const handle_t kept = keep_second ? halves.second : halves.first;
    result.to_source[into->get_id(kept)] = pos.node_id;
    result.cut_id = into->get_id(kept);
    if (result.to_source.size() != into->get_node_count()) throw std::logic_error("extract_extending_graph: translation and graph disagree");
    return result;
}

void extract_containing_graph(const HandleGraph* source, LocalGraph* into, const std::vector<Position>& positions,
                              const std::vector<size_t>& forward_search_lengths, const std::vector<size_t>& backward_search_lengths,
                              size_t reversing_walk_length) {
    if (forward_search_lengths.size() != positions.size() || backward_search_lengths.size() != positions.size())
        throw std::invalid_argument("extract_containing_graph: one forward and one backward search length per position");
    if (into->get_node_count()) throw std::invalid_argument("extract_containing_graph: the output graph must be empty");
    if (positions.empty()) return;
    // One shortest-first search serves every position: a search that may go less far starts with the difference already spent
    // (src/algorithms/extract_containing_graph.cpp:41-47).  The key of an oriented node is the distance walked when it is entered.
    int64_t longest = 0;
    for (size_t i = 0; i < positions.size(); ++i) longest = std::max<int64_t>(longest, (int64_t)std::max(forward_search_lengths[i], backward_search_lengths[i]));
    // (the reference's queue re-prioritises: an oriented node is handed out once, at the smallest key pushed for it before that)
    std::map<int64_t, int64_t> best; std::priority_queue<std::pair<int64_t, int64_t>, std::vector<std::pair<int64_t, int64_t>>, std::greater<std::pair<int64_t, int64_t>>> todo;
    std::unordered_set<int64_t> done;
    auto push = [&](const handle_t& h, int64_t dist) {
        if (done.count(h.v)) return;
        auto found = best.find(h.v);
        if (found != best.end() && found->second <= dist) return;
        best[h.v] = dist; todo.emplace(dist, h.v);
    };
    Harvest seen;
    for (size_t i = 0; i < positions.size(); ++i) {
        const Position& pos = positions[i];
        const handle_t fwd = source->get_handle(pos.node_id, false);
        seen.node(pos.node_id);
        const int64_t dist_forward = -pos.offset + longest - (int64_t)forward_search_lengths[i];
        const int64_t dist_backward = pos.offset - (int64_t)source->get_length(fwd) + longest - (int64_t)backward_search_lengths[i];
        push(pos.is_reverse ? source->flip(fwd) : fwd, dist_forward);
        push(pos.is_reverse ? fwd : source->flip(fwd), dist_backward);
    }
    while (!todo.empty()) {
        const auto top = todo.top(); todo.pop();
        if (done.count(top.second) || best[top.second] != top.first) continue;
        done.insert(top.second);
        const handle_t here{top.second};
        seen.node(source->get_id(here));
        const int64_t through = top.first + (int64_t)source->get_length(here);
        if (through < longest)
            source->follow_edges_v(here, false, [&](const handle_t& next) { seen.edges.emplace_back(here, next); push(next, through); });
        if (reversing_walk_length > 0 && top.first > 0) {       // (not from a start: their keys are <= 0 ... as in the reference)
            const handle_t flipped = source->flip(here);
            source->follow_edges_v(flipped, false, [&](const handle_t& next) { seen.edges.emplace_back(flipped, next); push(next, longest - (int64_t)reversing_walk_length); });
        }
    }
    for (nid_t id : seen.nodes) into->create_handle(source->get_sequence(source->get_handle(id, false)), id);
    for (const edge_t& e : seen.edges) into->create_edge(same_in(into, source, e.first), same_in(into, source, e.second));
}

}  // namespace vgamd
