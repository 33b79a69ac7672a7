// STAND-IN (vg_amd/host/vg_standin/): restates a slice of vg / libhandlegraph / libvgio that stays vg's own in a real
// integration; present only so the reference's unit tests can be driven without vg.  Excluded from size / originality claims.
// handle_graph.hpp — the slice of libhandlegraph's read-only HandleGraph
// interface that vg's alignment hot path touches (reference: src/handle.hpp:8-40;
// used by src/aligner.cpp:30-118, src/dozeu_interface.cpp:210-307,
// src/banded_global_aligner.cpp:1960-2109), plus the three graph views applied
// right around the kernels (src/reverse_graph.cpp, src/null_masking_graph.cpp).
//
// libhandlegraph itself is an empty submodule in the reference snapshot, so this
// is a from-scratch minimal interface with the same method names and meaning;
// a vg maintainer would instead pass vg's own HandleGraph and delete this file.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace vgamd {

using nid_t = int64_t;

struct handle_t {
    int64_t v = 0;                               // (id << 1) | is_reverse, like most libhandlegraph impls
    bool operator==(const handle_t& o) const { return v == o.v; }
    bool operator!=(const handle_t& o) const { return v != o.v; }
    bool operator<(const handle_t& o) const { return v < o.v; }
};
struct handle_hash { size_t operator()(const handle_t& h) const { return std::hash<int64_t>()(h.v); } };
using edge_t = std::pair<handle_t, handle_t>;

class HandleGraph {
public:
    virtual ~HandleGraph() = default;
    virtual bool has_node(nid_t id) const = 0;
    virtual handle_t get_handle(nid_t id, bool is_reverse = false) const { return handle_t{(id << 1) | (is_reverse ? 1 : 0)}; }
    virtual nid_t get_id(const handle_t& h) const { return h.v >> 1; }
    virtual bool get_is_reverse(const handle_t& h) const { return h.v & 1; }
    virtual handle_t flip(const handle_t& h) const { return handle_t{h.v ^ 1}; }
    virtual size_t get_length(const handle_t& h) const = 0;
    virtual std::string get_sequence(const handle_t& h) const = 0;
    // iteratee returns false to stop; returns false if stopped early
    virtual bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const = 0;
    virtual bool for_each_handle(const std::function<bool(const handle_t&)>& it) const = 0;
    virtual size_t get_node_count() const = 0;
    virtual nid_t min_node_id() const = 0;
    virtual nid_t max_node_id() const = 0;

    // void-returning conveniences (libhandlegraph accepts both)
    void follow_edges_v(const handle_t& h, bool go_left, const std::function<void(const handle_t&)>& it) const {
        follow_edges(h, go_left, [&](const handle_t& n) { it(n); return true; });
    }
    void for_each_handle_v(const std::function<void(const handle_t&)>& it) const {
        for_each_handle([&](const handle_t& n) { it(n); return true; });
    }
    size_t get_degree(const handle_t& h, bool go_left) const {
        size_t d = 0; follow_edges_v(h, go_left, [&](const handle_t&) { ++d; }); return d;
    }
};

// A plain mutable graph (stands in for bdsg::HashGraph / vg::VG in tests and tools).
class HashGraph : public HandleGraph {
public:
    handle_t create_handle(const std::string& seq);                 // ids 1,2,3,... like VG::create_node
    handle_t create_handle(const std::string& seq, nid_t id);
    void create_edge(const handle_t& from, const handle_t& to);     // forward-strand edges only
    bool has_node(nid_t id) const override { return index_.count(id) != 0; }
    size_t get_length(const handle_t& h) const override { return seqs_[index_.at(get_id(h))].size(); }
    std::string get_sequence(const handle_t& h) const override;
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return ids_.size(); }
    nid_t min_node_id() const override { return min_id_; }
    nid_t max_node_id() const override { return max_id_; }
private:
    std::vector<nid_t> ids_;                       // insertion order
    std::vector<std::string> seqs_;
    std::vector<std::vector<nid_t>> out_, in_;     // forward-strand adjacency, insertion order
    std::unordered_map<nid_t, size_t> index_;
    nid_t min_id_ = 0, max_id_ = 0, next_id_ = 1;
};

// ReverseGraph(&g, false): every sequence reversed (NOT complemented), every edge
// flipped (reference: src/reverse_graph.cpp:54-57 and get_sequence above it).
class ReverseGraph : public HandleGraph {
public:
    ReverseGraph(const HandleGraph* g, bool complement) : g_(g), complement_(complement) {}
    bool has_node(nid_t id) const override { return g_->has_node(id); }
    size_t get_length(const handle_t& h) const override { return g_->get_length(h); }
    std::string get_sequence(const handle_t& h) const override;
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override {
        return g_->follow_edges(h, !go_left, it);
    }
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override { return g_->for_each_handle(it); }
    size_t get_node_count() const override { return g_->get_node_count(); }
    nid_t min_node_id() const override { return g_->min_node_id(); }
    nid_t max_node_id() const override { return g_->max_node_id(); }
private:
    const HandleGraph* g_; bool complement_;
};

// NullMaskingGraph: hides zero-length nodes and the edges touching them; it does
// NOT bridge their neighbours (reference: src/null_masking_graph.cpp:56-80).
class NullMaskingGraph : public HandleGraph {
public:
    explicit NullMaskingGraph(const HandleGraph* g);
    bool has_node(nid_t id) const override { return g_->has_node(id) && g_->get_length(g_->get_handle(id)) > 0; }
    size_t get_length(const handle_t& h) const override { return g_->get_length(h); }
    std::string get_sequence(const handle_t& h) const override { return g_->get_sequence(h); }
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return g_->get_node_count() - nulls_; }
    nid_t min_node_id() const override { return g_->min_node_id(); }
    nid_t max_node_id() const override { return g_->max_node_id(); }
private:
    const HandleGraph* g_; size_t nulls_ = 0;
};

// DozeuPinningOverlay: removes empty nodes and duplicates any neighbour of an empty tip that would
// otherwise lose its tip status, so that dozeu can pin on it (reference: src/dozeu_pinning_overlay.cpp,
// used at src/aligner.cpp:641).  Duplicate handles / ids live above the underlying ranges.
class DozeuPinningOverlay : public HandleGraph {
public:
    DozeuPinningOverlay(const HandleGraph* graph, bool preserve_sinks);
    bool performed_duplications() const { return !duplicated_handles.empty(); }
    bool has_node(nid_t id) const override;
    handle_t get_handle(nid_t id, bool is_reverse = false) const override;
    nid_t get_id(const handle_t& h) const override;
    bool get_is_reverse(const handle_t& h) const override;
    handle_t flip(const handle_t& h) const override;
    size_t get_length(const handle_t& h) const override { return graph->get_length(get_underlying_handle(h)); }
    std::string get_sequence(const handle_t& h) const override { return graph->get_sequence(get_underlying_handle(h)); }
    bool follow_edges(const handle_t& h, bool go_left, const std::function<bool(const handle_t&)>& it) const override;
    bool for_each_handle(const std::function<bool(const handle_t&)>& it) const override;
    size_t get_node_count() const override { return graph->get_node_count() - num_null_nodes + duplicated_handles.size(); }
    nid_t min_node_id() const override { return graph->min_node_id(); }
    nid_t max_node_id() const override;
    handle_t get_underlying_handle(const handle_t& h) const;
private:
    bool is_a_duplicate_handle(const handle_t& h) const { return (uint64_t)h.v > max_handle; }
    bool is_a_duplicate_id(nid_t id) const { return id > graph->max_node_id(); }
    nid_t get_underlying_id(nid_t id) const { return id - (graph->max_node_id() - graph->min_node_id() + 1); }
    handle_t get_duplicate_handle(const handle_t& h) const { return handle_t{(int64_t)((uint64_t)h.v + handle_val_range)}; }
    const HandleGraph* graph; bool preserve_sinks;
    std::set<handle_t> duplicated_handles;          // forward handles of duplicated nodes (ordered: deterministic iteration)
    size_t num_null_nodes = 0; uint64_t max_handle = 0, handle_val_range = 0;
};

namespace handlealgs {
// Kahn's algorithm, always expanding the smallest ready handle (the ordered
// "s" map of vg's topological sort); on a DAG with forward handles this is what
// lazier_topological_order yields.  PARITY-UNPINNED: libhandlegraph is absent.
std::vector<handle_t> lazier_topological_order(const HandleGraph* g);
std::vector<handle_t> head_nodes(const HandleGraph* g);
std::vector<handle_t> tail_nodes(const HandleGraph* g);
}

}  // namespace vgamd
