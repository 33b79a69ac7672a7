// gbwt_types.cpp — STAND-IN for code that stays vg's own in a real integration.
//
// Everything in vg_amd/host/vg_standin/ restates, closely, small pieces of vg's host-side glue (result types, graph views, scorer
// tables) so that the reference's unit tests can be driven through the shim without vg: the arithmetic must round and tie exactly
// as vg's does, so these pieces follow the reference nearly line for line (citations inline).  None of it is part of the engine,
// none of it would ship inside vg (vg keeps its own classes and calls include/vgk.h), and it is excluded from any
// size or originality claim made for this repository.
#include "../gbwt_extender.hpp"
#include <algorithm>
#include <stdexcept>

namespace vgamd {

// ---- GaplessExtension (src/gbwt_extender.cpp:17-151) -----------------------------------------------------------------
bool GaplessExtension::contains(const HandleGraph& graph, const seed_type& seed) const {
    size_t read_offset = read_interval.first, node_offset = offset;
    for (const handle_t& handle : path) {
        const size_t len = std::min(graph.get_length(handle) - node_offset, read_interval.second - read_offset);
        if (seed_type(handle, (int64_t)read_offset - (int64_t)node_offset) == seed) return true;
        read_offset += len; node_offset = 0;
    }
    return false;
}
Position GaplessExtension::starting_position(const HandleGraph& graph) const {
    Position p;
    if (empty()) return p;
    p.node_id = graph.get_id(path.front()); p.is_reverse = graph.get_is_reverse(path.front()); p.offset = (int64_t)offset;
    return p;
}
size_t GaplessExtension::tail_offset(const HandleGraph& graph) const {
    size_t result = offset + length();
    for (size_t i = 0; i + 1 < path.size(); ++i) result -= graph.get_length(path[i]);
    return result;
}
Position GaplessExtension::tail_position(const HandleGraph& graph) const {
    Position p;
    if (empty()) return p;
    p.node_id = graph.get_id(path.back()); p.is_reverse = graph.get_is_reverse(path.back()); p.offset = (int64_t)tail_offset(graph);
    return p;
}
size_t GaplessExtension::overlap(const HandleGraph& graph, const GaplessExtension& another) const {
    size_t result = 0, this_pos = read_interval.first, another_pos = another.read_interval.first;
    auto this_iter = path.begin(), another_iter = another.path.begin();
    size_t this_offset = offset, another_offset = another.offset;
    while (this_pos < read_interval.second && another_pos < another.read_interval.second) {
        if (this_pos == another_pos && *this_iter == *another_iter && this_offset == another_offset) {
            const size_t len = std::min({graph.get_length(*this_iter) - this_offset, read_interval.second - this_pos, another.read_interval.second - another_pos});
            result += len; this_pos += len; another_pos += len; ++this_iter; ++another_iter; this_offset = another_offset = 0;
        } else if (this_pos <= another_pos) { this_pos += graph.get_length(*this_iter) - this_offset; ++this_iter; this_offset = 0; }
        else { another_pos += graph.get_length(*another_iter) - another_offset; ++another_iter; another_offset = 0; }
    }
    return result;
}
Path GaplessExtension::to_path(const HandleGraph& graph, const std::string& sequence) const {
    Path result;
    auto mismatch = mismatch_positions.begin();
    size_t read_offset = read_interval.first, node_offset = offset;
    for (size_t i = 0; i < path.size(); ++i) {
        const size_t limit = std::min(read_offset + graph.get_length(path[i]) - node_offset, read_interval.second);
        result.mapping.emplace_back();
        Mapping& mapping = result.mapping.back();
        mapping.position.node_id = graph.get_id(path[i]); mapping.position.offset = (int64_t)node_offset; mapping.position.is_reverse = graph.get_is_reverse(path[i]);
        while (mismatch != mismatch_positions.end() && *mismatch < limit) {
            if (read_offset < *mismatch) { Edit e; e.from_length = e.to_length = (int32_t)(*mismatch - read_offset); mapping.edit.push_back(e); }
            Edit e; e.from_length = e.to_length = 1; e.sequence = std::string(1, sequence[*mismatch]); mapping.edit.push_back(e);
            read_offset = *mismatch + 1; ++mismatch;
        }
        if (read_offset < limit) { Edit e; e.from_length = e.to_length = (int32_t)(limit - read_offset); mapping.edit.push_back(e); read_offset = limit; }
        mapping.rank = (int64_t)i + 1;
        node_offset = 0;
    }
    return result;
}

// ---- WFAAlignment (src/gbwt_extender.cpp:761-1123) --------------------------------------------------------------------
WFAAlignment WFAAlignment::from_extension(const GaplessExtension& extension) {
    WFAAlignment a;
    a.path = extension.path; a.node_offset = (uint32_t)extension.offset; a.seq_offset = (uint32_t)extension.read_interval.first;
    a.length = (uint32_t)extension.length(); a.score = extension.score; a.ok = true;
    size_t done = a.seq_offset;                       // sequence position after the last edit
    for (size_t at : extension.mismatch_positions) {
        if (!a.edits.empty() && done == at && a.edits.back().first == mismatch) ++a.edits.back().second;
        else {
            if (done < at) a.edits.emplace_back(match, (uint32_t)(at - done));
            a.edits.emplace_back(mismatch, 1u);
        }
        done = at;                                    // as the reference: the cursor stops ON the mismatch
    }
    if (done < a.seq_offset + a.length) a.edits.emplace_back(match, (uint32_t)(a.seq_offset + a.length - done));
    return a;
}
WFAAlignment WFAAlignment::make_unlocalized_insertion(size_t sequence_offset, size_t length, int score) {
    WFAAlignment a; a.edits.emplace_back(insertion, (uint32_t)length); a.seq_offset = (uint32_t)sequence_offset; a.length = (uint32_t)length; a.score = score; a.ok = true;
    return a;
}
WFAAlignment WFAAlignment::make_empty() { WFAAlignment a; a.ok = true; return a; }
bool WFAAlignment::unlocalized_insertion() const { return ok && path.empty() && edits.size() == 1 && edits.front().first == insertion; }
int64_t WFAAlignment::final_offset(const HandleGraph& graph) const {
    int64_t f = node_offset;
    for (const auto& e : edits) if (e.first != insertion) f += e.second;
    for (size_t i = 0; i + 1 < path.size(); ++i) f -= (int64_t)graph.get_length(path[i]);
    return f;
}
void WFAAlignment::flip(const HandleGraph& graph, const std::string& sequence) {
    seq_offset = (uint32_t)(sequence.length() - seq_offset - length);
    if (path.empty()) return;
    node_offset = (uint32_t)((int64_t)graph.get_length(path.back()) - final_offset(graph));
    std::reverse(path.begin(), path.end());
    for (handle_t& h : path) h = graph.flip(h);
    std::reverse(edits.begin(), edits.end());
}
void WFAAlignment::append(Edit edit, uint32_t len) {
    if (len == 0) return;
    if (edits.empty() || edits.back().first != edit) edits.emplace_back(edit, len);
    else edits.back().second += len;
}
void WFAAlignment::join(const WFAAlignment& second) {
    if (!ok) throw std::runtime_error("Cannot join onto an alignment that is not OK");
    if (!second.ok) throw std::runtime_error("Cannot join an alignment that is not OK onto another alignment");
    if (second.empty()) return;
    if (empty()) { *this = second; return; }
    if (seq_offset + length != second.seq_offset)
        throw std::runtime_error("Cannot join alignments because past-end position " + std::to_string(seq_offset + length) + " is not at start position " + std::to_string(second.seq_offset));
    if (path.empty() && !unlocalized_insertion()) throw std::runtime_error("Cannot join alignments because first alignment has no path");
    if (second.path.empty() && !second.unlocalized_insertion()) throw std::runtime_error("Cannot join alignments because second alignment has no path");
    if (edits.empty()) throw std::runtime_error("Cannot join alignments because first alignment has no edits");
    if (second.edits.empty()) throw std::runtime_error("Cannot join alignments because second alignment has no edits");
    if (!second.unlocalized_insertion()) {
        if (unlocalized_insertion()) { node_offset = second.node_offset; path.push_back(second.path.front()); }
        else if (second.node_offset == 0) path.push_back(second.path.front());
        else if (second.path.front() != path.back())
            throw std::runtime_error("Cannot join alignments because second alignment starts in the middle of a handle that first alignment doesn't end on");
        path.insert(path.end(), second.path.begin() + 1, second.path.end());
    }
    for (const auto& e : second.edits) append(e.first, e.second);
    length += second.length;
    score += second.score;
}
Path WFAAlignment::to_path(const HandleGraph& graph, const std::string& sequence) const {
    if (!ok) throw std::runtime_error("WFAAlignment is not OK and cannot become a path");
    if ((size_t)seq_offset + length > sequence.size()) throw std::runtime_error("WFAAlignment extends past end of sequence");
    Path result;
    if (unlocalized_insertion()) {
        result.mapping.emplace_back();
        vgamd::Edit e; e.to_length = (int32_t)edits.front().second; e.sequence = sequence.substr(seq_offset, edits.front().second);
        result.mapping.back().edit.push_back(e);
        return result;
    }
    if (path.empty()) return result;
    size_t seq_at = seq_offset, node_at = node_offset, step = 0;
    size_t node_end = graph.get_length(path[0]);
    if (node_offset >= node_end) throw std::runtime_error("WFAAlignment has offset to or past end of first node");
    if (edits.empty()) throw std::runtime_error("WFAAlignment has no edits");
    auto open_mapping = [&](size_t offset) {
        result.mapping.emplace_back();
        Position& p = result.mapping.back().position;
        p.node_id = graph.get_id(path[step]); p.is_reverse = graph.get_is_reverse(path[step]); p.offset = (int64_t)offset;
    };
    open_mapping(node_at);
    for (const auto& ed : edits) {
        if (ed.second == 0) throw std::runtime_error("WFAAlignment has empty edit");
        const bool uses_graph = ed.first != insertion, uses_seq = ed.first != deletion;
        size_t left = ed.second;
        while (left) {
            size_t take = left;
            if (uses_graph) {
                if (step == path.size()) throw std::runtime_error("WFAAlignment tried to go past end of path");
                if (node_at == node_end) throw std::runtime_error("WFAAlignment tried to go past end of node (" + std::to_string(node_end) + " bp)");
                take = std::min(take, node_end - node_at);
            }
            vgamd::Edit e;
            if (uses_graph) { e.from_length = (int32_t)take; node_at += take; }
            if (ed.first == mismatch || ed.first == insertion) {
                if (seq_at + take > (size_t)seq_offset + length) throw std::runtime_error("WFAAlignment uses more sequence than provided");
                e.sequence = sequence.substr(seq_at, take);
            }
            if (uses_seq) { e.to_length = (int32_t)take; seq_at += take; }
            result.mapping.back().edit.push_back(e);
            left -= take;
            if (uses_graph && node_at == node_end) {
                node_at = 0; ++step;
                if (step != path.size()) {
                    node_end = graph.get_length(path[step]);
                    if (node_end == 0) throw std::runtime_error("WFAAlignment has empty node " + std::to_string(graph.get_id(path[step])));
                    open_mapping(0);
                } else node_end = 0;
            }
        }
    }
    return result;
}

// ---- WFAExtender ------------------------------------------------------------------------------------------------------
const WFAExtender::ErrorModel WFAExtender::default_error_model { WFAExtender::ErrorModel::default_mismatches(), WFAExtender::ErrorModel::default_gaps(),
                                                                 WFAExtender::ErrorModel::default_gap_length(), WFAExtender::ErrorModel::default_distance() };

}  // namespace vgamd
