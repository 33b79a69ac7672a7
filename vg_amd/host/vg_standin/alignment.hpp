// STAND-IN (vg_amd/host/vg_standin/): restates a slice of vg / libhandlegraph / libvgio that stays vg's own in a real
// integration; present only so the reference's unit tests can be driven without vg.  Excluded from size / originality claims.
// alignment.hpp — plain-struct mirror of the vg.proto messages that cross the
// aligner boundary (libvgio is an empty submodule in the reference; shapes are
// taken from their uses: src/aligner.cpp:125-128,152-161,185-228,240,
// src/dozeu_interface.cpp:313-332,342-357, src/banded_global_aligner.cpp:108-132).
// Accessor names follow the protobuf-generated API so call sites read alike.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vgamd {

struct Position { int64_t node_id = 0; int64_t offset = 0; bool is_reverse = false; };

struct Edit {
    int32_t from_length = 0, to_length = 0;
    std::string sequence;
};
inline bool edit_is_match(const Edit& e) { return e.from_length == e.to_length && e.from_length > 0 && e.sequence.empty(); }
inline bool edit_is_sub(const Edit& e) { return e.from_length == e.to_length && e.from_length > 0 && !e.sequence.empty(); }
inline bool edit_is_insertion(const Edit& e) { return e.from_length == 0 && e.to_length > 0; }
inline bool edit_is_deletion(const Edit& e) { return e.from_length > 0 && e.to_length == 0; }

struct Mapping {
    Position position;
    std::vector<Edit> edit;
    int64_t rank = 0;
};

struct Path { std::vector<Mapping> mapping; };

struct Alignment {
    std::string sequence;
    std::string quality;          // raw phred bytes (not ASCII-33), as in vg
    std::string name;
    Path path;
    int32_t score = 0;
    double identity = 0.0;
    int64_t query_position = 0;
    bool has_path() const { return !path.mapping.empty(); }
    void clear_path() { path.mapping.clear(); }
};

inline int mapping_from_length(const Mapping& m) { int n = 0; for (auto& e : m.edit) n += e.from_length; return n; }
inline int mapping_to_length(const Mapping& m) { int n = 0; for (auto& e : m.edit) n += e.to_length; return n; }
inline int path_from_length(const Path& p) { int n = 0; for (auto& m : p.mapping) n += mapping_from_length(m); return n; }
inline int path_to_length(const Path& p) { int n = 0; for (auto& m : p.mapping) n += mapping_to_length(m); return n; }

// identity(path): matched bases / aligned read bases, soft clips at either end
// excluded from the denominator (reference: src/path.cpp:2316-2335)
inline double identity(const Path& p) {
    size_t total = (size_t)path_to_length(p), matched = 0;
    for (size_t i = 0; i < p.mapping.size(); ++i) {
        const Mapping& m = p.mapping[i];
        for (size_t j = 0; j < m.edit.size(); ++j) {
            const Edit& e = m.edit[j];
            if (edit_is_match(e)) matched += e.from_length;
            else if (edit_is_insertion(e)) {
                bool first = (i == 0) && (j == 0);
                bool last = (i + 1 == p.mapping.size()) && (j + 1 == m.edit.size());
                if (first || last) total -= e.to_length;
            }
        }
    }
    return total == 0 ? 0.0 : (double)matched / (double)total;
}

std::string alignment_to_json(const Alignment& a);

}  // namespace vgamd
