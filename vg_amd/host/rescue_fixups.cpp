// rescue_fixups.cpp — see rescue_fixups.hpp.
#include "rescue_fixups.hpp"

namespace vgamd {

void fix_dozeu_score(Alignment& rescued_alignment, const Aligner& aligner, const HandleGraph& rescue_graph, const std::vector<handle_t>& topological_order) {
    const int32_t score = aligner.scorer->score_contiguous_alignment(rescued_alignment);
    if (score > 0) { rescued_alignment.score = score; return; }
    rescued_alignment.clear_path();                              // not worth keeping: the full DP instead (:3510-3515)
    if (topological_order.empty()) aligner.align(rescued_alignment, rescue_graph, true);
    else aligner.align(rescued_alignment, rescue_graph, topological_order);
}

void fix_dozeu_end_deletions(Alignment& alignment, bool reference_indexing) {
    std::vector<Mapping>& mappings = alignment.path.mapping;
    // the first edit that consumes read bases: mapping i, edit j (:3521-3533)
    size_t i = 0, j = 0;
    for (; i < mappings.size(); ++i) {
        const Mapping& m = mappings[i];
        for (j = 0; j < m.edit.size(); ++j) if (m.edit[j].to_length != 0) break;
        if (j != m.edit.size()) break;
    }
    if (i == mappings.size()) { alignment.clear_path(); return; }        // nothing but deletions (:3534-3537; the right-hand loop below finds nothing to do then)
    if (i != 0 || j != 0) {
        // The reference takes the edits to drop from `(*mappings)[j]` — the EDIT index used as a mapping index (:3541) — where mapping i is
        // evidently meant; the two agree in its unit test (i = j = 1).  With i != j the reference erases from the wrong mapping and the
        // result no longer consumes the read: the default here is the evident intent, mapping i (a valid alignment);
        // reference_indexing = true keeps the line as written wherever that element exists (where it does not the reference reads past
        // the end: mapping i then too).  [PARITY-UNPINNED for i != j: the reference holds no test there.]
        Mapping& from = (reference_indexing && j < mappings.size()) ? mappings[j] : mappings[i];
        size_t removed = 0;
        const size_t drop = j < from.edit.size() ? j : from.edit.size();
        for (size_t k = 0; k < drop; ++k) removed += (size_t)from.edit[k].from_length;
        from.edit.erase(from.edit.begin(), from.edit.begin() + (std::ptrdiff_t)drop);
        mappings.erase(mappings.begin(), mappings.begin() + (std::ptrdiff_t)i);
        mappings[0].position.offset += (int64_t)removed;
    }
    // deletions on the right (:3552-3564)
    while (!mappings.empty()) {
        std::vector<Edit>& edits = mappings.back().edit;
        while (!edits.empty() && edits.back().to_length == 0) edits.pop_back();
        if (edits.empty()) mappings.pop_back(); else break;
    }
}

}  // namespace vgamd
