// rescue_fixups.hpp — what MinimizerMapper::attempt_rescue does to an alignment that came back from Aligner::align_xdrop
// (reference: src/minimizer_mapper.cpp:3382-3389, :3425-3429): the score dozeu reports is replaced by the scorer's own, an alignment the
// scorer does not like is redone with the full DP, and deletions dozeu left at either end are cut off.  Host bookkeeping beside row a12.
#pragma once
#include <vector>
#include "aligner.hpp"

namespace vgamd {

// MinimizerMapper::fix_dozeu_score (src/minimizer_mapper.cpp:3502-3517)
void fix_dozeu_score(Alignment& rescued_alignment, const Aligner& aligner, const HandleGraph& rescue_graph, const std::vector<handle_t>& topological_order);
// MinimizerMapper::fix_dozeu_end_deletions (src/minimizer_mapper.cpp:3519-3565)
// (reference_indexing: the reference indexes the mappings with the EDIT index when it drops the leading deletion, :3541 — right only when
//  the two agree; the default takes the mapping that holds the first read-consuming edit, which is what the code around it means)
void fix_dozeu_end_deletions(Alignment& alignment, bool reference_indexing = false);

}  // namespace vgamd
