// extension_scoring.hpp — what giraffe thinks a group of gapless extensions of one read is worth before it aligns anything:
// MinimizerMapper::score_extension_group / score_extensions (reference src/minimizer_mapper.cpp:5022-5262).  Full-length extensions are
// worth their own score; otherwise the best chain of extensions along the read, a chain paying for the read bases it skips between two
// extensions (gap open + extend per base) or for stepping back into an overlap.  Host arithmetic beside the extension stage;
// the reference holds no test vectors for it [PARITY-UNPINNED]: tests/test_extension_scoring.py pins it on an independent recurrence over
// the same model (quadratic in the extensions, no sweep line, no heaps).
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace vgamd {

struct ScoredInterval { size_t begin = 0, end = 0; int32_t score = 0; };      // an extension's read interval [begin, end) and score

// `extensions` in the order GaplessExtender::extend returns them (by read interval start); `full_length` = GaplessExtender::full_length_extensions
// of the group (the first extension spans the read within the mismatch bound)
int score_extension_group(size_t read_length, const std::vector<ScoredInterval>& extensions, bool full_length, int gap_open_penalty, int gap_extend_penalty);

}  // namespace vgamd
