// extension_scoring.cpp — see extension_scoring.hpp.
#include "extension_scoring.hpp"
#include <algorithm>
#include <functional>
#include <limits>
#include <queue>

namespace vgamd {

int score_extension_group(size_t read_length, const std::vector<ScoredInterval>& ext, bool full_length, int go, int ge) {
    if (ext.empty()) return 0;
    if (full_length) return ext.front().score;                 // (:5030-5032)
    if (read_length == 0) return 0;
    const int64_t L = (int64_t)read_length;
    // A sweep over the read positions where an extension starts or ends (:5037-5243).  `next_unswept` is the first base the previous
    // stop has not covered.
    int64_t next_unswept = 0;
    size_t entering = 0;                                        // the next extension to start
    typedef std::pair<size_t, size_t> EndItem;                  // (past-end position, extension)
    std::priority_queue<EndItem, std::vector<EndItem>, std::greater<EndItem>> ends;      // min-heap on the past-end position
    // backtracking into an overlap: (score if we step back to the current position, past-end of the extension stepped back from), scores
    // kept relative to a counter that grows by a gap extension per base swept, so that the heap never needs re-sorting
    std::priority_queue<std::pair<int, size_t>> overlaps;
    int overlap_offset = 0;
    int best_gap = 0;                                           // best chain ending in a gap just before here (0: none worth it)
    std::vector<int> chain(ext.size(), 0);                      // best chain ending with each extension
    int best_ever = 0;
    while (next_unswept <= L) {
        int64_t stop = L;
        if (entering < ext.size()) stop = std::min<int64_t>(stop, (int64_t)ext[entering].begin);
        if (!ends.empty()) stop = std::min<int64_t>(stop, (int64_t)ends.top().first);
        const int swept = (int)(stop - next_unswept + 1);
        int ended_here = 0;                                     // best chain whose last extension past-ends exactly here
        while (!ends.empty() && (int64_t)ends.top().first == stop) { ended_here = std::max(ended_here, chain[ends.top().second]); ends.pop(); }
        best_ever = std::max(best_ever, ended_here);
        if (stop == L) break;
        overlap_offset += swept * ge;
        int best_overlap = 0;
        while (!overlaps.empty()) {
            if ((int64_t)overlaps.top().second <= stop) { overlaps.pop(); continue; }      // we are past it already
            best_overlap = overlaps.top().first + overlap_offset;
            break;
        }
        if (best_gap != 0) best_gap -= swept * ge;
        best_gap = std::max(0, std::max(best_gap, ended_here - (go - ge)));
        while (entering < ext.size() && (int64_t)ext[entering].begin == stop) {
            chain[entering] = std::max(best_overlap, std::max(best_gap, ended_here)) + ext[entering].score;
            const size_t length = ext[entering].end - ext[entering].begin;
            overlaps.emplace(chain[entering] - go - ge * (int)length - overlap_offset, ext[entering].end);
            ends.emplace(ext[entering].end, entering);
            ++entering;
        }
        next_unswept = stop + 1;
    }
    return best_ever;
}

}  // namespace vgamd
