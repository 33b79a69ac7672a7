// seed_policy.cpp — see seed_policy.hpp.
#include "seed_policy.hpp"
#include <algorithm>
#include <cmath>
#include <deque>
#include <map>
#include <stdexcept>
#include <random>
#include <string>
#include <unordered_set>

namespace vgamd {

// The windows are [w, w + window_size) for w = 0 .. sequence_length - window_size; an element lies in a window when all of it does.  Between
// two "events" — the next element coming in, the front element dropping out — nothing changes, so the sweep jumps from event to event
// (src/algorithms/sample_minimal.cpp:89-180).  `line` holds the window's elements that no later element has displaced, in order of start.
void sample_minimal(size_t count, size_t element_length, size_t window_size, size_t sequence_length, const std::function<size_t(size_t)>& get_start,
                    const std::function<bool(size_t, size_t)>& should_beat, const std::function<void(size_t)>& sample) {
    if (!count) return;
    std::deque<size_t> line;
    size_t next = 0;
    auto admit = [&]() {                                         // the next element enters: whatever it displaces leaves from the back
        while (!line.empty() && should_beat(next, line.back())) line.pop_back();
        line.push_back(next++);
    };
    while (next < count && get_start(next) + element_length <= window_size) admit();      // the first window (:49-72)
    if (!line.empty()) sample(line.front());
    size_t at = 0;                                                // start of the window last dealt with
    while (at + window_size < sequence_length) {
        size_t to = sequence_length - window_size;               // the last window, unless something happens before it
        if (next < count) {
            const size_t end = get_start(next) + element_length;
            if (end < window_size) throw std::logic_error("sample_minimal: elements not sorted by start");      // (crash_unless, :103)
            to = std::min(to, end - window_size);                 // the first window the next element lies in
        }
        if (!line.empty()) to = std::min(to, get_start(line.front()) + 1);      // the first window the front element does not lie in
        // elements that are no longer inside drop out at the front; one that started where the dropped front did was tied with it in the
        // window before: sampled as well (:124-143)
        while (!line.empty() && to > get_start(line.front())) {
            line.pop_front();
            if (!line.empty() && to > get_start(line.front())) sample(line.front());
        }
        while (next < count && to + window_size >= get_start(next) + element_length) admit();
        if (!line.empty()) sample(line.front());
        at = to;
    }
    // the last window's ties: everything that starts where its front does (:183-206)
    if (!line.empty()) {
        const size_t tie = get_start(line.front());
        line.pop_front();
        while (!line.empty() && get_start(line.front()) == tie) { sample(line.front()); line.pop_front(); }
    }
}

void score_minimizers(std::vector<PolicyMinimizer>& ms, size_t hard_hit_cap) {
    const double base = 1.0 + std::log((double)hard_hit_cap);
    for (PolicyMinimizer& m : ms) m.score = !m.hits ? 0.0 : (m.hits <= hard_hit_cap ? base - std::log((double)m.hits) : 1.0);
}

uint32_t ReadRng::operator()() {
    if (!started_) {                                                     // (std::minstd_rand's seeding: the seed modulo 2^31 - 1, 1 when that is 0)
        uint32_t seed = 0;
        for (unsigned char byte : seed_) seed = seed * 13u + byte;
        state_ = seed % 2147483647u; if (!state_) state_ = 1u;
        started_ = true;
    }
    state_ = (uint32_t)(((uint64_t)state_ * 48271ull) % 2147483647ull);
    return state_;
}

static std::vector<size_t> by_score_drawing_from(const std::vector<PolicyMinimizer>& ms, const std::function<uint32_t()>* draw);

std::vector<size_t> minimizers_by_score(const std::vector<PolicyMinimizer>& ms, ReadRng& rng) {
    const std::function<uint32_t()> draw = [&]() { return rng(); };
    return by_score_drawing_from(ms, &draw);
}
std::vector<uint8_t> select_minimizers_in_order(const std::vector<PolicyMinimizer>& ms, size_t read_length, const SeedPolicy& P, const std::vector<size_t>& order);
std::vector<uint8_t> select_minimizers(const std::vector<PolicyMinimizer>& ms, size_t read_length, const SeedPolicy& P, ReadRng& rng) {
    return select_minimizers_in_order(ms, read_length, P, minimizers_by_score(ms, rng));
}

std::vector<size_t> minimizers_by_score(const std::vector<PolicyMinimizer>& ms, const std::string* sequence) {
    if (!sequence) return by_score_drawing_from(ms, nullptr);
    // (the single-end path: the generator std::minstd_rand itself, seeded as LazyRNG seeds it — the statement ReadRng's written-out form is tested against)
    uint32_t seed = 0;
    for (unsigned char byte : *sequence) seed = seed * 13u + byte;
    std::minstd_rand generator(seed);
    const std::function<uint32_t()> draw = [&]() { return (uint32_t)generator(); };
    return by_score_drawing_from(ms, &draw);
}

static std::vector<size_t> by_score_drawing_from(const std::vector<PolicyMinimizer>& ms, const std::function<uint32_t()>* draw) {
    std::vector<size_t> order(ms.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    // a key's occurrences have one hit count and so one score: sorting by (score descending, key, read position) keeps every run together
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        if (ms[a].score != ms[b].score) return ms[a].score > ms[b].score;
        return ms[a].key < ms[b].key;
    });
    if (!draw || order.empty()) return order;
    // sort_shuffling_ties over the runs (src/utility.hpp:771-799): the runs that share the best score are shuffled — deterministic_shuffle
    // (:720-727) with the generator LazyRNG makes from the read's sequence (src/utility.cpp:911-927), which is std::minstd_rand itself
    std::vector<std::pair<size_t, size_t>> runs;                        // [first, end) in `order`, the leading stretch of equal score only
    for (size_t at = 0; at < order.size() && ms[order[at]].score == ms[order[0]].score;) {
        size_t end = at + 1;
        while (end < order.size() && ms[order[end]].key == ms[order[at]].key) ++end;
        runs.emplace_back(at, end); at = end;
    }
    if (runs.size() < 2) return order;
    for (size_t i = 1; i < runs.size(); ++i) std::swap(runs[(*draw)() % (i + 1)], runs[i]);
    std::vector<size_t> laid; laid.reserve(order.size());
    for (const auto& run : runs) laid.insert(laid.end(), order.begin() + (std::ptrdiff_t)run.first, order.begin() + (std::ptrdiff_t)run.second);
    std::copy(laid.begin(), laid.end(), order.begin());
    return order;
}

std::vector<uint8_t> select_minimizers(const std::vector<PolicyMinimizer>& ms, size_t read_length, const SeedPolicy& P, const std::string* sequence) {
    return select_minimizers_in_order(ms, read_length, P, minimizers_by_score(ms, sequence));
}
std::vector<uint8_t> select_minimizers_in_order(const std::vector<PolicyMinimizer>& ms, size_t read_length, const SeedPolicy& P, const std::vector<size_t>& order) {
    const size_t n = ms.size();
    std::vector<uint8_t> verdict(n, SEED_TAKEN);
    const bool score_filter = P.hit_cap != 0 || P.minimizer_score_fraction != 1.0;
    double base_target = 0.0, target = 0.0, selected = 0.0;
    if (score_filter) { for (size_t i : order) base_target += ms[i].score; target = base_target * P.minimizer_score_fraction + 0.000001; }      // (summed in score order, as the reference does: :4120-4125)
    // ---- window downsampling, in read order, a minimizer length at a time (:4178-4238)
    std::vector<char> kept; bool downsampling = false;
    if (P.minimizer_downsampling_window_count != 0 && n) {
        std::map<size_t, std::vector<size_t>> by_length; size_t shortest = (size_t)-1;
        for (size_t i = 0; i < n; ++i) { by_length[ms[i].length].push_back(i); shortest = std::min(shortest, ms[i].length); }
        size_t window = read_length < P.minimizer_downsampling_window_count * shortest ? 0 : read_length / P.minimizer_downsampling_window_count;
        window = std::min(window, P.minimizer_downsampling_max_window_length);
        if (window) {
            kept.assign(n, 0);
            for (auto& group : by_length) {
                if (group.first > window) throw std::runtime_error("find_seeds: a minimizer is longer than the downsampling window");
                const std::vector<size_t>& idx = group.second;
                sample_minimal(idx.size(), group.first, window, read_length,
                    [&](size_t k) { return ms[idx[k]].forward_offset; },
                    [&](size_t a, size_t b) {                     // one that matches the index beats one that does not; among those, the higher score, then the smaller key (:4214-4222)
                        const PolicyMinimizer& x = ms[idx[a]]; const PolicyMinimizer& y = ms[idx[b]];
                        if (!x.hits) return false;
                        if (!y.hits) return true;
                        return x.score > y.score || (x.score == y.score && x.key < y.key);
                    },
                    [&](size_t k) { kept[idx[k]] = 1; downsampling = true; });
            }
        }
    }
    // ---- the filters, over the minimizers in score order, run by run (:4395-4440)
    std::vector<char> covered_by_minimizer(read_length + 1, 0), covered(read_length, 0);
    size_t taken = 0, worst_kept_hits = 0;
    const size_t by_read_length = P.num_bp_per_min ? read_length / P.num_bp_per_min : 0;
    size_t run_end = 0, run_hits = 0; bool taking_run = false;
    for (size_t at = 0; at < n; ++at) {
        if (at >= run_end) {                                       // a new run of one key
            run_end = at + 1; run_hits = ms[order[at]].hits;
            while (run_end < n && ms[order[run_end]].key == ms[order[at]].key) { run_hits += ms[order[run_end]].hits; ++run_end; }
            taking_run = false;
        }
        const size_t i = order[at]; const PolicyMinimizer& m = ms[i];
        uint8_t failed = SEED_TAKEN;
        if (downsampling && !kept[i]) failed = SEED_DOWNSAMPLED;                                    // (:4262-4268; an empty set filters nothing)
        else if (!m.hits) failed = SEED_NO_HITS;
        else if (run_hits > P.hard_hit_cap) failed = SEED_HARD_HIT_CAP;
        if (!failed && P.exclude_overlapping_min) {                                                  // (:4290-4308)
            if (covered_by_minimizer[m.forward_offset] || covered_by_minimizer[std::min(m.forward_offset + m.length, read_length)]) failed = SEED_OVERLAPPING;
            else for (size_t p = m.forward_offset; p < std::min(m.forward_offset + m.length, read_length + 1); ++p) covered_by_minimizer[p] = 1;
        }
        if (!failed && P.max_unique_min != 0) {                                                      // (:4310-4356)
            const size_t lo = m.forward_offset < P.minimizer_coverage_flank ? 0 : m.forward_offset - P.minimizer_coverage_flank;
            const size_t hi = std::min(read_length, m.forward_offset + m.length + P.minimizer_coverage_flank);
            if (taken < std::max(P.max_unique_min, by_read_length)) {
                for (size_t p = lo; p < hi; ++p) covered[p] = 1;
                worst_kept_hits = std::max(worst_kept_hits, m.hits);
            } else if (m.hits > worst_kept_hits) failed = SEED_MAX_MIN;
            else {
                bool fresh = true;
                for (size_t p = lo; p < hi && fresh; ++p) fresh = !covered[p];
                if (!fresh) failed = SEED_MAX_MIN;
                else for (size_t p = lo; p < hi; ++p) covered[p] = 1;
            }
        }
        if (!failed && score_filter) {                                                               // (:4358-4378)
            const bool pass = m.hits <= P.hit_cap || (run_hits <= P.hard_hit_cap && selected + m.score <= target) || taking_run;
            if (pass) selected += m.score;
            else { failed = SEED_HIT_CAP; target = selected; }                                       // once the fraction is reached nothing more is taken for it
        }
        verdict[i] = failed;
        if (!failed) { taking_run = true; ++taken; }
    }
    return verdict;
}

ClusterScore score_cluster(const std::vector<size_t>& seed_sources, const std::vector<PolicyMinimizer>& minimizers, size_t seq_length) {
    ClusterScore c; c.present.assign(minimizers.size(), 0);
    for (size_t source : seed_sources) c.present.at(source) = 1;                                  // (:4746-4748)
    std::vector<uint8_t> covered(seq_length, 0);
    for (size_t j = 0; j < minimizers.size(); ++j) {                                              // (:4755-4768) in read order: the sum's order is the reference's
        if (!c.present[j]) continue;
        c.score += minimizers[j].score;
        for (size_t b = minimizers[j].forward_offset; b < minimizers[j].forward_offset + minimizers[j].length && b < seq_length; ++b) covered[b] = 1;
    }
    size_t ones = 0; for (uint8_t b : covered) ones += b;
    c.coverage = seq_length ? (double)ones / (double)seq_length : 0.0;                             // (:4770)
    return c;
}

}  // namespace vgamd
