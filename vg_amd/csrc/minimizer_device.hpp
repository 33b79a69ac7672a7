// minimizer_device.hpp — minimizer seeding: the step before the extension stage (first half).
// Replaces, for a batch of reads, what MinimizerMapper::find_minimizers / find_seeds get from gbwtgraph's MinimizerIndex
// (src/minimizer_mapper.cpp:3918-3965, :4109-4290: minimizer_regions(sequence), find(minimizer), one seed per hit):
// the (k, w)-minimizers of every read and, for each, the graph positions of that k-mer on the haplotypes.
//
// [PARITY-UNPINNED] gbwtgraph is an un-vendored submodule, absent from the snapshot; this is its published scheme restated:
//   * a k-mer (k <= 31) is its bases packed two bits each (A C G T = 0 1 2 3), first base in the highest bits; a k-mer with any other
//     character is no candidate;
//   * its hash is Thomas Wang's 64-bit integer hash of that key; a k-mer stands for itself and its reverse complement: the
//     orientation with the smaller hash is the canonical one (equal hashes — palindromes — forward);
//   * the minimizer of a window of w consecutive k-mers is the candidate with the smallest canonical hash, the leftmost among equals;
//     a sequence shorter than k + w - 1 bases has none; every (position) is reported once, in read order;
//   * the index holds the minimizers of every haplotype path, keyed by the canonical key, with the graph position where the
//     canonical-orientation k-mer starts (the forward k-mer's first base, or the flipped position of its last base).
// A hit becomes a seed of the extension stage (vgk_seed: oriented node + read offset - node offset) on the strand the READ reads
// forward on: for a forward-canonical minimizer at read offset p the hit itself; for a reverse-canonical one the hit flipped,
// paired with the k-mer's last read base p + k - 1.
//
// The same code runs in the index builder (host), under tests/emu (test infrastructure only) and in the kernels.
#pragma once
#include <stdint.h>
#include "gapless_device.hpp"
#include "../../include/vgk.h"

namespace vgk {

constexpr uint32_t MZ_MAX_K = 31, MZ_MAX_W = 64, MZ_MAX_SEEDS = 64;      // (64 = the seeds a cluster of the extension stage may hold)
VGK_HD uint64_t mz_hash(uint64_t key) {                 // Thomas Wang's 64-bit mix
    key = (~key) + (key << 21); key ^= key >> 24; key = (key + (key << 3)) + (key << 8); key ^= key >> 14;
    key = (key + (key << 2)) + (key << 4); key ^= key >> 28; key += key << 31;
    return key;
}
VGK_HD int mz_code(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; } }

// the k-mers of a sequence, one base at a time: forward and reverse-complement keys, how many valid bases in a row
struct MzRoll {
    uint64_t fwd, rev, mask; uint32_t k, valid;
    VGK_HD void init(uint32_t kk) { k = kk; mask = kk == 32 ? ~0ull : ((1ull << (2 * kk)) - 1ull); fwd = rev = 0; valid = 0; }
    VGK_HD void push(char c) {
        const int x = mz_code(c);
        if (x < 0) { valid = 0; fwd = rev = 0; return; }
        fwd = ((fwd << 2) | (uint64_t)x) & mask;
        rev = (rev >> 2) | ((uint64_t)(3 - x) << (2 * (k - 1)));
        ++valid;
    }
    VGK_HD bool full() const { return valid >= k; }
};
struct MzKmer { uint64_t hash, key; bool reverse; };                     // canonical
VGK_HD MzKmer mz_canonical(const MzRoll& r) {
    const uint64_t hf = mz_hash(r.fwd), hr = mz_hash(r.rev);
    MzKmer m; m.reverse = hr < hf; m.hash = m.reverse ? hr : hf; m.key = m.reverse ? r.rev : r.fwd;
    return m;
}

// The minimizers of seq[0, L): calls out(start offset of the k-mer, canonical k-mer) for each, in order of offset.
// A ring of the last w candidates (hash, or "none"); the current minimum is tracked and re-found when it leaves the window.
template <class OUT>
VGK_HD void mz_minimizers(const char* seq, uint32_t L, uint32_t k, uint32_t w, OUT out) {
    if (L < k + w - 1 || k == 0 || k > MZ_MAX_K || w == 0 || w > MZ_MAX_W) return;
    constexpr uint64_t NONE = ~0ull;
    uint64_t ring_hash[MZ_MAX_W], ring_key[MZ_MAX_W]; uint8_t ring_rev[MZ_MAX_W];
    MzRoll roll; roll.init(k);
    for (uint32_t i = 0; i + 1 < k; ++i) roll.push(seq[i]);
    const uint32_t n_kmers = L - k + 1;
    int64_t best = -1;                                                    // k-mer index of the current window's minimizer
    int64_t reported = -1;
    for (uint32_t j = 0; j < n_kmers; ++j) {
        roll.push(seq[j + k - 1]);
        const uint32_t slot = j % w;
        if (roll.full()) { const MzKmer m = mz_canonical(roll); ring_hash[slot] = m.hash; ring_key[slot] = m.key; ring_rev[slot] = m.reverse ? 1 : 0; }
        else { ring_hash[slot] = NONE; ring_key[slot] = 0; ring_rev[slot] = 0; }
        // window = k-mers [j - w + 1, j]
        if (best >= 0 && best + (int64_t)w <= (int64_t)j) best = -1;       // the minimum slid out
        if (best < 0) {
            const uint32_t lo = j + 1 >= w ? j + 1 - w : 0;
            for (uint32_t i = lo; i <= j; ++i) { const uint64_t h = ring_hash[i % w]; if (h != NONE && (best < 0 || h < ring_hash[best % w])) best = i; }
        } else if (ring_hash[slot] != NONE && ring_hash[slot] < ring_hash[best % w]) best = j;
        if (j + 1 >= w && best >= 0 && best != reported) {
            MzKmer m; m.hash = ring_hash[best % w]; m.key = ring_key[best % w]; m.reverse = ring_rev[best % w] != 0;
            out((uint32_t)best, m);
            reported = best;
        }
    }
}

// The same for the windows [wa, wb) of the sequence only (window s = k-mers [s, s + w)): what mz_minimizers reports while it is at those windows —
// a minimizer that window wa - 1 already had is not reported again.  The best of a window is its leftmost smallest candidate whichever window the
// scan started at, so a long read cut into stretches of windows, one lane each, lists exactly the minimizers of the whole read, stretch by stretch.
template <class OUT>
VGK_HD void mz_minimizers_range(const char* seq, uint32_t L, uint32_t k, uint32_t w, uint32_t wa, uint32_t wb, OUT out) {
    if (L < k + w - 1 || k == 0 || k > MZ_MAX_K || w == 0 || w > MZ_MAX_W) return;
    constexpr uint64_t NONE = ~0ull;
    const uint32_t n_windows = L - k - w + 2;
    if (wb > n_windows) wb = n_windows;
    if (wa >= wb) return;
    uint64_t ring_hash[MZ_MAX_W], ring_key[MZ_MAX_W]; uint8_t ring_rev[MZ_MAX_W];
    const uint32_t j0 = wa ? wa - 1 : 0;                                  // first k-mer looked at: window wa - 1's first
    MzRoll roll; roll.init(k);
    for (uint32_t i = 0; i + 1 < k; ++i) roll.push(seq[j0 + i]);
    int64_t best = -1, reported = -1;
    for (uint32_t j = j0; j < wb + w - 1; ++j) {
        roll.push(seq[j + k - 1]);
        const uint32_t slot = j % w;
        if (roll.full()) { const MzKmer m = mz_canonical(roll); ring_hash[slot] = m.hash; ring_key[slot] = m.key; ring_rev[slot] = m.reverse ? 1 : 0; }
        else { ring_hash[slot] = NONE; ring_key[slot] = 0; ring_rev[slot] = 0; }
        if (j + 1 < j0 + w) continue;                                     // the first window is not complete yet
        const uint32_t sidx = j + 1 - w;                                  // the window that ends at k-mer j
        if (best >= 0 && best < (int64_t)sidx) best = -1;                 // the minimum slid out
        if (best < 0) { for (uint32_t i = sidx; i <= j; ++i) { const uint64_t h = ring_hash[i % w]; if (h != NONE && (best < 0 || h < ring_hash[best % w])) best = i; } }
        else if (ring_hash[slot] != NONE && ring_hash[slot] < ring_hash[best % w]) best = j;
        if (best >= 0 && best != reported) {
            if (sidx >= wa) { MzKmer m; m.hash = ring_hash[best % w]; m.key = ring_key[best % w]; m.reverse = ring_rev[best % w] != 0; out((uint32_t)best, m); }
            reported = best;
        }
    }
}

// ---- the index on the device: open addressing, linear probing --------------------------------------------------------------------
// A key with ONE position — nearly every key of a genome-scale index — holds it in its slot (count = MZ_INLINE | offset word, first =
// node): a lookup is then one 16-byte request instead of two dependent ones.  A position's offset word carries the offset in its low
// 16 bits and the offset seen from the node's other end (length - 1 - offset) above them, so that a reverse-canonical hit is flipped
// without a look at the node's length (nodes are at most 65535 bases long, as everywhere in the engine).
struct MzSlot { uint64_t key; uint32_t first, count; };                  // count == 0: free
struct MzPos { uint32_t node, offset; };                                 // offset: offset | flipped offset << 16
constexpr uint32_t MZ_INLINE = 0x80000000u;                              // in a slot's count: the slot holds the key's one position (only for offset words below 2^31: nodes of less than 32768 bases)
constexpr uint32_t MZ_ONE = 0xffffffffu;                                 // mz_find's `first` for such a key: the position is in `one`
struct MzIndex { const MzSlot* slots; uint32_t mask; const MzPos* pos; uint32_t k, w; };
VGK_HD uint32_t mz_pack_offset(uint32_t offset, uint32_t node_len) { return offset | ((node_len - 1u - offset) << 16); }
// the seed a hit gives for a minimizer at read offset p (minimizer_device.hpp header: the hit itself, or flipped for a reverse-canonical one)
VGK_HD vgk_seed mz_seed(const MzPos q, uint32_t p, bool reverse, uint32_t k) {
    vgk_seed s;
    if (!reverse) { s.node = q.node; s.diff = (int32_t)p - (int32_t)(q.offset & 0xffffu); }
    else { s.node = q.node ^ 1u; s.diff = (int32_t)(p + k - 1) - (int32_t)(q.offset >> 16); }
    return s;
}
// -> count hits at pos[first ..), or (first == MZ_ONE) the key's one position in `one`
VGK_HD bool mz_find(const MzIndex& x, const MzKmer& m, uint32_t& first, uint32_t& count, MzPos& one) {
    for (uint32_t s = (uint32_t)m.hash & x.mask;; s = (s + 1) & x.mask) {
        const MzSlot e = x.slots[s];
        if (!e.count) return false;
        if (e.key == m.key) {
            if (e.count & MZ_INLINE) { first = MZ_ONE; count = 1; one.node = e.first; one.offset = e.count & ~MZ_INLINE; }
            else { first = e.first; count = e.count; }
            return true;
        }
    }
}

// ---- find_seeds' choice of minimizers (include/vgk.h: vgk_seed_policy; reference src/minimizer_mapper.cpp:4109-4440) ---------------------
// `tab[h]` = the score of a minimizer with h hits, h = 0 .. hard_hit_cap, computed once on the host (1 + ln(hard) - ln(h): the device's log
// need not round like the host's); everything here is additions and comparisons of those doubles in one fixed order, with the one
// product kept apart from its sum (mz_mul / mz_add: no fused multiply-add), so that every backend selects the same minimizers.
constexpr uint32_t MZ_POLICY_MAX = 64;                                   // minimizers per read the selection takes
struct MzPolicy { uint32_t on, hit_cap, hard_hit_cap; double fraction; const double* tab; uint32_t paired; };      // paired: include/vgk.h vgk_seed_policy
VGK_HD double mz_add(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(a, b);
#else
    volatile double r = a + b; return r;
#endif
}
VGK_HD double mz_mul(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dmul_rn(a, b);
#else
    volatile double r = a * b; return r;
#endif
}
VGK_HD double mz_score(const MzPolicy& Q, uint32_t hits) { return !hits ? 0.0 : (hits <= Q.hard_hit_cap ? Q.tab[hits] : 1.0); }
VGK_HD bool mz_better(double sa, uint64_t ka, uint32_t ia, double sb, uint64_t kb, uint32_t ib) {      // a before b in the order the filters run in
    return sa > sb || (sa == sb && (ka < kb || (ka == kb && ia < ib)));
}
// ---- the ties at the top (sort_minimizers_by_score, :4074-4107 -> sort_shuffling_ties, src/utility.hpp:771-799) ----
// The runs are stable-sorted by descending score — which leaves them in the order above — and then the runs whose score EQUALS THE BEST ONE'S are
// shuffled: Knuth's shuffle as deterministic_shuffle writes it (:720-727: for i = 1 .. width - 1: swap(x[rng() % (i + 1)], x[i])) over a
// std::minstd_rand (x <- 48271 x mod (2^31 - 1); first state = seed mod (2^31 - 1), or 1 when that is 0) that LazyRNG seeds from the read's
// sequence: seed = seed * 13 + byte over its bytes, in 32 bits (src/utility.cpp:911-927; single-end: the read's own sequence, a fresh generator per
// read, src/minimizer_mapper.cpp:620-627).  Ties further down keep the order above.  [The PAIRED path seeds ONE generator from both mates' sequences
// and carries its state from the first mate's sort to the second's (:1529-1541): not restated — a pair's reads are shuffled as single reads here.]
// The engine keeps reads masked (anything but ACGT became 'X'): a read with such a base whose shuffle could change the choice is not chosen for
// here but flagged VGK_MINIMIZERS_POLICY_SKIPPED, like a read with more than 64 minimizers — the host shim (seed_policy.cpp) has its real bytes.
VGK_HD uint32_t mz_shuffle_seed(const char* seq, uint32_t L, bool& masked) {
    uint32_t s = 0; masked = false;
    for (uint32_t i = 0; i < L; ++i) { const char c = seq[i]; masked = masked || !(c == 'A' || c == 'C' || c == 'G' || c == 'T'); s = s * 13u + (uint32_t)(uint8_t)c; }
    return s;
}
// the leading runs of `order` whose score equals the first one's -> their number; *elements = the minimizers in them
template <class IDX, class KEY> VGK_HD uint32_t mz_top_ties(const IDX* order, const KEY* key, const double* score, uint32_t n, uint32_t* elements) {
    uint32_t runs = 0, r = 0;
    if (n) { const double s0 = score[order[0]];
             while (r < n && score[order[r]] == s0) { ++runs; const KEY k = key[order[r]]; while (r < n && key[order[r]] == k) ++r; } }
    *elements = r;
    return runs;
}
// shuffles those runs in place; start / perm / tmp: room for MZ_POLICY_MAX + 1 entries each (LDS in the kernel, the stack elsewhere)
template <class IDX, class KEY> VGK_HD void mz_shuffle_top_ties(IDX* order, const KEY* key, uint32_t elements, uint32_t runs, uint32_t seed, uint8_t* start, uint8_t* perm, IDX* tmp) {
    uint32_t r = 0, q = 0;
    while (r < elements) { start[q] = (uint8_t)r; perm[q] = (uint8_t)q; ++q; const KEY k = key[order[r]]; while (r < elements && key[order[r]] == k) ++r; }
    start[q] = (uint8_t)elements;
    uint32_t x = seed % 2147483647u; if (!x) x = 1u;
    for (uint32_t i = 1; i < runs; ++i) {
        x = (uint32_t)(((uint64_t)x * 48271ull) % 2147483647ull);
        const uint32_t j = x % (i + 1u);
        const uint8_t t = perm[j]; perm[j] = perm[i]; perm[i] = t;
    }
    uint32_t at = 0;
    for (uint32_t i = 0; i < runs; ++i) for (uint32_t e = start[perm[i]]; e < start[perm[i] + 1u]; ++e) tmp[at++] = order[e];
    for (uint32_t e = 0; e < elements; ++e) order[e] = tmp[e];
}
// whether the order inside the top tie can change the choice: only when its runs are held against the score fraction (more hits than hit_cap)
VGK_HD bool mz_tie_matters(const MzPolicy& Q, uint32_t runs, uint32_t top_hits) { return runs >= 2u && (Q.hit_cap != 0 || Q.fraction != 1.0) && top_hits > Q.hit_cap; }

// n <= MZ_POLICY_MAX minimizers of a read in read order (key, hits) -> bit i set: minimizer i gives seeds.  The serial form (one lane; the
// emulator, the checker of the wave-parallel form in backend_hip.hip's minimizer_kernel).  seq / L: the read as the engine holds it (masked);
// *unsure: the read has a masked base and its top tie matters — not chosen for (the result is to be ignored).
VGK_HD uint64_t mz_policy_select(const MzPolicy& Q, const uint64_t* key, const uint32_t* hits, uint32_t n, const char* seq, uint32_t L, bool* unsure) {
    uint32_t order[MZ_POLICY_MAX]; double score[MZ_POLICY_MAX];
    for (uint32_t i = 0; i < n; ++i) score[i] = mz_score(Q, hits[i]);
    for (uint32_t i = 0; i < n; ++i) { uint32_t rank = 0; for (uint32_t j = 0; j < n; ++j) rank += mz_better(score[j], key[j], j, score[i], key[i], i) ? 1u : 0u; order[rank] = i; }
    *unsure = false;
    { uint32_t elements = 0; const uint32_t runs = mz_top_ties(order, key, score, n, &elements);
      if (mz_tie_matters(Q, runs, n ? hits[order[0]] : 0u)) {                  // (tied runs of at most hit_cap hits are all taken whatever their order: no generator is made for them)
          bool masked = false; const uint32_t seed = mz_shuffle_seed(seq, L, masked);
          if (masked || Q.paired) { *unsure = true; return 0; }                  // (a pair's one generator over both mates is the caller's: include/vgk.h)
          uint8_t start[MZ_POLICY_MAX + 1], perm[MZ_POLICY_MAX + 1]; uint32_t tmp[MZ_POLICY_MAX + 1];
          mz_shuffle_top_ties(order, key, elements, runs, seed, start, perm, tmp);
      } }
    const bool use_score = Q.hit_cap != 0 || Q.fraction != 1.0;
    double target = 0.0, selected = 0.0;
    if (use_score) { double base = 0.0; for (uint32_t r = 0; r < n; ++r) base = mz_add(base, score[order[r]]); target = mz_add(mz_mul(base, Q.fraction), 0.000001); }
    uint64_t mask = 0; bool taking = false;
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t i = order[r];
        if (r == 0 || key[order[r - 1]] != key[i]) taking = false;           // a new run of one key
        uint32_t run = 0; for (uint32_t j = 0; j < n; ++j) if (key[j] == key[i]) run += hits[j];
        bool pass = hits[i] != 0 && run <= Q.hard_hit_cap;
        if (pass && use_score) {
            if (hits[i] <= Q.hit_cap || mz_add(selected, score[i]) <= target || taking) selected = mz_add(selected, score[i]);
            else { pass = false; target = selected; }
        }
        if (pass) { mask |= 1ull << i; taking = true; }
    }
    return mask;
}

struct MinimizerParams {
    MzIndex index; GIndex graph;                                          // graph: node lengths (for the flip of reverse hits)
    MzPolicy policy;                                                      // on == 0: none
    const char* reads; const uint64_t* read_off; uint32_t n;
    uint32_t hit_cap;                                                      // minimizers with more hits give no seeds (hard_hit_cap, :4180)
    uint32_t* counts;                                                      // pass 1: seeds per read ([n + 1], last 0); minimizers per read in mins[]
    uint32_t* mins;
    const uint32_t* first;                                                 // pass 2: exclusive prefix sums of counts
    vgk_seed* seeds; int pass;
    uint32_t lo, hi;                                                       // pass 1 over reads [lo, hi): the reads go up in slices, a slice's kernel runs under the next slice's copy
};
// one read: count its seeds (pass 1) or write them (pass 2), in order of the minimizers' read offsets, a minimizer's hits in index order.
// A cluster is a SET of seeds (GaplessExtender::cluster_type is a hash set, src/gbwt_extender.hpp:143): a (node, diagonal) pair that
// a second minimizer of the read hits again is reported once; seeds beyond MZ_MAX_SEEDS are dropped and the read is flagged
// (VGK_MINIMIZERS_TRUNCATED in mins[]: the cap was reached with hits left unexamined).
VGK_HD void minimizer_one(const MinimizerParams& P, uint32_t i) {
    const uint64_t a = P.read_off[i]; const uint32_t L = (uint32_t)(P.read_off[i + 1] - a);
    uint32_t n_seeds = 0, n_min = 0; bool truncated = false;
    vgk_seed* dst = P.pass == 2 ? P.seeds + P.first[i] : nullptr;
    const uint32_t k = P.index.k;
    uint64_t seen[MZ_MAX_SEEDS];
    // with a policy: the read's minimizers first (key, hits), then the choice, then the seeds of the chosen ones
    uint64_t chosen = ~0ull; bool skipped = false;
    if (P.policy.on) {
        uint64_t pkey[MZ_POLICY_MAX]; uint32_t phits[MZ_POLICY_MAX]; uint32_t np = 0;
        mz_minimizers(P.reads + a, L, k, P.index.w, [&](uint32_t, const MzKmer& m) {
            uint32_t first = 0, count = 0; MzPos one{0, 0};
            const bool found = mz_find(P.index, m, first, count, one);
            if (np < MZ_POLICY_MAX) { pkey[np] = m.key; phits[np] = found ? count : 0u; }
            ++np;
        });
        if (np > MZ_POLICY_MAX) skipped = true;
        else { bool unsure = false; chosen = mz_policy_select(P.policy, pkey, phits, np, P.reads + a, L, &unsure); if (unsure) { skipped = true; chosen = ~0ull; } }
    }
    const uint32_t cap = P.policy.on && !skipped ? 0xffffffffu : P.hit_cap;      // (the run's hits were held against the hard cap by the choice)
    mz_minimizers(P.reads + a, L, k, P.index.w, [&](uint32_t p, const MzKmer& m) {
        const uint32_t ordinal = n_min++;
        uint32_t first = 0, count = 0; MzPos one{0, 0};
        if (!mz_find(P.index, m, first, count, one) || count > cap) return;
        if (ordinal < 64u && !((chosen >> ordinal) & 1ull)) return;
        for (uint32_t h = 0; h < count; ++h) {
            if (n_seeds >= MZ_MAX_SEEDS) { truncated = true; break; }      // the cap: this hit and the rest are never looked at (reported in mins[])
            const MzPos q = first == MZ_ONE ? one : P.index.pos[first + h];
            const vgk_seed s = mz_seed(q, p, m.reverse, k);
            const uint64_t key = ((uint64_t)s.node << 32) | (uint32_t)s.diff;
            bool dup = false;
            for (uint32_t j = 0; j < n_seeds && !dup; ++j) dup = seen[j] == key;
            if (dup) continue;
            seen[n_seeds] = key;
            if (dst) dst[n_seeds] = s;
            ++n_seeds;
        }
    });
    if (P.pass == 1) { P.counts[i] = n_seeds; if (P.mins) P.mins[i] = n_min | (truncated ? VGK_MINIMIZERS_TRUNCATED : 0u) | (skipped ? VGK_MINIMIZERS_POLICY_SKIPPED : 0u); }
}

// ---- reads of any length (vgk_minimizer_list / vgk_minimizer_seeds_of): no caps, the choice between the two calls is the caller's ----------
// List: a lane per read, pass 1 counts the read's minimizers, pass 2 writes them behind the reads before it.  Seeds: a lane per minimizer,
// pass 1 = its hits if it is taken, pass 2 = one seed per hit, index order.
constexpr uint32_t MZ_LIST_WINDOWS = 192;     // windows per work item: a 15 kbp read is 78 items
struct MzListItem { uint32_t read, window; };      // windows [window, window + MZ_LIST_WINDOWS) of read `read`
struct MzListParams {
    MzIndex index; const char* reads; const uint64_t* read_off; uint32_t n;      // n: ITEMS (a read is cut into stretches of windows, a lane each)
    const MzListItem* items;
    uint32_t* counts;                     // [n + 1] pass 1 (entry n = 0)
    const uint32_t* first;                // their exclusive prefix sums
    vgk_read_minimizer* out; int pass;
};
VGK_HD void mz_list_one(const MzListParams& P, uint32_t i) {
    if (i >= P.n) { if (P.pass == 1 && i == P.n) P.counts[i] = 0; return; }
    const MzListItem it = P.items[i];
    const uint64_t a = P.read_off[it.read]; const uint32_t L = (uint32_t)(P.read_off[it.read + 1] - a);
    uint32_t n_min = 0;
    vgk_read_minimizer* dst = P.pass == 2 ? P.out + P.first[i] : nullptr;
    mz_minimizers_range(P.reads + a, L, P.index.k, P.index.w, it.window, it.window + MZ_LIST_WINDOWS, [&](uint32_t p, const MzKmer& m) {
        if (dst) {
            uint32_t first = 0, count = 0; MzPos one{0, 0};
            const bool found = mz_find(P.index, m, first, count, one);
            vgk_read_minimizer r; r.key = m.key; r.offset = p; r.hits = found ? count : 0u; r.flags = m.reverse ? VGK_MINIMIZER_REVERSE : 0u; r.reserved = 0;
            dst[n_min] = r;
        }
        ++n_min;
    });
    if (P.pass == 1) P.counts[i] = n_min;
}
struct MzSeedsOfParams {
    MzIndex index; const vgk_read_minimizer* mins; const uint8_t* take; uint32_t n;
    uint32_t* counts; const uint32_t* first; vgk_seed* out; int pass;
};
VGK_HD void mz_seeds_of_one(const MzSeedsOfParams& P, uint32_t j) {
    if (j >= P.n) { if (P.pass == 1 && j == P.n) P.counts[j] = 0; return; }
    const vgk_read_minimizer r = P.mins[j];
    uint32_t first = 0, count = 0; MzPos one{0, 0};
    MzKmer m; m.key = r.key; m.hash = mz_hash(r.key); m.reverse = (r.flags & VGK_MINIMIZER_REVERSE) != 0;
    const bool found = P.take[j] != 0 && mz_find(P.index, m, first, count, one);
    if (P.pass == 1) { P.counts[j] = found ? count : 0u; return; }
    if (!found) return;
    vgk_seed* dst = P.out + P.first[j];
    for (uint32_t h = 0; h < count; ++h) dst[h] = mz_seed(first == MZ_ONE ? one : P.index.pos[first + h], r.offset, m.reverse, P.index.k);
}

}  // namespace vgk
