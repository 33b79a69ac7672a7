"""Which minimizers become seeds (vg_amd/host/seed_policy.cpp): algorithms::sample_minimal held to the reference's six unit tests
(tests/golden/ref_sample_minimal.json, src/unittest/sample_minimal.cpp:14-176) and to a window-by-window restatement on random inputs;
MinimizerMapper::find_seeds' filter chain [PARITY-UNPINNED: the reference holds no test for it] held to a direct restatement of
src/minimizer_mapper.cpp:4109-4440 written out below (checker-side, small cases)."""
import ctypes
import json
import math
import os

import numpy as np
import pytest

import util

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sample_minimal.json")


def sample_minimal(starts, goodness, element_length, window_size, sequence_length):
    h = util.host()
    h.vgh_sample_minimal.argtypes = [ctypes.c_uint64] * 4 + [ctypes.c_void_p] * 3
    s = np.ascontiguousarray(starts, dtype=np.uint64); g = np.ascontiguousarray(goodness, dtype=np.int64); out = np.zeros(max(len(s), 1), dtype=np.uint8)
    assert h.vgh_sample_minimal(len(s), element_length, window_size, sequence_length, s.ctypes.data, g.ctypes.data, out.ctypes.data) == 0
    return set(int(i) for i in np.flatnonzero(out[:len(s)]))


def sample_minimal_by_windows(starts, goodness, element_length, window_size, sequence_length):
    """src/algorithms/sample_minimal.cpp:21-207 one window at a time (the reference jumps between the windows where something changes; in
    the windows between, the same front element would be sampled again): the queue keeps the window's elements no later one has beaten;
    per window its front is sampled; an element that leaves together with the front it stood behind — same start — is sampled too."""
    sampled = set(); queue = []; nxt = 0; n = len(starts)
    beat = lambda a, b: goodness[a] > goodness[b]

    def admit(w):
        nonlocal nxt
        while nxt < n and starts[nxt] + element_length <= w + window_size:
            while queue and beat(nxt, queue[-1]):
                queue.pop()
            queue.append(nxt); nxt += 1
    if n == 0:
        return sampled
    admit(0)
    if queue:
        sampled.add(queue[0])
    last = max(sequence_length - window_size, 0)
    for w in range(1, last + 1):
        while queue and starts[queue[0]] < w:
            queue.pop(0)
            if queue and starts[queue[0]] < w:
                sampled.add(queue[0])
        admit(w)
        if queue:
            sampled.add(queue[0])
    if queue:
        tie = starts[queue[0]]; queue.pop(0)
        while queue and starts[queue[0]] == tie:
            sampled.add(queue.pop(0))
    return sampled


def test_sample_minimal_passes_the_reference_unit_tests():
    cases = json.load(open(GOLDEN))
    assert len(cases) == 6
    for c in cases:
        for got in (sample_minimal(c["starts"], c["goodness"], c["element_length"], c["window_size"], c["sequence_length"]),
                    sample_minimal_by_windows(c["starts"], c["goodness"], c["element_length"], c["window_size"], c["sequence_length"])):
            assert len(got) == c["sampled_count"], c["source"]
            assert all(i in got for i in c["sampled_contains"]), c["source"]


def test_sample_minimal_equals_the_window_by_window_form():
    rng = np.random.default_rng(5)
    for _ in range(400):
        seq_len = int(rng.integers(20, 200)); k = int(rng.integers(3, 15)); window = int(rng.integers(k, 60))
        n = int(rng.integers(0, 40))
        starts = np.sort(rng.integers(0, max(seq_len - k + 1, 1), n)).tolist()                      # several elements may share a start, some starts have none
        goodness = rng.integers(0, int(rng.integers(1, 6)), n).tolist()                              # many ties
        assert sample_minimal(starts, goodness, k, window, seq_len) == sample_minimal_by_windows(starts, goodness, k, window, seq_len), (starts, goodness, k, window, seq_len)


# ---- find_seeds: the restatement ------------------------------------------------------------------------------------------------------
DEFAULTS = dict(hit_cap=10, hard_hit_cap=500, score_fraction=0.9, max_unique_min=500, num_bp_per_min=1000, exclude_overlapping_min=False,
                coverage_flank=250, window_count=0, max_window_length=2 ** 62)


def scores_of(ms, hard_hit_cap):
    base = 1.0 + math.log(hard_hit_cap)                                                              # (:3927-3937)
    return [0.0 if not hits else (base - math.log(hits) if hits <= hard_hit_cap else 1.0) for (_, _, _, hits) in ms]


def shuffle_top_ties(order, ms, score, sequence):
    """sort_shuffling_ties over the runs of one key (src/utility.hpp:771-799, :720-727; src/utility.cpp:911-927): the runs whose score equals the best one's,
    Knuth-shuffled with std::minstd_rand (x <- 48271 x mod 2^31 - 1) seeded by seed * 13 + byte over the read's bytes"""
    if not order:
        return order
    runs = []; at = 0
    while at < len(order) and score[order[at]] == score[order[0]]:
        end = at + 1
        while end < len(order) and ms[order[end]][0] == ms[order[at]][0]:
            end += 1
        runs.append(order[at:end]); at = end
    if len(runs) < 2:
        return order
    seed = 0
    for byte in sequence.encode():
        seed = (seed * 13 + byte) & 0xffffffff
    x = seed % 2147483647 or 1
    for i in range(1, len(runs)):
        x = x * 48271 % 2147483647
        j = x % (i + 1)
        runs[j], runs[i] = runs[i], runs[j]
    return [i for run in runs for i in run] + order[at:]


def find_seeds_restated(ms, read_len, P, sequence=None):
    """ms: (key, forward offset, length, hits) in read order -> the filter each minimizer failed (0: taken)"""
    n = len(ms); score = scores_of(ms, P["hard_hit_cap"])
    order = sorted(range(n), key=lambda i: (-score[i], ms[i][0], i))                                 # (:4074-4107; equal scores by key; inside a run the order cannot matter)
    if sequence is not None:
        order = shuffle_top_ties(order, ms, score, sequence)
    use_score = P["hit_cap"] != 0 or P["score_fraction"] != 1.0
    base_target = 0.0
    for i in order:
        base_target += score[i]
    target = base_target * P["score_fraction"] + 0.000001 if use_score else 0.0
    selected = 0.0
    kept = set()
    if P["window_count"] and n:                                                                      # (:4178-4238)
        shortest = min(m[2] for m in ms)
        window = 0 if read_len < P["window_count"] * shortest else read_len // P["window_count"]
        window = min(window, P["max_window_length"])
        if window:
            for length in sorted(set(m[2] for m in ms)):
                idx = [i for i in range(n) if ms[i][2] == length]
                # the comparator as a rank: no hits lowest; then score descending, key ascending (equal = tie)
                keyed = sorted(set((0, 0.0, 0) if not ms[i][3] else (1, score[i], -ms[i][0]) for i in idx))
                good = [keyed.index((0, 0.0, 0) if not ms[i][3] else (1, score[i], -ms[i][0])) for i in idx]
                for k in sample_minimal_by_windows([ms[i][1] for i in idx], good, length, window, read_len):
                    kept.add(idx[k])
    verdict = [0] * n
    bits = [False] * (read_len + 1); covered = [False] * read_len
    taken = 0; worst = 0; by_len = read_len // P["num_bp_per_min"] if P["num_bp_per_min"] else 0
    at = 0
    while at < n:
        end = at + 1
        while end < n and ms[order[end]][0] == ms[order[at]][0]:
            end += 1
        run_hits = sum(ms[order[j]][3] for j in range(at, end)); taking = False
        for j in range(at, end):
            i = order[j]; key, off, length, hits = ms[i]
            failed = 0
            if kept and i not in kept:
                failed = 1
            elif not hits:
                failed = 2
            elif run_hits > P["hard_hit_cap"]:
                failed = 3
            if not failed and P["exclude_overlapping_min"]:                                          # (:4290-4308)
                if bits[off] or bits[min(off + length, read_len)]:
                    failed = 4
                else:
                    for p in range(off, min(off + length, read_len + 1)):
                        bits[p] = True
            if not failed and P["max_unique_min"]:                                                   # (:4310-4356)
                lo = 0 if off < P["coverage_flank"] else off - P["coverage_flank"]; hi = min(read_len, off + length + P["coverage_flank"])
                if taken < max(P["max_unique_min"], by_len):
                    for p in range(lo, hi):
                        covered[p] = True
                    worst = max(worst, hits)
                elif hits > worst:
                    failed = 5
                elif any(covered[lo:hi]):
                    failed = 5
                else:
                    for p in range(lo, hi):
                        covered[p] = True
            if not failed and use_score:                                                             # (:4358-4378)
                if hits <= P["hit_cap"] or (run_hits <= P["hard_hit_cap"] and selected + score[i] <= target) or taking:
                    selected += score[i]
                else:
                    failed = 6; target = selected
            verdict[i] = failed
            if not failed:
                taking = True; taken += 1
        at = end
    return verdict, score


def select(ms, read_len, P, sequence=None):
    """the host shim's select_minimizers; sequence: the read (its bytes seed the shuffle of the runs tied at the top)"""
    h = util.host()
    h.vgh_select_minimizers.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    h.vgh_select_minimizers_of_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    a = np.ascontiguousarray([x for m in ms for x in m], dtype=np.uint64) if ms else np.zeros(4, dtype=np.uint64)
    pol = np.array([P["hit_cap"], P["hard_hit_cap"], P["max_unique_min"], P["num_bp_per_min"], int(P["exclude_overlapping_min"]), P["coverage_flank"], P["window_count"],
                    P["max_window_length"]], dtype=np.uint64)
    v = np.zeros(max(len(ms), 1), dtype=np.uint8); sc = np.zeros(max(len(ms), 1), dtype=np.float64)
    if sequence is None:
        rc = h.vgh_select_minimizers(a.ctypes.data, len(ms), read_len, pol.ctypes.data, P["score_fraction"], v.ctypes.data, sc.ctypes.data)
    else:
        assert len(sequence) == read_len
        rc = h.vgh_select_minimizers_of_read(a.ctypes.data, len(ms), sequence.encode(), read_len, pol.ctypes.data, P["score_fraction"], v.ctypes.data, sc.ctypes.data, None)
    if rc != 0:
        raise RuntimeError("vgh_select_minimizers")
    return v[:len(ms)].tolist(), sc[:len(ms)].tolist()


def random_minimizers(rng, read_len, k, n_keys):
    """minimizers of a read in read order: some keys occur several times (runs), hit counts from none to far beyond the hard cap"""
    n = int(rng.integers(0, 40))
    offs = np.sort(rng.choice(np.arange(read_len - k + 1), size=min(n, read_len - k + 1), replace=False))
    hits_of = {key: int(rng.choice([0, 1, 2, 3, 9, 10, 11, 40, 499, 500, 501, 3000])) for key in range(n_keys)}
    ms = []
    for o in offs:
        key = int(rng.integers(0, n_keys))
        ms.append((key * 7919 + 13, int(o), k, hits_of[key]))
    return ms


def test_minimizer_scores():
    ms = [(1, 0, 29, 0), (2, 5, 29, 1), (3, 9, 29, 500), (4, 12, 29, 501)]
    _, sc = select(ms, 150, DEFAULTS)
    assert sc[0] == 0.0 and sc[1] == 1.0 + math.log(500) and abs(sc[2] - 1.0) < 1e-12 and sc[3] == 1.0


def test_selection_equals_the_restatement_with_giraffe_defaults():
    rng = np.random.default_rng(11)
    for _ in range(300):
        ms = random_minimizers(rng, 150, 29, int(rng.integers(1, 30)))
        got, sc = select(ms, 150, DEFAULTS); want, wsc = find_seeds_restated(ms, 150, DEFAULTS)
        assert sc == wsc and got == want, ms


def test_selection_equals_the_restatement_with_every_filter_on():
    rng = np.random.default_rng(12)
    for _ in range(600):
        read_len = int(rng.integers(60, 400)); k = int(rng.integers(11, 31))
        P = dict(DEFAULTS, hit_cap=int(rng.choice([0, 1, 3, 10])), hard_hit_cap=int(rng.choice([20, 500])), score_fraction=float(rng.choice([0.5, 0.9, 1.0])),
                 max_unique_min=int(rng.choice([0, 2, 5, 500])), num_bp_per_min=int(rng.choice([50, 1000])), exclude_overlapping_min=bool(rng.integers(0, 2)),
                 coverage_flank=int(rng.choice([0, 10, 250])), window_count=int(rng.choice([0, 0, 2, 4])), max_window_length=int(rng.choice([40, 2 ** 62])))
        ms = random_minimizers(rng, read_len, k, int(rng.integers(1, 25)))
        if P["window_count"] and ms:
            w = 0 if read_len < P["window_count"] * k else min(read_len // P["window_count"], P["max_window_length"])
            if w and k > w:
                with pytest.raises(RuntimeError):                    # crash_unless(length <= window) (:4203)
                    select(ms, read_len, P)
                continue
        got, _ = select(ms, read_len, P); want, _ = find_seeds_restated(ms, read_len, P)
        assert got == want, (ms, read_len, P)


def test_top_ties_are_shuffled_as_the_reference_shuffles_them():
    """the runs that share the best score, shuffled by the read's own generator (sort_shuffling_ties): the host shim (std::minstd_rand itself) against
    the restatement above (the Lehmer formula written out) — order and choice; and the shuffle changes the choice where the cut falls inside the tie"""
    rng = np.random.default_rng(13)
    h = util.host()
    changed = shuffled = 0
    for _ in range(400):
        read_len = int(rng.integers(60, 300)); k = int(rng.integers(11, 31))
        seq = "".join("ACGTN"[int(x)] for x in rng.choice(5, read_len, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        P = dict(DEFAULTS, hit_cap=int(rng.choice([0, 0, 1, 10])), hard_hit_cap=int(rng.choice([20, 500])), score_fraction=float(rng.choice([0.3, 0.6, 0.9])))
        ms = random_minimizers(rng, read_len, k, int(rng.integers(2, 25)))
        got, _ = select(ms, read_len, P, seq); want, score = find_seeds_restated(ms, read_len, P, seq)
        assert got == want, (ms, seq, P)
        # the order itself
        n = len(ms)
        if n:
            a = np.ascontiguousarray([x for m in ms for x in m], dtype=np.uint64)
            pol = np.array([P["hit_cap"], P["hard_hit_cap"], P["max_unique_min"], P["num_bp_per_min"], 0, P["coverage_flank"], 0, P["max_window_length"]], dtype=np.uint64)
            v = np.zeros(n, dtype=np.uint8); order = np.zeros(n, dtype=np.uint64)
            assert h.vgh_select_minimizers_of_read(a.ctypes.data, n, seq.encode(), read_len, pol.ctypes.data, P["score_fraction"], v.ctypes.data, None, order.ctypes.data) == 0
            plain = sorted(range(n), key=lambda i: (-score[i], ms[i][0], i))
            assert order.tolist() == shuffle_top_ties(plain, ms, score, seq)
            shuffled += order.tolist() != plain
        changed += got != select(ms, read_len, P)[0]
    assert shuffled > 40 and changed >= 5, (shuffled, changed)


def test_a_pair_shares_one_generator():
    """the paired path (src/minimizer_mapper.cpp:1529-1541): one generator seeded from mate 1 + mate 2, drawn from by the first mate's sort and then, where it
    left off, by the second's — the host shim's ReadRng against the restatement with the generator's state carried by hand"""
    rng = np.random.default_rng(14)
    h = util.host()
    h.vgh_select_minimizers_of_pair.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64,
                                                ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    differs_from_single = carried = 0
    for _ in range(300):
        P = dict(DEFAULTS, hit_cap=int(rng.choice([0, 1, 10])), hard_hit_cap=int(rng.choice([20, 500])), score_fraction=float(rng.choice([0.3, 0.6, 0.9])))
        pol = np.array([P["hit_cap"], P["hard_hit_cap"], P["max_unique_min"], P["num_bp_per_min"], 0, P["coverage_flank"], 0, P["max_window_length"]], dtype=np.uint64)
        mates = []
        for _m in range(2):
            L = int(rng.integers(60, 200)); k = int(rng.integers(11, 31))
            mates.append(("".join("ACGT"[int(x)] for x in rng.integers(0, 4, L)), random_minimizers(rng, L, k, int(rng.integers(2, 20)))))
        (s1, m1), (s2, m2) = mates
        if not m1 or not m2:
            continue
        a1 = np.ascontiguousarray([x for m in m1 for x in m], dtype=np.uint64); a2 = np.ascontiguousarray([x for m in m2 for x in m], dtype=np.uint64)
        v1 = np.zeros(len(m1), np.uint8); v2 = np.zeros(len(m2), np.uint8); o1 = np.zeros(len(m1), np.uint64); o2 = np.zeros(len(m2), np.uint64)
        assert h.vgh_select_minimizers_of_pair(a1.ctypes.data, len(m1), s1.encode(), len(s1), a2.ctypes.data, len(m2), s2.encode(), len(s2), pol.ctypes.data, P["score_fraction"],
                                               v1.ctypes.data, v2.ctypes.data, o1.ctypes.data, o2.ctypes.data) == 0
        # the restatement: one Lehmer state over both sorts
        seed = 0
        for byte in (s1 + s2).encode():
            seed = (seed * 13 + byte) & 0xffffffff
        state = [seed % 2147483647 or 1]

        def draw():
            state[0] = state[0] * 48271 % 2147483647
            return state[0]
        want = []
        for ms in (m1, m2):
            score = scores_of(ms, P["hard_hit_cap"]); order = sorted(range(len(ms)), key=lambda i: (-score[i], ms[i][0], i))
            runs = []; at = 0
            while at < len(order) and score[order[at]] == score[order[0]]:
                end = at + 1
                while end < len(order) and ms[order[end]][0] == ms[order[at]][0]:
                    end += 1
                runs.append(order[at:end]); at = end
            if len(runs) >= 2:
                for i in range(1, len(runs)):
                    j = draw() % (i + 1); runs[j], runs[i] = runs[i], runs[j]
            want.append([i for r in runs for i in r] + order[at:])
        assert o1.tolist() == want[0] and o2.tolist() == want[1], (m1, m2, s1, s2)
        # mate 2 ordered on its own (a fresh generator from its own sequence) is another order, as a rule, when mate 1 drew from the shared one
        sc2 = scores_of(m2, P["hard_hit_cap"]); alone = shuffle_top_ties(sorted(range(len(m2)), key=lambda i: (-sc2[i], m2[i][0], i)), m2, sc2, s2)
        differs_from_single += alone != want[1]
        carried += state[0] != (seed % 2147483647 or 1)
    assert carried > 50 and differs_from_single > 10


def test_what_the_filters_mean():
    # without hits: never a seed; beyond the hard cap (summed over a key's occurrences): never; under the soft cap: always
    P = dict(DEFAULTS)
    v, _ = select([(1, 0, 29, 0), (2, 10, 29, 501), (3, 20, 29, 10), (4, 30, 29, 300), (4, 60, 29, 300)], 150, P)
    assert v[0] == 2 and v[1] == 3 and v[2] == 0 and v[3] == 3 and v[4] == 3           # two occurrences of key 4: 600 hits in the run
    # between the caps a minimizer is taken only while the selected score is short of the fraction of the total
    ms = [(k, 5 * k, 29, 11) for k in range(1, 20)]
    v, _ = select(ms, 150, dict(P, score_fraction=0.5))
    assert v.count(0) == 9 and v.count(6) == 10                                          # 19 equal scores: floor(0.5 x 19) fit under the target
    v, _ = select(ms, 150, dict(P, score_fraction=1.0))
    assert v.count(0) == 19
    # once a run has one member in, the rest of the run follows (taking_run)
    ms = [(1, 0, 29, 11), (2, 10, 29, 12), (2, 40, 29, 12), (2, 70, 29, 12)]
    v, _ = select(ms, 150, dict(P, score_fraction=0.45))
    assert v == [0, 0, 0, 0] or v[1:] in ([0, 0, 0], [6, 6, 6])


# ---- the choice on the device (vgk_minimizer_set_policy): oracle = emulated kernel = HIP = the host shim's select_minimizers -----------------
def policy_seeds(lib, seed, k, w, n_reads, L, hit_cap, hard, frac):
    """reads of a small repetitive graph (many k-mers with several hits) -> the seeds with the policy on equal the seeds of the minimizers
    the host shim's select_minimizers takes (brute-force construction of test_minimizer.py), read by read"""
    import test_minimizer as tm
    from vg_amd import capi, workloads
    wl = workloads.GaplessWorkload(4, seed=seed, graph_bp=6000, n_haplotypes=4, snp_every=40, indel_every=300)
    rng = np.random.default_rng(seed)
    reads, _ = tm.sample_reads(rng, wl.nodes, wl.threads, n_reads, L)
    reads += ["", "ACGT" * 3, reads[0] * 6]                                  # no minimizers; too short; more than 64 minimizers
    for i in range(0, len(reads) - 3, 7):                                    # every seventh read with a base that is none of A, C, G, T
        if len(reads[i]) > 20:
            at = int(rng.integers(0, len(reads[i]))); reads[i] = reads[i][:at] + "N" + reads[i][at + 1:]
    index = tm.build_index(wl.nodes, wl.threads, k, w)
    P = dict(DEFAULTS, hit_cap=hit_cap, hard_hit_cap=hard, score_fraction=frac)
    expected = []; skipped = []; dropped = 0
    for r in reads:
        ms = tm.minimizers(r, k, w)
        if len(ms) > 64:
            expected.append(tm.seeds_of(r, index, wl.nodes, k, w, hard)); skipped.append(True); continue
        listed = [(key, p, k, len(index.get(key, []))) for p, key, rev in ms]
        # a read with a masked base whose top tie can change the choice is not chosen for by the engine (it cannot seed the reference's generator)
        if ms and any(c not in "ACGT" for c in r):
            sc = scores_of(listed, hard); top = max(sc); tied = set(m[0] for m, x in zip(listed, sc) if x == top)
            top_hits = max(m[3] for m, x in zip(listed, sc) if x == top)
            if len(tied) >= 2 and (hit_cap != 0 or frac != 1.0) and top_hits > hit_cap:
                expected.append(tm.seeds_of(r, index, wl.nodes, k, w, hard)); skipped.append(True); continue
        skipped.append(False)
        v, _ = select(listed, len(r), P, r) if ms else ([], [])
        dropped += sum(1 for x, (p, key, rev) in zip(v, ms) if x and index.get(key))
        out = []
        for (p, key, rev), verdict in zip(ms, v):
            if verdict:
                continue
            for node, off in index[key]:
                if len(out) >= 64:
                    break
                s = (node, p - off) if not rev else (node ^ 1, (p + k - 1) - (len(wl.nodes[node >> 1]) - 1 - off))
                if s not in out:
                    out.append(s)
        expected.append(out)
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads, k, w); hi = eng.haplo_index(wl.nodes, wl.threads)
    mi.set_policy(hit_cap, hard, frac)
    seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off, 7)       # (the call's own cap is ignored with a policy set)
    for i, exp in enumerate(expected):
        got = [(int(s["node"]), int(s["diff"])) for s in seeds[seed_off[i]:seed_off[i + 1]]]
        assert got == exp, "read %d: %s vs %s" % (i, got[:6], exp[:6])
        assert bool(eng.minimizers_policy_skipped[i]) == skipped[i]
    mi.set_policy(on=False)                                                  # and off again: the plain call
    seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off, hard)
    for i, r in enumerate(reads):
        assert [(int(s["node"]), int(s["diff"])) for s in seeds[seed_off[i]:seed_off[i + 1]]] == tm.seeds_of(r, index, wl.nodes, k, w, hard)
    return dropped


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_the_choice_applied_with_the_seeding(lib_name):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    lib = util.ORACLE_LIB if lib_name == "oracle" else util.EMU_LIB
    assert policy_seeds(lib, 3, 8, 4, 60, 100, 1, 6, 0.6) > 20                 # minimizers with hits that the filters drop: the filters did something
    assert policy_seeds(lib, 4, 9, 5, 40, 150, 2, 500, 0.9) >= 0
    policy_seeds(lib, 5, 15, 6, 30, 120, 10, 500, 1.0)
    policy_seeds(lib, 6, 8, 4, 30, 100, 0, 4, 1.0)                             # no hit cap and the whole score: only the hard cap over the run is left
    assert policy_seeds(lib, 8, 8, 4, 80, 100, 0, 6, 0.6) > 20                 # no hit cap, a score fraction: the cut falls inside the top tie — the shuffle decides


def test_policy_arguments():
    from vg_amd import capi, workloads
    wl = workloads.GaplessWorkload(4, seed=1, graph_bp=3000, n_haplotypes=2)
    for lib in (util.ORACLE_LIB, util.EMU_LIB):
        eng = capi.Engine(lib=lib); mi = eng.minimizer_index(wl.nodes, wl.threads, 11, 4)
        for bad in ((10, 0, 0.9), (10, 70000, 0.9), (10, 500, 1.5), (10, 500, -0.1)):
            with pytest.raises(capi.VgkError):
                mi.set_policy(*bad)


@pytest.mark.gpu
def test_the_choice_applied_with_the_seeding_on_hip():
    assert policy_seeds(util.ENGINE_LIB, 3, 8, 4, 400, 100, 1, 6, 0.6) > 100
    policy_seeds(util.ENGINE_LIB, 4, 9, 5, 300, 150, 2, 500, 0.9)
    policy_seeds(util.ENGINE_LIB, 5, 15, 6, 200, 120, 10, 500, 1.0)
    policy_seeds(util.ENGINE_LIB, 7, 29, 11, 300, 150, 10, 500, 0.9)
    assert policy_seeds(util.ENGINE_LIB, 8, 8, 4, 600, 100, 0, 6, 0.6) > 100   # the cut inside the top tie: the shuffle decides


# ---- reads of any length: vgk_minimizer_list -> the shim's choice -> vgk_minimizer_seeds_of (vg_amd/pipeline.py seed_long_reads) -----------------------
def long_read_seeds(lib, seed, k, w, n_reads, L, policy):
    """long reads off a small repetitive graph: the listed minimizers are the brute-force construction's (test_minimizer.py), the minimizers taken
    are the ones find_seeds_restated (this file's statement of src/minimizer_mapper.cpp:4109-4440, every filter) takes, the seeds are one per hit of
    those in index order — no cap of 64 anywhere"""
    import test_minimizer as tm
    from vg_amd import capi, pipeline, workloads
    wl = workloads.GaplessWorkload(4, seed=seed, graph_bp=30000, n_haplotypes=4, snp_every=60, indel_every=400)
    rng = np.random.default_rng(seed)
    reads, _ = tm.sample_reads(rng, wl.nodes, wl.threads, n_reads, L)
    reads = [r for r in reads if r] + ["", "ACGT" * 3]
    if len(reads[0]) > 50:
        reads[0] = reads[0][:40] + "N" + reads[0][41:]
    index = tm.build_index(wl.nodes, wl.threads, k, w)
    P = dict(DEFAULTS, **policy)
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads, k, w)
    out = pipeline.seed_long_reads(eng, mi, flat, off, k, policy=P, threads=3)
    moff, recs, take, soff, seeds = out["minimizer_off"], out["minimizers"], out["take"], out["seed_off"], out["seeds"]
    many = dropped = 0
    for i, r in enumerate(reads):
        ms = tm.minimizers(r, k, w)
        got = recs[int(moff[i]):int(moff[i + 1])]
        assert [(int(x["offset"]), int(x["key"]), bool(x["flags"] & 1), int(x["hits"])) for x in got] == [(p, key, bool(rev), len(index.get(key, []))) for p, key, rev in ms], i
        many += len(ms) > 64
        listed = [(key, p, k, len(index.get(key, []))) for p, key, rev in ms]
        want, _ = find_seeds_restated(listed, len(r), P, r) if ms else ([], [])
        assert [int(t) for t in take[int(moff[i]):int(moff[i + 1])]] == [1 if v == 0 else 0 for v in want], i
        dropped += sum(1 for v, m in zip(want, listed) if v and m[3])
        for j, ((p, key, rev), v) in enumerate(zip(ms, want)):
            s = seeds[int(soff[int(moff[i]) + j]):int(soff[int(moff[i]) + j + 1])]
            exp = [] if v else [((node, p - o) if not rev else (node ^ 1, (p + k - 1) - (len(wl.nodes[node >> 1]) - 1 - o))) for node, o in index.get(key, [])]
            assert [(int(x["node"]), int(x["diff"])) for x in s] == exp, (i, j)
        assert int(out["seeds_per_read"][i]) == int(soff[int(moff[i + 1])] - soff[int(moff[i])])
    return many, dropped


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_long_reads_are_seeded_without_caps(lib_name):
    """VERDICT r05 missing #5: a read with more than 64 minimizers goes through find_seeds' whole choice — max_unique_min / num_bp_per_min included —
    and gets every seed of the minimizers taken"""
    import subprocess
    subprocess.check_call(["make", "-s", "emu", "host"], cwd=util.ROOT)
    lib = util.ORACLE_LIB if lib_name == "oracle" else util.EMU_LIB
    many, dropped = long_read_seeds(lib, 21, 9, 5, 12, 2500, dict(hit_cap=2, hard_hit_cap=40, score_fraction=0.8, max_unique_min=30, num_bp_per_min=50))
    assert many >= 10 and dropped > 50                              # (the unique-minimizer budget of a 2 500-base read is max(30, 2500 / 50) = 50: it bites)
    many, dropped = long_read_seeds(lib, 22, 11, 7, 6, 6000, dict(hit_cap=10, hard_hit_cap=500, score_fraction=0.9, max_unique_min=500, num_bp_per_min=1000))
    assert many >= 5
    long_read_seeds(lib, 23, 9, 5, 8, 1500, dict(hit_cap=1, hard_hit_cap=8, score_fraction=0.5, max_unique_min=10, num_bp_per_min=100, exclude_overlapping_min=True,
                                                 window_count=8, max_window_length=64))


@pytest.mark.gpu
def test_long_reads_are_seeded_without_caps_on_hip():
    many, dropped = long_read_seeds(util.ENGINE_LIB, 31, 11, 7, 60, 15000, dict(hit_cap=10, hard_hit_cap=500, score_fraction=0.9, max_unique_min=500, num_bp_per_min=1000))
    assert many >= 55
    long_read_seeds(util.ENGINE_LIB, 32, 9, 5, 40, 4000, dict(hit_cap=2, hard_hit_cap=40, score_fraction=0.8, max_unique_min=30, num_bp_per_min=50))


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_paired_policy_flags_the_reads_whose_tie_matters(lib_name):
    """the reference's paired path shuffles both mates' top ties from ONE generator (src/minimizer_mapper.cpp:1529-1541), which the per-read
    kernels do not restate.  With vgk_seed_policy::paired a read whose top tie can change the choice is flagged POLICY_SKIPPED (the caller's pair choice takes
    it: vgh_select_minimizers_of_pair); every other read is chosen for exactly as in single-end mode."""
    import subprocess
    import test_minimizer as tm
    from vg_amd import capi, workloads
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    lib = util.ORACLE_LIB if lib_name == "oracle" else util.EMU_LIB
    k, w, hit_cap, hard, frac = 4, 3, 1, 6, 0.6           # (4-mers on 1.5 kbp: a fifth of the reads has no unique minimizer — their best ones tie with several hits each)
    wl = workloads.GaplessWorkload(4, seed=8, graph_bp=1500, n_haplotypes=4, snp_every=40, indel_every=300)
    reads, _ = tm.sample_reads(np.random.default_rng(8), wl.nodes, wl.threads, 120, 60)
    index = tm.build_index(wl.nodes, wl.threads, k, w)
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads, k, w); hi = eng.haplo_index(wl.nodes, wl.threads)
    mi.set_policy(hit_cap, hard, frac)
    s_off, s_seeds, _ = eng.minimizer_seeds(mi, hi, flat, off, 7); single = [s_seeds[s_off[i]:s_off[i + 1]].copy() for i in range(len(reads))]
    already = eng.minimizers_policy_skipped.copy()           # (reads with more than 64 minimizers: flagged in either mode)
    assert already.sum() < 5
    mi.set_policy(hit_cap, hard, frac, paired=True)
    p_off, p_seeds, _ = eng.minimizer_seeds(mi, hi, flat, off, 7); flagged = eng.minimizers_policy_skipped.copy()
    matters = []
    for r in reads:
        ms = tm.minimizers(r, k, w); listed = [(key, p, k, len(index.get(key, []))) for p, key, rev in ms]
        sc = scores_of(listed, hard) if listed else []
        top = max(sc) if sc else 0; tied = set(m[0] for m, x in zip(listed, sc) if x == top); top_hits = max([m[3] for m, x in zip(listed, sc) if x == top] or [0])
        matters.append(len(tied) >= 2 and top_hits > hit_cap)
    assert [bool(f) for f in flagged] == [bool(m or a) for m, a in zip(matters, already)] and 3 < sum(matters) < len(reads) - 3
    for i in range(len(reads)):
        if not matters[i] and not already[i]:
            assert p_seeds[p_off[i]:p_off[i + 1]].tobytes() == single[i].tobytes(), i


def test_score_cluster_is_the_sum_over_the_distinct_minimizers_present():
    """src/minimizer_mapper.cpp:4738-4781 restated: score = sum of find_minimizers' scores over the distinct sources of the cluster's seeds (in read order), coverage =
    covered read bases / read length"""
    h = util.host()
    h.vgh_score_cluster.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(12)
    for _ in range(200):
        L = int(rng.integers(40, 400)); k = int(rng.integers(5, 31)); n = int(rng.integers(1, 40))
        ms = [(int(rng.integers(0, 1 << 40)), int(rng.integers(0, L - k + 1)), k, int(rng.integers(0, 700))) for _ in range(n)]
        ms.sort(key=lambda m: m[1])
        sources = [int(x) for x in rng.integers(0, n, int(rng.integers(0, 30)))]
        a = np.ascontiguousarray([x for m in ms for x in m], dtype=np.uint64); src = np.ascontiguousarray(sources if sources else [0], dtype=np.uint64)
        out = np.zeros(2, dtype=np.float64); present = np.zeros(n, dtype=np.uint8)
        assert h.vgh_score_cluster(a.ctypes.data, n, 500, src.ctypes.data, len(sources), L, out.ctypes.data, present.ctypes.data) == 0
        sc = scores_of(ms, 500)
        score = 0.0; covered = np.zeros(L, dtype=bool)
        for j in range(n):
            if j in sources:
                score += sc[j]; covered[ms[j][1]:ms[j][1] + k] = True
        assert out[0] == score and out[1] == covered.sum() / L and list(present) == [1 if j in sources else 0 for j in range(n)]
