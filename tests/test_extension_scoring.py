"""MinimizerMapper::score_extension_group (vg_amd/host/extension_scoring.cpp; reference src/minimizer_mapper.cpp:5022-5243).  The reference
holds no vectors for it [PARITY-UNPINNED]; the sweep-line form is held to the recurrence it implements, written out directly: the best
chain ending with extension j = its score + the best of nothing, a chain ending exactly where j starts, a chain ending earlier (affine
gap over the skipped read bases) and a chain through an extension that started earlier and is still open (step back into the overlap:
gap open + an extension per overlapping base)."""
import ctypes

import numpy as np

import util


def score_group(read_length, ivs, full_length=False, go=6, ge=1):
    h = util.host()
    h.vgh_score_extension_group.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    a = np.ascontiguousarray([x for iv in ivs for x in iv], dtype=np.int64)
    return h.vgh_score_extension_group(read_length, a.ctypes.data if len(ivs) else None, len(ivs), int(full_length), go, ge)


def recurrence(read_length, ivs, go, ge):
    chain = []
    for j, (b, e, s) in enumerate(ivs):
        best = 0
        for i in range(j):
            bi, ei, _ = ivs[i]
            if ei == b:
                best = max(best, chain[i])
            elif ei < b:
                best = max(best, chain[i] - go - ge * (b - ei - 1))
            elif bi < b < ei:
                best = max(best, chain[i] - go - ge * (ei - b))
        chain.append(best + s)
    return max([0] + chain)


def test_simple_groups():
    assert score_group(100, []) == 0
    assert score_group(100, [(0, 100, 110), (0, 100, 90)], full_length=True) == 110       # full-length: the first extension's own score (:5030)
    assert score_group(0, [(0, 0, 5)]) == 0
    assert score_group(100, [(10, 40, 30)]) == 30
    assert score_group(100, [(10, 40, 30), (40, 70, 25)]) == 55                          # adjacent: no penalty
    assert score_group(100, [(10, 40, 30), (45, 70, 25)]) == 30 + 25 - 6 - 1 * 4          # five read bases skipped: open + four extensions
    assert score_group(100, [(10, 40, 30), (35, 70, 25)]) == 30 + 25 - 6 - 1 * 5          # five bases of overlap stepped back over
    assert score_group(100, [(10, 40, 30), (60, 70, 3)]) == 30                           # not worth the gap
    assert score_group(100, [(0, 30, 20), (30, 100, 60)], go=6, ge=1) == 80              # reaches the end of the read


def test_sweep_equals_the_recurrence_on_random_groups():
    rng = np.random.default_rng(8)
    for _ in range(4000):
        L = int(rng.integers(20, 200))
        n = int(rng.integers(1, 9))
        ivs = []
        for _ in range(n):
            b = int(rng.integers(0, L - 1)); e = int(rng.integers(b + 1, L + 1))
            ivs.append((b, e, int(rng.integers(1, e - b + 6))))
        ivs.sort(key=lambda x: x[0])                      # the extender's order: by read interval
        go = int(rng.integers(1, 9)); ge = int(rng.integers(1, go + 1))
        assert score_group(L, ivs, go=go, ge=ge) == recurrence(L, ivs, go, ge), (L, ivs, go, ge)
