"""MinimizerMapper::faster_cap (vg_amd/host/mapq_cap.cpp), held to the reference's two unit tests for it
(src/unittest/minimizer_mapper.cpp:154-252): the cap of a read covered in explored minimizers is finite — for 150 G's under 25-base cores
with 10-base flanks at every offset, and for a fuzz of random agglomerations over 100 G's at quality 60.  The minimizer's hash is
gbwtgraph's for the all-G k-mer (Thomas Wang's mix of the 2-bit key: vg_amd/csrc/minimizer_device.hpp, pinned on test/primers/y.min)."""
import ctypes
import math

import numpy as np

import util

M64 = (1 << 64) - 1


def wang_hash(key):        # mz_hash (minimizer_device.hpp)
    key = (~key + (key << 21)) & M64; key ^= key >> 24
    key = (key + (key << 3) + (key << 8)) & M64; key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M64; key ^= key >> 28
    return (key + (key << 31)) & M64


def g_key(k):              # 'G' * k, two bits per base, first base highest
    v = 0
    for _ in range(k):
        v = (v << 2) | 2
    return v


def faster_cap(minimizers, explored, sequence, quality):
    h = util.host()
    h.vgh_faster_cap.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    ms = np.ascontiguousarray(minimizers, dtype=np.uint64).reshape(-1, 6)
    ex = np.ascontiguousarray(explored, dtype=np.uint64)
    cap = ctypes.c_double()
    rc = h.vgh_faster_cap(ms.ctypes.data, len(ms), ex.ctypes.data, len(ex), sequence.encode(), bytes(quality), len(quality), ctypes.byref(cap))
    assert rc == 0, h.vgh_last_error().decode()
    return cap.value


def cover_in_minimizers(sequence, core_width, flank_width, stride):
    """cover_in_minimizers (:110-152)"""
    hash_ = wang_hash(g_key(core_width))
    ms = []
    L = len(sequence)
    core_start = 0
    while core_start + core_width < L:
        if core_start <= flank_width:
            start, length = 0, core_width + flank_width + core_start
        elif L - core_start - core_width <= flank_width:
            start = core_start - flank_width; length = L - start - 1
        else:
            start, length = core_start - flank_width, core_width + 2 * flank_width
        ms.append((hash_, core_start, 0, start, length, core_width))
        core_start += stride
    return ms


def test_cap_is_not_confused_by_excessive_gs():
    """:154-176"""
    sequence = "G" * 150; quality = [0x1E] * 150
    ms = cover_in_minimizers(sequence, 25, 10, 1)
    cap = faster_cap(ms, list(range(len(ms))), sequence, quality)
    assert not math.isinf(cap) and cap > 0


def test_cap_is_not_confused_by_fuzzing_with_high_base_qualities():
    """:178-252 (20 000 of its 100 000 tries; the generator is the test's own, seeded)"""
    rng = np.random.default_rng(2026)
    sequence = "G" * 100; quality = [60] * 100
    L = len(sequence)
    hashes = {k: wang_hash(g_key(k)) for k in range(1, 33)}
    for _ in range(20000):
        n = int(rng.integers(0, 100)) + 5
        ms = []
        for _ in range(n):
            core_width = int(rng.integers(0, min(L // 2 - 1, 31))) + 1
            run_length = int(rng.integers(0, min(L - core_width, 32 - core_width)))
            flank_width = int(rng.integers(0, 10))
            core_start = int(rng.integers(0, L - core_width - run_length))
            start = core_start; length = core_width + run_length + 2 * flank_width
            if flank_width > start:
                length -= flank_width - start; start = 0
            else:
                start -= flank_width
            if start + length > L:
                length = L - start
            ms.append((hashes[core_width], core_start, 0, start, length, core_width))
        cap = faster_cap(ms, list(range(n)), sequence, quality)
        assert not math.isinf(cap)


def test_cap_without_qualities_is_infinite_and_grows_with_quality():
    ms = cover_in_minimizers("G" * 150, 25, 10, 5)
    assert math.isinf(faster_cap(ms, list(range(len(ms))), "G" * 150, []))
    lo = faster_cap(ms, list(range(len(ms))), "G" * 150, [10] * 150)
    hi = faster_cap(ms, list(range(len(ms))), "G" * 150, [40] * 150)
    assert 0 < lo < hi
    # a single minimizer: the cap is the Phred of "some base of its agglomeration is wrong in a way that disrupts it"
    one = [(wang_hash(g_key(20)), 10, 0, 5, 30, 20)]
    cap = faster_cap(one, [0], "G" * 60, [20] * 60)
    p = 0.0
    for i in range(5, 35):
        q = 0.01
        if not (10 <= i < 30):
            n = min(20, i - 5 + 1, 35 - i)
            bucket = wang_hash(g_key(20)) >> 56
            q *= 1.0 - (1.0 - (2 * bucket + 1) / 512.0) ** n
        p = p + q - p * q
    assert abs(cap - (-10 * math.log10(p))) < 1e-9
