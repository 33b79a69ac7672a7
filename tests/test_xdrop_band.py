"""X-drop extension with dozeu's band restated (vgk_xdrop_band_align).  dozeu's source is not in the reference
snapshot, so the band rules are this engine's reading of the published algorithm [PARITY-UNPINNED]; what CAN be checked:
  * engine (wavefront kernel + host traceback) == oracle (oracle/vgo_xdrop.c, band mode), bit for bit, incl. the cells kept;
  * a banded score never exceeds the exact extension's, and equals it when max_gap_length is generous;
  * every reference X-drop unit test still passes with the band switched on (their optima lie inside dozeu's band);
  * bad input is answered per problem."""
import subprocess

import numpy as np
import pytest

import util
from gen import problem_set, random_problem
from test_golden_gssw_oracle import run_xdrop_cases_with_band
from vg_amd import capi


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    return util.EMU_LIB


def random_xdrop_set(seed, n, max_gap=None, **kw):
    rng = np.random.default_rng(seed)
    probs = [random_problem(rng, mode=capi.VGK_XDROP_PINNED, **kw) for _ in range(n)]
    if max_gap is not None:
        for p in probs:
            p["max_gap"] = max_gap
    return problem_set(probs)


def same(a, b):
    (ra, oa, sa), (rb, ob, sb) = a, b
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        assert (ra[f] == rb[f]).all(), f
    assert len(oa) == len(ob) and (oa.view(np.uint64) == ob.view(np.uint64)).all()
    assert sa == sb


def band_vs_oracle(lib, n, max_read=110):
    for seed, mg in ((1, None), (2, 1), (3, 3), (4, 12), (5, 200)):
        ps = random_xdrop_set(seed, n, mg, max_nodes=12, max_node_len=16, max_read=max_read, with_n=0.05)
        eng = capi.Engine(lib=lib); ora = capi.Engine(lib=util.ORACLE_LIB)
        band = eng.xdrop_band_align(ps)
        same(band, ora.xdrop_band_align(ps))
        exact, _ = ora.align(ps)
        assert (band[0]["score"] <= exact["score"]).all()
        if mg == 200:
            assert (band[0]["score"] == exact["score"]).all()
        if mg == 1:
            assert band[2][0] < 0.2 * band[2][1]               # a tight x-drop keeps a sliver of the rectangle
    sc = capi.Scoring.simple(2, 3, 5, 2, 7)
    ps = random_xdrop_set(9, n // 2, None, max_nodes=8, max_node_len=30, max_read=2 * max_read + 30)
    same(capi.Engine(sc, lib=lib).xdrop_band_align(ps), capi.Engine(sc, lib=util.ORACLE_LIB).xdrop_band_align(ps))


def test_emulated_banded_xdrop_equals_the_oracle(emu_lib):
    band_vs_oracle(emu_lib, 40, max_read=60)          # (the lock-step emulation pays a barrier per cross-lane step: kept small; the gpu test is the big one)


def test_reference_xdrop_cases_hold_with_the_band_on_the_emulated_engine(emu_lib):
    assert run_xdrop_cases_with_band(emu_lib) >= 22


def test_banded_xdrop_bad_input(emu_lib):
    ok = {"read": "ACGTACGT", "nodes": ["ACGT", "ACGT"], "preds": [[], [0]], "flags": capi.VGK_XDROP_PINNED | 16, "pinning": None, "max_gap": 10}
    for lib in (emu_lib, util.ORACLE_LIB):
        eng = capi.Engine(lib=lib)
        res, ops, _ = eng.xdrop_band_align(problem_set([ok, dict(ok, read="A" * 600), dict(ok, preds=[[1], []])]))
        assert list(res["status"]) == [0, -4, -1] and res["score"][0] == 8 + 5
        with pytest.raises(capi.VgkError):                          # a batch of the wrong mode is refused whole
            eng.xdrop_band_align(problem_set([dict(ok, flags=16)]))


def long_graphs_and_a_reused_context(lib, n, scale=1):
    """graphs of more columns than the kernels' LDS stage holds (1024 per problem with four to a wavefront, 4096 alone), read after read
    of every length class, and ONE context for all sets: only the front of a column is written, so whatever an earlier, larger set left
    in the matrices must not be taken for this set's cells"""
    eng = capi.Engine(lib=lib); ora = capi.Engine(lib=util.ORACLE_LIB)
    for seed, kw, mg in ((31, dict(max_nodes=60 // scale, max_node_len=60, max_read=120), 30), (32, dict(max_nodes=40 // scale, max_node_len=140, max_read=400 // scale), 40),
                         (33, dict(max_nodes=10, max_node_len=20, max_read=100), 4), (34, dict(max_nodes=60 // scale, max_node_len=60, max_read=126), None)):
        ps = random_xdrop_set(seed, n, mg, with_n=0.02, **kw)
        same(eng.xdrop_band_align(ps), ora.xdrop_band_align(ps))


def quality_adjusted_band(lib, n):
    """the same on a quality-adjusted context (QualAdjAligner's 25 x qualities table and per-quality bonus)"""
    from qualadj import qual_adj_tables
    qa = qual_adj_tables()
    rng = np.random.default_rng(77)
    probs = [random_problem(rng, mode=capi.VGK_XDROP_PINNED, max_nodes=12, max_node_len=16, max_read=90, with_n=0.03) for _ in range(n)]
    for p in probs:
        p["qual"] = rng.integers(0, 45, len(p["read"])).astype(np.uint8); p["max_gap"] = int(rng.integers(1, 30))
    ps = problem_set(probs)
    same(capi.Engine(lib=lib, qual_adj=qa).xdrop_band_align(ps), capi.Engine(lib=util.ORACLE_LIB, qual_adj=qa).xdrop_band_align(ps))


def test_emulated_band_on_a_quality_adjusted_context(emu_lib):
    quality_adjusted_band(emu_lib, 60)


def test_emulated_band_with_long_graphs_and_a_reused_context(emu_lib):
    long_graphs_and_a_reused_context(emu_lib, 4, scale=3)          # (the emulator's LDS stage holds 64 columns: these graphs refill it many times over)


@pytest.mark.gpu
def test_banded_xdrop_on_hip_equals_the_oracle():
    band_vs_oracle(util.ENGINE_LIB, 1500)
    assert run_xdrop_cases_with_band(util.ENGINE_LIB) >= 22
    long_graphs_and_a_reused_context(util.ENGINE_LIB, 300)
    quality_adjusted_band(util.ENGINE_LIB, 1000)


# Three fills behind one entry: two rows to a register with 16-bit cells (the default when the call's bounds allow,
# xdrop_band_pk_lane), int32 arithmetic over 16-bit cells, int32 throughout (GsswMatrixParams::xb_cell16 = 2, 1, 0): the same answers
def test_emulated_band_with_32_bit_cells(emu_lib, monkeypatch):
    monkeypatch.setenv("VGAMD_XBAND_CELLS32", "1")
    band_vs_oracle(emu_lib, 16, max_read=60)


def test_emulated_band_with_32_bit_arithmetic_over_16_bit_cells(emu_lib, monkeypatch):
    monkeypatch.setenv("VGAMD_XBAND_ARITH32", "1")
    band_vs_oracle(emu_lib, 16, max_read=60)


def test_emulated_band_falls_back_to_32_bit_cells_on_large_scores(emu_lib):
    sc = capi.Scoring.simple(60, 90, 100, 20, 50)          # a step may cost 260: 16-bit cells hold (read + graph) * 260 only for the shortest problems
    ps = random_xdrop_set(21, 24, None, max_nodes=8, max_node_len=16, max_read=60)
    same(capi.Engine(sc, lib=emu_lib).xdrop_band_align(ps), capi.Engine(sc, lib=util.ORACLE_LIB).xdrop_band_align(ps))


@pytest.mark.gpu
def test_banded_xdrop_on_hip_with_32_bit_cells(monkeypatch):
    monkeypatch.setenv("VGAMD_XBAND_ARITH32", "1")
    band_vs_oracle(util.ENGINE_LIB, 600)
    monkeypatch.delenv("VGAMD_XBAND_ARITH32")
    monkeypatch.setenv("VGAMD_XBAND_CELLS32", "1")
    band_vs_oracle(util.ENGINE_LIB, 600)
    sc = capi.Scoring.simple(60, 90, 100, 20, 50)
    ps = random_xdrop_set(21, 400, None, max_nodes=8, max_node_len=16, max_read=100)
    monkeypatch.delenv("VGAMD_XBAND_CELLS32")
    same(capi.Engine(sc, lib=util.ENGINE_LIB).xdrop_band_align(ps), capi.Engine(sc, lib=util.ORACLE_LIB).xdrop_band_align(ps))


# A call runs as sub-batches, two in flight (pack the next while one runs, hand out the one before): forced small here, with problems the
# checks decline in between — the answers and their order are the one-batch call's
def many_sub_batches(lib, n, monkeypatch):
    ps = random_xdrop_set(31, n, None, max_nodes=10, max_node_len=16, max_read=60, with_n=0.02)
    whole = capi.Engine(lib=lib).xdrop_band_align(ps)
    monkeypatch.setenv("VGAMD_MAX_BATCH_BYTES", "30000")
    same(capi.Engine(lib=lib).xdrop_band_align(ps), whole)
    monkeypatch.delenv("VGAMD_MAX_BATCH_BYTES")
    same(whole, capi.Engine(lib=util.ORACLE_LIB).xdrop_band_align(ps))


def test_emulated_band_with_an_op_buffer_too_small(emu_lib, monkeypatch):
    """the caller's op buffer runs out half way: VGK_EOPS for the call and for the problems whose ops did not fit — the same ones, with the
    same bytes for the others, whether the call ran as one batch or as many"""
    import ctypes
    ps = random_xdrop_set(33, 60, None, max_nodes=10, max_node_len=16, max_read=60)
    full = capi.Engine(lib=emu_lib).xdrop_band_align(ps)
    cap = len(full[1]) // 2
    outs = []
    for env in ({}, {"VGAMD_MAX_BATCH_BYTES": "30000"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = capi.Engine(lib=emu_lib)
        res = np.zeros(ps.n, dtype=capi.RESULT_DT); ops = np.zeros(cap, dtype=capi.OP_DT); written = ctypes.c_size_t(); stats = (ctypes.c_uint64 * 2)()
        rc = eng.lib.vgk_xdrop_band_align(eng.h, ps.ptr, ps.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written), ctypes.byref(stats))
        for k in env:
            monkeypatch.delenv(k)
        assert rc == -6
        outs.append((res.copy(), ops[:written.value].copy()))
    one, many = outs
    assert (one[0]["status"] == -6).any() and (one[0]["status"] == 0).any()
    assert one[0].tobytes() == many[0].tobytes() and one[1].tobytes() == many[1].tobytes()
    fits = one[0]["status"] == 0
    assert (one[0]["score"][fits] == full[0]["score"][fits]).all() and (one[0]["n_ops"][fits] == full[0]["n_ops"][fits]).all()


def test_emulated_band_in_many_sub_batches(emu_lib, monkeypatch):
    many_sub_batches(emu_lib, 48, monkeypatch)


@pytest.mark.gpu
def test_band_in_many_sub_batches_on_hip(monkeypatch):
    many_sub_batches(util.ENGINE_LIB, 3000, monkeypatch)
    ps = random_xdrop_set(32, 70000, None, max_nodes=6, max_node_len=12, max_read=40)      # large enough for the call to cut itself in four
    same(capi.Engine(lib=util.ENGINE_LIB).xdrop_band_align(ps), capi.Engine(lib=util.ORACLE_LIB).xdrop_band_align(ps))


# ---- tails of at most 63 bases EIGHT to a wavefront (8 lanes of 8 rows: XlDpp8 in backend_hip.hip; the lane code is the 16-lane class's) ----
def eight_lane_class(lib, n, monkeypatch):
    """reads of every length around the class boundaries (63 | 64 rows: one lane more or fewer; 127 | 128), the same answers from the oracle, and
    from the engine without the class (the default)"""
    rng = np.random.default_rng(77)
    lengths = [1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 55, 56, 57, 61, 62, 63, 64, 65, 71, 72, 96, 126, 127, 128, 129, 160]
    probs = []
    for k in range(n):
        L = lengths[k % len(lengths)]
        p = random_problem(rng, mode=capi.VGK_XDROP_PINNED, max_nodes=14, max_node_len=20, max_read=L, with_n=0.03)
        while len(p["read"]) != L:                                   # (the generator draws a length up to max_read)
            p = random_problem(rng, mode=capi.VGK_XDROP_PINNED, max_nodes=14, max_node_len=20, max_read=L, with_n=0.03)
        if k % 3 == 0:
            p["max_gap"] = int(rng.integers(0, 5))
        probs.append(p)
    ps = problem_set(probs)
    monkeypatch.setenv("VGAMD_XBAND_EIGHTS", "1")                      # (measured slower than four to a wavefront: not the default)
    eng = capi.Engine(lib=lib); ora = capi.Engine(lib=util.ORACLE_LIB)
    got = eng.xdrop_band_align(ps)
    c8, c16, c64 = eng.xdrop_band_last_classes()
    short = sum(1 for p in probs if len(p["read"]) <= 63); mid = sum(1 for p in probs if 63 < len(p["read"]) <= 127)
    assert (c8, c16, c64) == (short, mid, n - short - mid) and c8 > n // 3
    same(got, ora.xdrop_band_align(ps))
    monkeypatch.delenv("VGAMD_XBAND_EIGHTS")
    plain = capi.Engine(lib=lib)
    same(got, plain.xdrop_band_align(ps))
    assert plain.xdrop_band_last_classes() == (0, short + mid, n - short - mid)


def test_emulated_band_with_eight_tails_to_a_wavefront(emu_lib, monkeypatch):
    eight_lane_class(emu_lib, 120, monkeypatch)


@pytest.mark.gpu
def test_band_with_eight_tails_to_a_wavefront_on_hip(monkeypatch):
    eight_lane_class(util.ENGINE_LIB, 6000, monkeypatch)
