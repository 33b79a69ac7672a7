"""C-ABI behaviour that does not depend on the device: oversize calls are cut into sub-batches, concurrent
callers on one context serialise correctly, invalid input is rejected with error codes (never a crash or exit)."""
import os
import subprocess
import threading

import numpy as np
import pytest

from gen import problem_set, random_problem
from util import ORACLE_LIB, ROOT
from vg_amd import capi

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvgamd_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def _same(ra, oa, rb, ob, n):
    return all(ra["score"][i] == rb["score"][i] and ra["status"][i] == rb["status"][i] and
               capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob) for i in range(n))


def test_oversize_call_is_split_into_sub_batches(emu_lib, monkeypatch):
    rng = np.random.default_rng(8)
    problems = [random_problem(rng) for _ in range(300)]
    ps = problem_set(problems)
    whole = capi.Engine(lib=emu_lib).align(ps)
    monkeypatch.setenv("VGAMD_MAX_BATCH_BYTES", "20000")      # a handful of problems per sub-batch
    import ctypes
    eng = capi.Engine(lib=emu_lib)
    res = np.zeros(ps.n, dtype=capi.RESULT_DT)
    cap = int(np.diff(ps.read_off).sum() + np.diff(ps.seq_off).sum() + 2 * ps.n)
    ops = np.zeros(cap, dtype=capi.OP_DT)
    written = ctypes.c_size_t()
    rc = eng.lib.vgk_gssw_align(eng.h, ps.ptr, ps.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written))
    assert rc == 0
    assert _same(whole[0], whole[1], res, ops[:written.value], ps.n)


def test_concurrent_callers_on_one_context(emu_lib):
    rng = np.random.default_rng(9)
    sets = [problem_set([random_problem(rng) for _ in range(120)]) for _ in range(4)]
    eng = capi.Engine(lib=emu_lib)
    expect = [capi.Engine(lib=ORACLE_LIB).align(ps) for ps in sets]
    got = [None] * 4

    def work(i):
        got[i] = eng.align(sets[i])
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for i in range(4):
        assert _same(got[i][0], got[i][1], expect[i][0], expect[i][1], sets[i].n)


def test_invalid_input_is_rejected_with_codes(emu_lib):
    eng = capi.Engine(lib=emu_lib)
    ok = {"read": "ACGT", "nodes": ["ACGT", "AC"], "preds": [[], [0]], "flags": 16, "pinning": None}
    bad_order = dict(ok, preds=[[1], []])                      # predecessor after its successor: not topological
    empty_node = dict(ok, nodes=["ACGT", ""])
    too_long = dict(ok, read="A" * 1025)
    no_pin = dict(ok, flags=17, pinning=None)
    for bad in (bad_order, empty_node, too_long, no_pin):
        with pytest.raises(capi.VgkError):
            eng.align(problem_set([bad]))
    with pytest.raises(capi.VgkError):                         # scoring the packed arithmetic cannot hold
        capi.Engine(capi.Scoring.simple(100, 120, 6, 1, 30), lib=emu_lib)
    res, ops = eng.align(problem_set([ok]))                    # the context is still usable afterwards
    assert res["score"][0] == 4 + 10


def test_banded_call_is_split_into_sub_batches_and_rejects_bad_input(emu_lib, monkeypatch):
    import gen
    rng = np.random.default_rng(10)
    problems = [gen.random_banded_problem(rng) for _ in range(24)]
    bs = capi.BandedSet.from_lists(problems)
    whole = capi.Engine(lib=emu_lib).banded_align(bs)
    monkeypatch.setenv("VGAMD_MAX_BATCH_BYTES", "30000")
    cut = capi.Engine(lib=emu_lib).banded_align(bs)
    for a, b in zip(whole, cut):
        assert (a == b).all()
    # rerun needs a single resident sub-batch
    eng = capi.Engine(lib=emu_lib)
    eng.banded_align(bs)
    with pytest.raises(capi.VgkError):
        eng.banded_rerun()
    monkeypatch.delenv("VGAMD_MAX_BATCH_BYTES")
    eng = capi.Engine(lib=emu_lib)
    first = eng.banded_align(bs)
    eng.banded_rerun()                                               # same kernels on the resident inputs: nothing to fetch, must not fail
    again = eng.banded_align(bs)
    for a, b in zip(first, again):
        assert (a == b).all()
    # per-problem errors never abort the batch: a backward edge, an empty read, a budget of one cell
    bad = [dict(read="ACGT", nodes=["AC", "GT"], preds=[[1], []], band_padding=1),
           dict(read="", nodes=["ACGT"], preds=[[]], band_padding=1),
           dict(read="ACGTACGT", nodes=["ACGTACGT"], preds=[[]], band_padding=2, max_cells=1),
           dict(read="ACGT", nodes=["ACGT"], preds=[[]], band_padding=1)]
    res, ops = capi.Engine(lib=emu_lib).banded_align(capi.BandedSet.from_lists(bad))
    assert list(res["status"]) == [-1, -1, -7, 0] and res["score"][3] == 4
    ores, _ = capi.Engine(lib=ORACLE_LIB).banded_align(capi.BandedSet.from_lists(bad))
    assert list(ores["status"]) == list(res["status"])


def test_gapless_limits_and_bad_input_are_reported_per_problem(emu_lib):
    eng = capi.Engine(lib=emu_lib)
    index = eng.haplo_index(["ACGTACGTAC", "GGGTTTAAAC"], [[0, 2]])
    ok = dict(read="ACGTACGTACGGG", seeds=[(0, 0)])
    too_many = dict(read="ACGTACGTAC", seeds=[(o, -k) for o in range(4) for k in range(10)] + [(o, k) for o in range(4) for k in range(1, 8)])      # 68 seeds: beyond the 64 the kernel takes
    out_of_range = dict(read="ACGT", seeds=[(99, 0)])
    no_seeds = dict(read="ACGT", seeds=[])
    res, ext, nodes, mism = eng.gapless_extend(index, [ok, too_many, out_of_range, no_seeds, ok])
    assert list(res["status"]) == [0, -7, -1, 0, 0]
    assert list(res["n_ext"]) == [1, 0, 0, 0, 1] and res["full_length"][0] == 1
    assert ext["score"][0] == 13 + 10 and list(nodes[:2]) == [0, 2]
    # a thread may go round a cycle: the seed still extends along it
    cyc = eng.haplo_index(["AC", "GT"], [[0, 2, 0, 2]])
    res, ext, nodes, mism = eng.gapless_extend(cyc, [dict(read="ACGTACGT", seeds=[(0, 0)])])
    assert res["status"][0] == 0 and res["full_length"][0] == 1 and list(nodes[:4]) == [0, 2, 0, 2] and ext["score"][0] == 8 + 10


def test_gssw_align_answers_out_of_range_problems_per_problem(emu_lib):
    import ctypes
    rng = np.random.default_rng(12)
    good = [random_problem(rng, max_read=60) for _ in range(6)]
    long_read = dict(good[0], read="ACGT" * 300)                      # 1200 bases: beyond the packed kernels' 1024 rows -> the wide route answers it
    huge_read = dict(good[0], read="ACGT" * 16384)                    # 65536 bases: beyond vgk_op's 16-bit run length -> declined, alone
    problems = good[:3] + [long_read] + good[3:] + [huge_read]
    ps = problem_set(problems)
    eng = capi.Engine(lib=emu_lib)
    res = np.zeros(ps.n, dtype=capi.RESULT_DT)
    cap = int(np.diff(ps.read_off).sum() + np.diff(ps.seq_off).sum() + 2 * ps.n)
    ops = np.zeros(cap, dtype=capi.OP_DT)
    written = ctypes.c_size_t()
    assert eng.lib.vgk_gssw_align(eng.h, ps.ptr, ps.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written)) == 0
    assert res["status"][7] == -4                                      # VGK_ETOOLONG for that read only
    ref = capi.Engine(lib=ORACLE_LIB).align(problem_set(good))
    one = capi.Engine(lib=ORACLE_LIB).align(problem_set([long_read]))
    assert res["status"][3] == 0 and res["score"][3] == one[0]["score"][0] and capi.cigar_string(res[3], ops) == capi.cigar_string(one[0][0], one[1])
    keep = [0, 1, 2, 4, 5, 6]
    for k, i in enumerate(keep):
        assert res["score"][i] == ref[0]["score"][k] and res["status"][i] == 0
        assert capi.cigar_string(res[i], ops) == capi.cigar_string(ref[0][k], ref[1])
    # the strict batch API still refuses the whole batch
    with pytest.raises(capi.VgkError):
        eng.pack(ps)


def test_wfa_bad_input_limits_and_rerun(emu_lib):
    import ctypes
    for lib in (emu_lib, ORACLE_LIB):
        eng = capi.Engine(lib=lib)
        index = eng.haplo_index(["ACGTACGTAC", "GGGTTTAAAC"], [[0, 2]])
        ok = dict(seq="GTACGTACGG", mode="connect", **{"from": (0, 1), "to": (2, 2)})
        bad_offset = dict(ok, **{"from": (0, 10)})                      # offset past the node
        no_such_from = dict(seq="ACGT", mode="suffix", **{"from": (99, 0)})   # !has_node(from): an alignment that is not ok (:2059-2064)
        bad_prefix = dict(seq="ACGT", mode="prefix", to=(99, 0))          # prefix needs the target node's length
        empty = dict(seq="", mode="connect", **{"from": (0, 8), "to": (2, 1)})
        res, paths, edits = eng.wfa_extend(index, [ok, bad_offset, no_such_from, bad_prefix, empty, ok])
        assert list(res["status"]) == [0, -1, 0, -1, 0, 0], lib
        assert list(res["ok"]) == [1, 0, 0, 0, 1, 1]
        assert res["score"][0] == 10 and res["score"][5] == 10 and list(paths[:2]) == [0, 2]
        assert res["score"][4] == -(6 + 1) and res["n_edits"][4] == 1 and int(edits[res["edit_begin"][4]]) == (2 << 2 | capi.WFA_DELETION)
        # an error model that does not make sense, scoring WFA cannot convert
        with pytest.raises(capi.VgkError):
            eng.wfa_extend(index, [ok], ((0.1, 2, 1), (0.05, 1, 10), (0.1, 1, 20), (0.1, 10, 200)))
        with pytest.raises(capi.VgkError):
            e2 = capi.Engine(capi.Scoring.simple(1, 4, 1, 2, 5), lib=lib)      # gap_open < gap_extend
            e2.wfa_extend(e2.haplo_index(["ACGT"], [[0]]), [dict(seq="A", mode="suffix", **{"from": (0, 0)})])
        # a score cap beyond the kernel's penalty table is refused for that problem only
        if lib == emu_lib:
            big = ((1.0, 50, 50), (0.05, 1, 10), (0.1, 1, 20), (0.1, 10, 200))
            r2, _, _ = eng.wfa_extend(index, [ok], big)
            assert r2["status"][0] == -7
            # output arrays too small: VGK_EOPS for the call and the problem that did not fit
            ws = capi.WfaSet.from_lists([ok, ok]); ws.path_cap = 3; ws.edit_cap = 8
            r3 = np.zeros(2, dtype=capi.WFA_RESULT_DT); p3 = np.zeros(3, np.uint32); e3 = np.zeros(8, np.uint32); w = (ctypes.c_size_t * 2)()
            rc = eng.lib.vgk_wfa_extend(eng.h, index.h, None, ws.array.ctypes.data, 2, r3.ctypes.data, p3.ctypes.data, 3, e3.ctypes.data, 8, ctypes.byref(w))
            assert rc == -6 and list(r3["status"]) == [0, -6] and w[0] == 2
            eng.wfa_extend(index, [ok, ok]); eng.wfa_rerun(); assert eng.wfa_last_ms() >= 0.0
        else:
            with pytest.raises(capi.VgkError):
                eng.wfa_rerun()                                              # nothing is resident on the CPU


def test_pinned_multi_bad_input_and_sub_batches(emu_lib, monkeypatch):
    rng = np.random.default_rng(14)
    good = [random_problem(rng, max_nodes=6, max_node_len=8, max_read=30, mode=capi.VGK_GSSW_PINNED) for _ in range(12)]
    local = random_problem(rng, max_nodes=4, max_node_len=8, max_read=20, mode=capi.VGK_GSSW_LOCAL)      # not a pinned problem
    problems = good[:5] + [local] + good[5:]
    ps = problem_set(problems)
    for lib in (emu_lib, ORACLE_LIB):
        res, cnt, ops = capi.Engine(lib=lib).align_multi(ps, 4)
        assert res[5, 0]["status"] == -1 and cnt[5] == 0
        assert all(res[i, 0]["status"] == 0 for i in range(ps.n) if i != 5)
    whole = capi.Engine(lib=emu_lib).align_multi(ps, 4)
    monkeypatch.setenv("VGAMD_MAX_BATCH_BYTES", "20000")           # a few problems' matrices per sub-batch
    cut = capi.Engine(lib=emu_lib).align_multi(ps, 4)
    for a, b in zip(whole, cut):
        assert (a == b).all()
    with pytest.raises(capi.VgkError):
        capi.Engine(lib=emu_lib).align_multi(ps, 0)


def test_device_arenas_of_freed_batches_are_reused_without_leaking_state(emu_lib):
    """A freed batch's device arenas go to a pool on the context and serve the next pack (a 35 GB allocation costs about a second on
    the GPU); the next batch must not see anything of the previous one — smaller batch, other modes, then a larger one."""
    rng = np.random.default_rng(31)
    ora = capi.Engine(lib=ORACLE_LIB)
    eng = capi.Engine(lib=emu_lib)
    sets = [problem_set([random_problem(rng, max_nodes=12, max_node_len=20, max_read=150) for _ in range(200)]),
            problem_set([random_problem(rng, max_read=40, mode=capi.VGK_XDROP_PINNED) for _ in range(150)]),
            problem_set([random_problem(rng, max_read=60, traceback=False) for _ in range(90)]),
            problem_set([random_problem(rng, max_nodes=14, max_node_len=24, max_read=200) for _ in range(260)])]
    for round_ in range(2):
        for ps in sets:
            ra, oa = eng.align(ps)
            rb, ob = ora.align(ps)
            assert _same(ra, oa, rb, ob, ps.n), round_
    # two batches alive at once take different arenas
    with eng.pack(sets[0], 0) as a, eng.pack(sets[3], 0) as b:
        a.run(); b.run()
        ra, oa = a.fetch(); rb, ob = b.fetch()
    ea, eoa = ora.align(sets[0]); eb, eob = ora.align(sets[3])
    assert _same(ra, oa, ea, eoa, sets[0].n) and _same(rb, ob, eb, eob, sets[3].n)


def test_large_batch_goes_through_the_threaded_packing_paths(emu_lib):
    """140 000 small problems: enough for the chunked prefix sums, the threaded wave description and the threaded copies of
    vgk_gssw_pack / vgk_gssw_fetch to run on several host threads (smaller batches take their serial branches)."""
    rng = np.random.default_rng(77)
    base = [random_problem(rng, max_nodes=3, max_node_len=6, max_read=int(rng.choice([4, 9, 17, 30]))) for _ in range(61)]
    base += [random_problem(rng, max_nodes=3, max_node_len=6, max_read=12, mode=capi.VGK_XDROP_PINNED) for _ in range(18)]
    base += [random_problem(rng, max_nodes=2, max_node_len=5, max_read=8, traceback=False) for _ in range(18)]
    problems = [base[i % len(base)] for i in range(140000)]
    ps = problem_set(problems)
    ra, oa = capi.Engine(lib=emu_lib).align(ps)
    rb, ob = capi.Engine(lib=ORACLE_LIB).align(ps)
    for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops", "ops_begin"):
        assert (ra[f] == rb[f]).all(), f
    assert len(oa) == len(ob) and (oa.view(np.uint64) == ob.view(np.uint64)).all()


def test_runs_longer_than_a_16_bit_op_length_are_refused_not_wrapped(emu_lib):
    """vgk_op.len is uint16: a 70000 bp node with an identical read used to come back as one op of length 70000 mod 65536
    (ADVICE r1).  Engine and oracle now answer VGK_ETOOBIG; a caller keeps its CPU path for such problems (vg chops nodes to
    <= 1024 bp, so none arise from vg graphs)."""
    big = "ACGT" * 17500
    banded = [dict(read=big, nodes=[big], preds=[[]], band_padding=1), dict(read="ACGT", nodes=["ACGT"], preds=[[]], band_padding=1)]
    for lib in (emu_lib, ORACLE_LIB):
        res, _ = capi.Engine(lib=lib).banded_align(capi.BandedSet.from_lists(banded))
        assert list(res["status"]) == [-7, 0], lib
    bs = capi.BandedSet.from_lists([dict(read=big, nodes=["AC", "GT"], preds=[[], [0]], band_padding=1)])      # the read alone is too long
    for lib in (emu_lib, ORACLE_LIB):
        assert capi.Engine(lib=lib).banded_align(bs)[0]["status"][0] == -7
    gssw = {"read": "ACGTACGT", "nodes": [big], "preds": [[]], "flags": 16, "pinning": None}
    with pytest.raises(capi.VgkError, match="too big"):
        capi.Engine(lib=emu_lib).align(problem_set([gssw]))
    res, _ = capi.Engine(lib=ORACLE_LIB).align(problem_set([gssw]))
    assert res["status"][0] == -7
