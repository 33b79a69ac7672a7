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
