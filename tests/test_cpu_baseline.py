"""The CPU baseline of bench.py (oracle/vgo_gssw_fast.c: int16 AVX2 rows, per-thread arenas, 1 B/cell traceback) must give
exactly what the scalar checker (oracle/vgo_gssw.c) gives before its rate may be quoted."""
import ctypes

import numpy as np
import pytest

from gen import problem_set, random_problem
from util import ORACLE_LIB
from vg_amd import capi, workloads


def both(ps, scoring=None, ops_per=0):
    eng = capi.Engine(scoring or capi.Scoring.simple(), lib=ORACLE_LIB)
    eng.lib.vgo_gssw_run_fast.argtypes = [ctypes.c_void_p]
    if not eng.lib.vgo_gssw_fast_supported():
        pytest.skip("no AVX2 on this CPU")
    with eng.pack(ps, ops_per) as b:
        rc = eng.lib.vgo_gssw_run_fast(b.h)
        assert rc == 0, rc
        fast = b.fetch()
    slow = eng.align(ps, ops_per)
    return fast, slow


def assert_identical(fast, slow):
    (rf, of), (rs, os_) = fast, slow
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        bad = np.nonzero(rf[f] != rs[f])[0]
        assert len(bad) == 0, (f, bad[:5], rf[f][bad[:5]], rs[f][bad[:5]])
    assert (of.view(np.uint64) == os_.view(np.uint64)).all()


def test_fast_cpu_path_equals_the_checker_on_random_dags():
    rng = np.random.default_rng(31)
    problems = [random_problem(rng, max_nodes=14, max_node_len=20, max_read=200, with_n=0.1) for _ in range(1500)]
    problems += [random_problem(rng, traceback=False) for _ in range(200)]
    fast, slow = both(problem_set(problems))
    assert_identical(fast, slow)
    assert (fast[0]["score"] > 0).sum() > 1000


def test_fast_cpu_path_other_scorings():
    rng = np.random.default_rng(32)
    problems = [random_problem(rng, max_nodes=10, max_node_len=30, max_read=300) for _ in range(300)]
    for sc in (capi.Scoring.simple(2, 3, 5, 2, 7), capi.Scoring.simple(1, 1, 1, 1, 0), capi.Scoring.simple(5, 4, 9, 3, 11)):
        assert_identical(*both(problem_set(problems), sc))


def test_fast_cpu_path_on_bench_reads():
    wl = workloads.LinearWorkload(3000, ref_len=100_000)
    fast, slow = both(wl, ops_per=48)
    assert_identical(fast, slow)
    assert (fast[0]["score"] > 100).mean() > 0.95
