"""Windows of one resident graph, packed on the device (vgk_graph_create / vgk_gssw_pack_windows).

Three independent constructions of "the induced subgraph on nodes [first, first + n)" must agree bit for bit:
  * the engine's device-side packer (gssw_pack_device.hpp: arenas derived from the resident tables) — on the CPU under the
    lock-step emulator here, on the MI355X in the gpu tests;
  * the oracle's (oracle/vgo_engine.c: a fresh predecessor CSR per problem, then the per-problem oracle);
  * this file's (numpy: an explicit per-problem graph handed to the ordinary vgk_gssw_pack of the same engine).
The subgraph extraction they stand in for is Mapper's cluster subgraph (src/mapper.cpp:2445-2518) + create_gssw_graph
(src/aligner.cpp:30-85)."""
import os
import subprocess

import numpy as np
import pytest

from gen import random_dag
from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, workloads

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def graph_arrays(nodes, preds):
    node_len = np.array([len(s) for s in nodes], dtype=np.uint32)
    seq = np.frombuffer("".join(nodes).encode(), dtype=np.uint8).copy()
    pred_off = np.concatenate([[0], np.cumsum([len(p) for p in preds])]).astype(np.uint32)
    pred_idx = np.array([q for p in preds for q in p] or [0], dtype=np.uint32)
    return node_len, seq, pred_off, pred_idx


def random_windows(rng, nodes, preds, n, max_read=120, xdrop_fraction=0.4, score_only_fraction=0.1):
    """n window problems over the graph: the read is a noisy walk that starts inside the window (so most align well)."""
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for p in pr:
            succ[p].append(v)
    reads, read_off, first, count, flags, max_gap = [], [0], [], [], [], []
    for _ in range(n):
        a = int(rng.integers(0, len(nodes)))
        k = int(rng.integers(1, min(24, len(nodes) - a) + 1))
        L = int(rng.integers(1, max_read + 1))
        xdrop = rng.random() < xdrop_fraction
        v = a if xdrop else int(rng.integers(a, a + k)); off = 0 if xdrop else int(rng.integers(0, len(nodes[v])))
        out = []
        while len(out) < L:
            if off >= len(nodes[v]):
                nx = [w for w in succ[v] if w < a + k]
                if not nx:
                    break
                v = nx[int(rng.integers(0, len(nx)))]; off = 0
                continue
            c = nodes[v][off]; off += 1
            r = rng.random()
            if r < 0.04:
                c = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.05:
                continue
            elif r < 0.06:
                out.append("ACGT"[int(rng.integers(0, 4))])
            out.append(c)
        if not out or rng.random() < 0.1:
            out = ["ACGTN"[int(rng.integers(0, 5))] for _ in range(L)]
        rd = "".join(out[:L])
        reads.append(np.frombuffer(rd.encode(), dtype=np.uint8)); read_off.append(read_off[-1] + len(rd))
        first.append(a); count.append(k)
        tb = 0 if rng.random() < score_only_fraction else capi.VGK_GSSW_TRACEBACK
        flags.append((capi.VGK_XDROP_PINNED if xdrop else capi.VGK_GSSW_LOCAL) | tb)
        max_gap.append(int(rng.integers(0, 60)))
    return np.concatenate(reads), np.array(read_off), np.array(first), np.array(count), np.array(flags, dtype=np.uint32), np.array(max_gap)


def window_set(graph_col, reads, read_off, first, count, flags, max_gap):
    cols = graph_col[first + count] - graph_col[first]
    return capi.WindowSet(reads, read_off, first, count, flags, max_gap, cols=cols)


def induced_problem_set(nodes, preds, reads, read_off, first, count, flags, max_gap):
    """The same problems as explicit per-problem graphs (edges from outside the window dropped, indices re-based)."""
    problems = []
    for i in range(len(first)):
        a, k = int(first[i]), int(count[i])
        problems.append(dict(read=bytes(reads[read_off[i]:read_off[i + 1]]).decode(), nodes=nodes[a:a + k],
                             preds=[[q - a for q in preds[v] if q >= a] for v in range(a, a + k)],
                             flags=int(flags[i]), pinning=None, max_gap=int(max_gap[i])))
    return capi.ProblemSet.from_lists(problems)


def assert_same(ra, oa, rb, ob, what):
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        bad = np.nonzero(ra[f] != rb[f])[0]
        assert len(bad) == 0, "%s: %s differs at problems %s" % (what, f, bad[:8])
    for i in range(len(ra)):
        a = oa[ra["ops_begin"][i]:ra["ops_begin"][i] + ra["n_ops"][i]]; b = ob[rb["ops_begin"][i]:rb["ops_begin"][i] + rb["n_ops"][i]]
        assert (a.view(np.uint64) == b.view(np.uint64)).all(), "%s: ops of problem %d" % (what, i)


def uniform_local_windows(rng, nodes, preds, n, read_len, k, sub=0.02, indel=0.003):
    """n LOCAL + traceback windows of k nodes each, every read read_len bases long: a noisy walk that starts in the window's first two nodes
    (one lane geometry and windows of similar widths: a batch the packers let speculate)"""
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for p in pr:
            succ[p].append(v)
    reads, first = [], []
    while len(reads) < n:
        a = int(rng.integers(0, len(nodes) - k))
        v = a + int(rng.integers(0, 2)); off = int(rng.integers(0, len(nodes[v])))
        out = []
        while len(out) < read_len:
            if off >= len(nodes[v]):
                nx = [w for w in succ[v] if w < a + k]
                if not nx:
                    break
                v = nx[int(rng.integers(0, len(nx)))]; off = 0
                continue
            c = nodes[v][off]; off += 1
            r = rng.random()
            if r < sub:
                c = "ACGT"[int(rng.integers(0, 4))]
            elif r < sub + indel:
                continue
            elif r < sub + 2 * indel:
                out.append("ACGT"[int(rng.integers(0, 4))])
            out.append(c)
        if len(out) < read_len:
            continue                                                   # the walk ran out of window: another start
        reads.append(np.frombuffer("".join(out[:read_len]).encode(), dtype=np.uint8)); first.append(a)
    read_off = np.arange(n + 1) * read_len
    flags = np.full(n, capi.VGK_GSSW_LOCAL | capi.VGK_GSSW_TRACEBACK, dtype=np.uint32)
    return np.concatenate(reads), read_off, np.array(first), np.full(n, k), flags, np.zeros(n, dtype=np.int64)


def speculative_windows_of_a_variation_graph(lib, n_nodes, n_problems, seed=91):
    """the device-packed windows of a resident DAG with bubbles: the batch speculates (first fill without codes, the misses filled again), and
    every result and op equals the oracle's on the same induced subgraphs"""
    rng = np.random.default_rng(seed)
    nodes, preds = random_dag(rng, n_nodes, 12)
    arrays = graph_arrays(nodes, preds)
    w = uniform_local_windows(rng, nodes, preds, n_problems, read_len=90, k=16)
    for sc in (capi.Scoring.simple(1, 4, 6, 1, 5), capi.Scoring.simple(2, 3, 5, 2, 0)):
        eng = capi.Engine(sc, lib=lib)
        g = eng.graph(*arrays)
        ws = window_set(g.col, *w)
        with eng.pack_windows(g, ws, 0) as b:
            b.run(); b.sync()
            assert b.speculated() and b.kernel_ms(3) > 0, "the window batch did not speculate"
            rw, ow = b.fetch()
        st = eng.speculation_state()
        assert st["observed"] == 1 and 0.0 < st["last_miss"] < 0.9
        ora = capi.Engine(sc, lib=ORACLE_LIB)
        ro, oo = ora.align_windows(ora.graph(*arrays), ws, 0)
        assert_same(rw, ow, ro, oo, "speculative engine windows vs oracle windows")
        assert (rw["status"] == 0).all() and (rw["score"] > 40).mean() > 0.9


def test_emulated_speculative_fill_over_windows_of_a_variation_graph(emu_lib):
    speculative_windows_of_a_variation_graph(emu_lib, n_nodes=600, n_problems=1100)


def run_three_ways(lib, seed, n_nodes, n_problems, scoring=None, ops_per=0):
    rng = np.random.default_rng(seed)
    nodes, preds = random_dag(rng, n_nodes, 12, with_n=0.05)
    arrays = graph_arrays(nodes, preds)
    w = random_windows(rng, nodes, preds, n_problems)
    sc = scoring or capi.Scoring.simple()
    eng = capi.Engine(sc, lib=lib)
    g = eng.graph(*arrays)
    ws = window_set(g.col, *w)
    rw, ow = eng.align_windows(g, ws, ops_per)
    ora = capi.Engine(sc, lib=ORACLE_LIB)
    ro, oo = ora.align_windows(ora.graph(*arrays), ws, ops_per)
    assert_same(rw, ow, ro, oo, "engine windows vs oracle windows")
    rp, op = eng.align(induced_problem_set(nodes, preds, *w), ops_per)
    assert_same(rw, ow, rp, op, "engine windows vs engine per-problem graphs")
    assert (rw["status"] == 0).all()
    return rw


def test_windows_match_oracle_and_explicit_subgraphs(emu_lib):
    res = run_three_ways(emu_lib, 2027, 300, 500)
    assert (res["score"] > 0).sum() > 350


def test_windows_other_scoring_and_fixed_op_budget(emu_lib):
    run_three_ways(emu_lib, 5, 120, 200, scoring=capi.Scoring.simple(2, 3, 5, 2, 7), ops_per=200)


def test_windows_of_the_linear_bench_graph(emu_lib):
    """configs[1] in miniature: the bench's window problems give the results of the bench's per-problem-graph problems."""
    wl = workloads.LinearWorkload(300, ref_len=20_000)
    eng = capi.Engine(lib=emu_lib)
    rp, op = eng.align(wl, 48)
    g = eng.graph(*wl.graph_arrays())
    rw, ow = eng.align_windows(g, wl.windows(), 48)
    assert_same(rw, ow, rp, op, "bench windows vs bench per-problem graphs")
    assert (rw["score"] > 100).mean() > 0.95


def test_window_errors_are_reported_like_a_serial_scan(emu_lib):
    rng = np.random.default_rng(1)
    nodes, preds = random_dag(rng, 40, 12)
    arrays = graph_arrays(nodes, preds)
    for lib in (emu_lib, ORACLE_LIB):
        eng = capi.Engine(lib=lib)
        g = eng.graph(*arrays)
        rd = np.frombuffer(b"ACGT" * 400, dtype=np.uint8)
        ok = dict(reads=rd, read_off=[0, 50, 100], first_node=[0, 5], n_nodes=[10, 10], flags=capi.VGK_GSSW_TRACEBACK)

        def rc(**kw):
            a = dict(ok); a.update(kw)
            try:
                eng.align_windows(g, capi.WindowSet(a["reads"], a["read_off"], a["first_node"], a["n_nodes"], a["flags"], cols=[200, 200]))
            except capi.VgkError as e:
                return str(e)
            return "ok"
        assert rc() == "ok"
        assert "invalid" in rc(n_nodes=[10, 36])                     # window runs off the graph
        assert "invalid" in rc(n_nodes=[0, 10])                      # empty window
        assert "invalid" in rc(read_off=[0, 50, 1700])               # read runs off the buffer
        assert "invalid" in rc(flags=capi.VGK_GSSW_PINNED | capi.VGK_GSSW_TRACEBACK)
        assert "too long" in rc(read_off=[0, 1100, 1200])
        assert "too long" in rc(read_off=[0, 1100, 1700], n_nodes=[10, 36])      # the first failing problem decides
    # a graph that is not topological, has an empty node, or a node beyond the 16-bit run length
    eng = capi.Engine(lib=emu_lib)
    with pytest.raises(capi.VgkError):
        eng.graph([4, 4], rd[:8], [0, 1, 1], [1])
    with pytest.raises(capi.VgkError):
        eng.graph([4, 0], rd[:4], [0, 0, 1], [0])
    with pytest.raises(capi.VgkError):
        eng.graph([70000], np.resize(rd, 70000), [0, 0], [])


@pytest.mark.gpu
def test_windows_on_the_gpu_match_oracle_and_explicit_subgraphs():
    res = run_three_ways(ENGINE_LIB, 99, 3000, 6000)
    assert (res["score"] > 0).sum() > 4000
    run_three_ways(ENGINE_LIB, 100, 500, 1500, scoring=capi.Scoring.simple(2, 3, 5, 2, 7), ops_per=200)


@pytest.mark.gpu
def test_bench_windows_on_the_gpu_equal_per_problem_graphs():
    wl = workloads.LinearWorkload(20_000)
    eng = capi.Engine(lib=ENGINE_LIB)
    rp, op = eng.align(wl, 48)
    g = eng.graph(*wl.graph_arrays())
    rw, ow = eng.align_windows(g, wl.windows(), 48)
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        assert (rw[f] == rp[f]).all(), f
    assert (ow.view(np.uint64) == op.view(np.uint64)).all()


def tails_both_strands(lib, ref_len, n, sample):
    """configs[2] in miniature: giraffe-style tails on either strand of a SNP + indel graph, as windows; engine = oracle."""
    g = workloads.VariationGraph(ref_len=ref_len)
    for graph, seed in ((g, 7), (g.reverse_complement(), 8)):
        tails = workloads.GraphTails(graph, n, seed=seed)
        sub = tails.subset(sample)
        eng = capi.Engine(lib=lib); ora = capi.Engine(lib=ORACLE_LIB)
        ra, oa = eng.align_windows(eng.graph(*graph.arrays()), sub, 48)
        rb, ob = ora.align_windows(ora.graph(*graph.arrays()), sub, 48)
        assert_same(ra, oa, rb, ob, "tails on a variation graph")
        assert (ra["status"] == 0).all()
        assert (ra["score"] >= tails.tails[:sample] - 10).mean() > 0.98          # a tail follows a haplotype: nearly all of it aligns


def test_tails_on_a_variation_graph_as_windows(emu_lib):
    tails_both_strands(emu_lib, 300_000, 4000, 1500)


@pytest.mark.gpu
def test_tails_on_a_variation_graph_as_windows_on_the_gpu():
    tails_both_strands(ENGINE_LIB, 2_000_000, 60_000, 20_000)
