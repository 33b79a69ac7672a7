"""Extension windows (vgk_gssw_pack_extensions): one pass of Aligner::align_xdrop — a pinned X-drop extension from a position INSIDE a window
of the resident graph (DozeuInterface::align, src/dozeu_interface.cpp:608-685; the sub-DAG as :236-283 hands it to dozeu).

Three independent constructions of the sub-DAG must agree bit for bit:
  * the engine's device-side packer (gssw_pack_device.hpp ext_size_one / ext_emit_one: reachability, compaction and reversal from the resident
    tables) — under the lock-step emulator here, on the MI355X in the gpu test;
  * the oracle's (oracle/vgo_engine.c vgk_gssw_pack_extensions: explicit strings and a fresh predecessor CSR per problem);
  * this file's (python: the sub-DAG built node by node the way vg_amd/host/aligner.cpp xdrop_extend_prepare builds it from a HandleGraph,
    handed to the ordinary vgk_gssw_pack as an explicit per-problem graph), with the results' nodes mapped back by hand."""
import subprocess

import numpy as np
import pytest

from gen import random_dag
from test_windows import graph_arrays
from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def sub_dag(nodes, preds, first, count, start, start_offset, leftward):
    """-> (sub nodes' sequences, their predecessor lists, window index of every sub node) — or ([], [], []) when nothing lies that way"""
    a, b = first, first + count
    succ = {v: [] for v in range(a, b)}
    for v in range(a, b):
        for q in preds[v]:
            if q >= a:
                succ[q].append(v)
    reach = {start}
    order = range(start, a - 1, -1) if leftward else range(start, b)
    toward = (lambda v: [q for q in succ[v] if q <= start]) if leftward else (lambda v: [q for q in preds[v] if start <= q < b])
    for v in order:
        if v != start and any(q in reach for q in toward(v)):
            reach.add(v)
    start_seq = nodes[start][:start_offset][::-1] if leftward else nodes[start][start_offset:]
    kept = [v for v in order if v in reach and not (v == start and not start_seq)]
    at = {v: k for k, v in enumerate(kept)}
    seqs = [start_seq if v == start else (nodes[v][::-1] if leftward else nodes[v]) for v in kept]
    plist = [[] if v == start else [at[q] for q in toward(v) if q in at] for v in kept]
    return seqs, plist, [v - a for v in kept]


def random_extensions(rng, nodes, preds, n, max_read=140):
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for p in pr:
            succ[p].append(v)
    reads, read_off, rows = [], [0], []
    while len(rows) < n:
        a = int(rng.integers(0, len(nodes) - 2)); k = int(rng.integers(1, min(30, len(nodes) - a) + 1))
        s = int(rng.integers(a, a + k))
        mode = rng.random()
        so = 0 if mode < 0.1 else len(nodes[s]) if mode < 0.2 else int(rng.integers(0, len(nodes[s]) + 1))
        left = int(rng.random() < 0.5)
        L = int(rng.integers(2, max_read + 1))
        qo = int(rng.integers(1, L + 1)) if left else int(rng.integers(0, L))
        # the read part on the extension's side: a noisy walk from the start position (so that most extensions align), the rest random
        qlen = qo if left else L - qo
        v, off = s, so
        out = []
        while len(out) < qlen:
            if left:
                if off == 0:
                    nx = [q for q in preds[v] if q >= a]
                    if not nx:
                        break
                    v = nx[int(rng.integers(0, len(nx)))]; off = len(nodes[v]); continue
                off -= 1; c = nodes[v][off]
            else:
                if off >= len(nodes[v]):
                    nx = [w for w in succ[v] if w < a + k]
                    if not nx:
                        break
                    v = nx[int(rng.integers(0, len(nx)))]; off = 0; continue
                c = nodes[v][off]; off += 1
            r = rng.random()
            if r < 0.04:
                c = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.05:
                continue
            elif r < 0.06:
                out.append("ACGT"[int(rng.integers(0, 4))])
            out.append(c)
        part = "".join(out[:qlen]) + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, qlen - min(qlen, len(out))))
        other = "".join("ACGTN"[int(x)] for x in rng.integers(0, 5, L - qlen))
        rd = (part[::-1] + other) if left else (other + part)
        assert len(rd) == L
        reads.append(np.frombuffer(rd.encode(), dtype=np.uint8)); read_off.append(read_off[-1] + L)
        tb = 0 if rng.random() < 0.15 else capi.VGK_GSSW_TRACEBACK
        rows.append((a, k, capi.VGK_XDROP_PINNED | tb, int(rng.integers(0, 60)), s, so, qo, left))
    cols = np.array([sum(len(nodes[v]) for v in range(r[0], r[0] + r[1])) for r in rows])
    f = list(zip(*rows))
    return capi.ExtensionSet(np.concatenate(reads), np.array(read_off), f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], cols=cols)


def by_hand(eng, nodes, preds, es):
    """the same problems as explicit per-problem graphs through the ordinary packer, mapped back to the window's node numbering"""
    problems, maps = [], []
    a = es.array
    for i in range(es.n):
        seqs, plist, kept = sub_dag(nodes, preds, int(a["first_node"][i]), int(a["n_nodes"][i]), int(a["start_node"][i]), int(a["start_offset"][i]), bool(a["leftward"][i]))
        rd = bytes(es.reads[es.read_off[i]:es.read_off[i + 1]]).decode()
        q = rd[:int(a["query_offset"][i])][::-1] if a["leftward"][i] else rd[int(a["query_offset"][i]):]
        maps.append(kept)
        if not kept:
            seqs, plist = ["N"], [[]]
        problems.append(dict(read=q, nodes=seqs, preds=plist, flags=int(a["flags"][i]), pinning=None, max_gap=int(a["max_gap_length"][i])))
    res, ops = eng.align(capi.ProblemSet.from_lists(problems))
    res = res.copy(); ops = ops.copy()
    for i in range(es.n):
        r = res[i]
        if not maps[i]:
            res["score"][i] = 0; res["n_ops"][i] = 0; res["end_node"][i] = res["end_offset"][i] = res["end_read"][i] = -1; res["first_offset"][i] = 0
            continue
        if r["end_node"] >= 0:
            res["end_node"][i] = maps[i][int(r["end_node"])]
        for k in range(int(r["ops_begin"]), int(r["ops_begin"]) + int(r["n_ops"])):
            ops["node"][k] = maps[i][int(ops["node"][k])]
    return res, ops


def same(ra, oa, rb, ob, what):
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        bad = np.nonzero(ra[f] != rb[f])[0]
        assert len(bad) == 0, "%s: %s differs at problems %s: %s vs %s" % (what, f, bad[:6], ra[f][bad[:6]], rb[f][bad[:6]])
    for i in range(len(ra)):
        x = oa[ra["ops_begin"][i]:ra["ops_begin"][i] + ra["n_ops"][i]]; y = ob[rb["ops_begin"][i]:rb["ops_begin"][i] + rb["n_ops"][i]]
        assert (x.view(np.uint64) == y.view(np.uint64)).all(), "%s: ops of problem %d" % (what, i)


def three_ways(lib, seed, n_nodes, n_problems, scoring=None):
    rng = np.random.default_rng(seed)
    nodes, preds = random_dag(rng, n_nodes, 14, with_n=0.03)
    preds = [sorted(p) for p in preds]
    arrays = graph_arrays(nodes, preds)
    es = random_extensions(rng, nodes, preds, n_problems)
    sc = scoring or capi.Scoring.simple()
    eng = capi.Engine(sc, lib=lib)
    rw, ow = eng.align_extensions(eng.graph(*arrays), es)
    ora = capi.Engine(sc, lib=ORACLE_LIB)
    ro, oo = ora.align_extensions(ora.graph(*arrays), es)
    same(rw, ow, ro, oo, "engine extension windows vs the oracle's")
    rh, oh = by_hand(ora, nodes, preds, es)
    same(ro, oo, rh, oh, "the oracle's extension windows vs sub-DAGs built by hand")
    assert (rw["status"] == 0).all()
    return rw, es


def test_emulated_extension_windows_three_ways(emu_lib):
    rw, es = three_ways(emu_lib, 5, 300, 700)
    assert (rw["score"] > 10).mean() > 0.3                                   # a good share of the extensions align a good part of their read
    left = es.array["leftward"] != 0
    assert (rw["score"][left] > 10).sum() > 60 and (rw["score"][~left] > 10).sum() > 60
    assert ((rw["score"] == 0) & (rw["end_node"] == -1)).sum() >= 1          # ... and some had nothing in their direction
    three_ways(emu_lib, 6, 120, 300, capi.Scoring.simple(2, 3, 5, 2, 7))


def test_extension_windows_reject_what_a_window_would():
    ora = capi.Engine(lib=ORACLE_LIB)
    nodes, preds = ["ACGT", "GGCA", "TTAC"], [[], [0], [1]]
    g = ora.graph(*graph_arrays(nodes, preds))
    rd = np.frombuffer(b"ACGTGG", dtype=np.uint8)
    ok = dict(first_node=[0], n_nodes=[3], flags=[capi.VGK_XDROP_PINNED], max_gap=[10], start_node=[1], start_offset=[1], query_offset=[2], leftward=[0])
    for change in (dict(start_node=[3]), dict(start_offset=[5]), dict(query_offset=[7]), dict(query_offset=[6]), dict(query_offset=[0], leftward=[1]), dict(flags=[capi.VGK_GSSW_LOCAL])):
        f = dict(ok, **change)
        es = capi.ExtensionSet(rd, [0, 6], **f)
        with pytest.raises(capi.VgkError):
            ora.align_extensions(g, es)
    ora.align_extensions(g, capi.ExtensionSet(rd, [0, 6], **ok))


@pytest.mark.gpu
def test_hip_extension_windows_three_ways():
    rw, es = three_ways(ENGINE_LIB, 15, 3000, 20000)
    assert (rw["score"] > 10).mean() > 0.3
    three_ways(ENGINE_LIB, 16, 600, 4000, capi.Scoring.simple(2, 3, 5, 2, 7))
