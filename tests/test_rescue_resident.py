"""The rescue stage on the RESIDENT graph (vg_amd/host/rescue_resident.cpp: extension windows + flat fix-ups) against the reference-shaped one
(vg_amd/host/rescue_stage.cpp: one HashGraph, one Alignment, Aligner::align_xdrop_many, fix_dozeu_score, fix_dozeu_end_deletions per mate —
MinimizerMapper::attempt_rescue, src/minimizer_mapper.cpp:3264-3440) on synthetic requests that reach the corners the paired workload does not:
mates that are noise (dozeu's scan fails, or the rescored alignment is not worth keeping: the full DP), seeds at node ends, seeds at either end
of the read, windows of one node, N bases, requests the cell budget refuses, empty node ranges.  Every answer and every op run must agree."""
import ctypes
import subprocess

import numpy as np
import pytest

from gen import random_dag
from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, pipeline


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu", "host"], cwd=ROOT)
    return EMU_LIB


class Graph:
    def __init__(self, nodes, preds):
        self.nodes = nodes; self.preds = preds; self.n_nodes = len(nodes)
        self.node_len = np.array([len(s) for s in nodes], dtype=np.uint32)
        self.seq = np.frombuffer("".join(nodes).encode(), dtype=np.uint8).copy()
        self.col = np.concatenate([[0], np.cumsum(self.node_len)]).astype(np.int64)
        self.pred_off = np.concatenate([[0], np.cumsum([len(p) for p in preds])]).astype(np.uint32)
        self.pred_idx = np.array([q for p in preds for q in p] or [0], dtype=np.uint32)
        succ = [[] for _ in nodes]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        self.succ = succ
        self.succ_off = np.concatenate([[0], np.cumsum([len(x) for x in succ])]).astype(np.uint32)
        self.succ_idx = np.array([w for x in succ for w in x] or [0], dtype=np.uint32)


def random_requests(rng, g, n, read_len_range=(30, 150)):
    reads, read_off, req = [], [0], []
    for _ in range(n):
        lo = int(rng.integers(0, g.n_nodes - 1)); hi = min(g.n_nodes, lo + int(rng.integers(1, 40)))
        L = int(rng.integers(*read_len_range))
        kind = rng.random()
        # a noisy walk from somewhere in the window (most mates), or noise (a mate that does not belong there)
        v = int(rng.integers(lo, hi)); off = int(rng.integers(0, len(g.nodes[v])))
        walk = []            # (node, offset) of every read base that came from the graph, None for inserted bases
        out = []
        if kind < 0.8:
            err = 0.02 if kind < 0.6 else 0.15
            while len(out) < L:
                if off >= len(g.nodes[v]):
                    nx = [w for w in g.succ[v] if w < hi]
                    if not nx:
                        break
                    v = nx[int(rng.integers(0, len(nx)))]; off = 0; continue
                c = g.nodes[v][off]; pos = (v, off); off += 1
                r = rng.random()
                if r < err:
                    c = "ACGT"[int(rng.integers(0, 4))]
                elif r < err * 1.3:
                    continue
                elif r < err * 1.6:
                    out.append("ACGT"[int(rng.integers(0, 4))]); walk.append(None)
                out.append(c); walk.append(pos)
        lead = int(rng.integers(0, 8)) if rng.random() < 0.3 else 0
        out = ["ACGT"[int(x)] for x in rng.integers(0, 4, lead)] + out; walk = [None] * lead + walk
        while len(out) < L:
            out.append("ACGTN"[int(rng.integers(0, 5))]); walk.append(None)
        out, walk = out[:L], walk[:L]
        # dozeu's seed: a stretch of the walk (its first base's graph position), sometimes at the read's very start / end, sometimes none
        seed = (0, 0, -1, 0)
        cand = [i for i, w in enumerate(walk) if w is not None]
        if cand and rng.random() < 0.75:
            pick = rng.random()
            i = cand[0] if pick < 0.15 else cand[-1] if pick < 0.3 else cand[int(rng.integers(0, len(cand)))]
            ln = int(rng.integers(1, 30))
            seed = (i, min(L, i + ln), walk[i][0], walk[i][1])
        if rng.random() < 0.03:
            lo, hi = hi, lo                                                   # an empty range
        reads.append(np.frombuffer("".join(out).encode(), dtype=np.uint8)); read_off.append(read_off[-1] + L)
        req.append((lo, hi) + seed)
    return np.concatenate(reads), np.array(read_off, dtype=np.uint64), np.array(req, dtype=np.int64)


def both_paths(lib, g, reads, read_off, req, max_cells=0, threads=2):
    h = pipeline._host_lib()
    n = len(req)
    outs = []
    for resident in (False, True):
        aligner = pipeline.HostAlignerHandle(lib)
        out = np.zeros((n, 6), dtype=np.int64); ops_begin = np.zeros(n + 1, dtype=np.uint64); cap = 64 * n + 64
        ops = np.zeros(cap, dtype=capi.OP_DT); written = ctypes.c_uint64()
        if resident:
            h.vgh_rescue_graph_create.restype = ctypes.c_void_p
            h.vgh_rescue_graph_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            h.vgh_rescue_graph_destroy.argtypes = [ctypes.c_void_p]
            rg = h.vgh_rescue_graph_create(aligner.ptr, g.n_nodes, g.node_len.ctypes.data, g.seq.ctypes.data, g.pred_off.ctypes.data, g.pred_idx.ctypes.data)
            assert rg, h.vgh_last_error()
            h.vgh_rescue_stage_resident.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            counts = np.zeros(6, dtype=np.uint64)
            rc = h.vgh_rescue_stage_resident(aligner.ptr, rg, n, reads.ctypes.data, reads.size, read_off.ctypes.data, req.ctypes.data, max_cells, threads, out.ctypes.data,
                                             ops_begin.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written), None, counts.ctypes.data)
            assert rc == 0, h.vgh_last_error()
            h.vgh_rescue_graph_destroy(rg)
            outs.append((out, ops_begin, ops[:written.value].copy(), counts))
        else:
            seq_off = np.ascontiguousarray(g.col[:-1], dtype=np.uint64)
            h.vgh_rescue_stage_ops.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            rc = h.vgh_rescue_stage_ops(aligner.ptr, g.n_nodes, g.node_len.ctypes.data, seq_off.ctypes.data, g.seq.ctypes.data, g.succ_off.ctypes.data, g.succ_idx.ctypes.data,
                                        n, reads.ctypes.data, read_off.ctypes.data, req.ctypes.data, max_cells, threads, out.ctypes.data,
                                        ops_begin.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written))
            assert rc == 0, h.vgh_last_error()
            outs.append((out, ops_begin, ops[:written.value].copy(), None))
        aligner.close()
    return outs


def same(a, b, req, what):
    bad = np.nonzero((a[0] != b[0]).any(axis=1))[0]
    assert len(bad) == 0, "%s: request %d %s: %s vs %s" % (what, bad[0], req[bad[0]], a[0][bad[0]], b[0][bad[0]])
    assert (a[1] == b[1]).all(), what
    diff = np.nonzero(a[2].view(np.uint64) != b[2].view(np.uint64))[0]
    assert len(diff) == 0, "%s: ops differ first at request %d" % (what, int(np.searchsorted(a[1], diff[0], side="right") - 1))


def corners(lib, seed, n_nodes, n, max_cells=0):
    rng = np.random.default_rng(seed)
    nodes, preds = random_dag(rng, n_nodes, 20, with_n=0.02)
    g = Graph(nodes, [sorted(p) for p in preds])
    reads, read_off, req = random_requests(rng, g, n)
    old, new = both_paths(lib, g, reads, read_off, req, max_cells=max_cells)
    same(new, old, req, "resident path vs reference-shaped path (%s)" % lib.split("/")[-1])
    return old, new, req


def test_resident_rescue_equals_the_reference_shaped_path_over_the_oracle():
    old, new, req = corners(ORACLE_LIB, 3, 400, 2500)
    c = new[3]
    assert c[1] > 200 and c[3] > 50, c                                       # scans and full-DP fallbacks both happen
    st = old[0][:, 1]
    assert (st == 2).sum() > 20 and (st == 0).sum() > 2000
    assert (old[0][:, 0] > 20).sum() > 400
    for threads in (1, 3):                                                   # (one host thread: the chunks one after the other)
        rng = np.random.default_rng(8)
        nodes, preds = random_dag(rng, 200, 20)
        g = Graph(nodes, [sorted(p) for p in preds])
        reads, read_off, req = random_requests(rng, g, 700)
        a, b = both_paths(ORACLE_LIB, g, reads, read_off, req, threads=threads)
        same(b, a, req, "resident vs reference-shaped, %d host threads" % threads)
    old, new, req = corners(ORACLE_LIB, 4, 300, 1200, max_cells=12000)       # a cell budget that refuses the larger windows
    assert (old[0][:, 1] == 1).sum() > 50 and (old[0][:, 1] == 0).sum() > 50


def test_resident_rescue_on_the_emulated_kernels(emu_lib):
    old, new, req = corners(emu_lib, 5, 300, 900)
    ora_old, ora_new, _ = corners(ORACLE_LIB, 5, 300, 900)
    same(new, ora_old, req, "resident path on the emulated kernels vs reference-shaped path on the oracle")


@pytest.mark.gpu
def test_resident_rescue_on_hip():
    old, new, req = corners(ENGINE_LIB, 6, 3000, 30000)
    ora_old, _, _ = corners(ORACLE_LIB, 6, 3000, 30000)
    same(new, ora_old, req, "resident path on HIP vs reference-shaped path on the oracle")
