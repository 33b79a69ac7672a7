"""The wide route of vgk_gssw_align (vg_amd/csrc/gssw_wide_device.hpp): problems outside the packed kernels' range — reads of more than
1024 rows (dozeu has no such limit: the reference's own "can align a long tail" is a 4.4 kbp tail, src/unittest/minimizer_mapper.cpp:682-709)
and scorings whose scores leave 11 bits — must give the oracle's results bit for bit: status, score, end cell, every CIGAR element.
On the CPU the identical lane code runs under the lock-step emulator (tests/emu); on the GPU the HIP kernels."""
import os
import subprocess

import numpy as np
import pytest

from gen import BASES, problem_set, random_dag, random_problem
from util import ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvgamd_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def long_problem(rng, mode, read_len, n_nodes, max_node_len, with_n=0.0, traceback=True, sub=0.06, indel=0.03):
    nodes, preds = random_dag(rng, n_nodes, max_node_len, p_chain=0.85, with_n=with_n)
    # the read follows a walk from the start of node 0 (left-pinned tails start at a source; the others may as well) with substitutions
    # and indels; where the walk runs out of graph the rest is random (an overhanging tail)
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for q in pr:
            succ[q].append(v)
    v, ref = 0, []
    while True:
        ref.append(nodes[v])
        if not succ[v]:
            break
        v = succ[v][0] if rng.random() < 0.7 else succ[v][int(rng.integers(0, len(succ[v])))]
    out = []
    for c in "".join(ref):
        r = rng.random()
        if r < sub:
            out.append(BASES[int(rng.integers(0, 4))])
        elif r < sub + indel / 2:
            continue
        elif r < sub + indel:
            out.append(BASES[int(rng.integers(0, 4))]); out.append(c)
        else:
            out.append(c)
    skip = 0 if mode == capi.VGK_XDROP_PINNED else int(rng.integers(0, 30))
    read = "".join(out)[skip:skip + read_len]
    if len(read) < read_len:                      # the walk ran out of graph: random bases to the wanted length (an overhanging tail)
        read += "".join(BASES[i] for i in rng.integers(0, 4, read_len - len(read)))
    if with_n and rng.random() < 0.5:
        k = int(rng.integers(0, len(read))); read = read[:k] + "N" + read[k + 1:]
    flags = mode | (capi.VGK_GSSW_TRACEBACK if traceback else 0)
    p = {"read": read, "nodes": nodes, "preds": preds, "flags": flags, "pinning": None}
    if mode == capi.VGK_XDROP_PINNED:
        p["max_gap"] = int(rng.integers(0, 80))
    if mode == capi.VGK_GSSW_PINNED:
        has_succ = [False] * n_nodes
        for v, pr in enumerate(preds):
            for q in pr:
                has_succ[q] = True
        p["pinning"] = [0 if h else 1 for h in has_succ]
    return p


def bubble_chain_problem(rng, mode, read_len, n_sites, seg_len, sub=0.04, indel=0.02):
    """A chain of segments with a SNP / indel bubble between consecutive ones (what a variation graph looks like), the read a walk from
    the start through random alleles: alignments that cross every strip boundary and many non-chain node boundaries."""
    nodes, preds, ref = [], [], []
    last = []
    for k in range(n_sites):
        seg = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(seg_len // 2, seg_len + 1))))
        nodes.append(seg); preds.append(list(last)); ref.append(seg)
        s = len(nodes) - 1
        a1 = BASES[int(rng.integers(0, 4))]; a2 = "".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(1, 4))))
        nodes.append(a1); preds.append([s]); nodes.append(a2); preds.append([s])
        last = [s + 1, s + 2] if rng.random() < 0.8 else [s + 2, s + 1, s]       # sometimes the alleles can be skipped (a deletion edge)
        ref.append(a1 if rng.random() < 0.5 else a2)
    out = []
    for c in "".join(ref):
        r = rng.random()
        if r < sub:
            out.append(BASES[int(rng.integers(0, 4))])
        elif r < sub + indel / 2:
            continue
        elif r < sub + indel:
            out.append(BASES[int(rng.integers(0, 4))]); out.append(c)
        else:
            out.append(c)
    read = "".join(out)[:read_len]
    p = {"read": read, "nodes": nodes, "preds": preds, "flags": mode | capi.VGK_GSSW_TRACEBACK, "pinning": None}
    if mode == capi.VGK_XDROP_PINNED:
        p["max_gap"] = 60
    if mode == capi.VGK_GSSW_PINNED:
        has_succ = [False] * len(nodes)
        for v, pr in enumerate(preds):
            for q in pr:
                has_succ[q] = True
        p["pinning"] = [0 if h else 1 for h in has_succ]
    return p


def compare(lib_a, problems, scoring=None, qual_adj=None):
    ps = problem_set(problems)
    sc = scoring or capi.Scoring.simple()
    ra, oa = capi.Engine(sc, lib=lib_a, qual_adj=qual_adj).align_call(ps)
    rb, ob = capi.Engine(sc, lib=ORACLE_LIB, qual_adj=qual_adj).align(ps)
    for i in range(ps.n):
        ctx = "problem %d: read %d bases, %d nodes, flags %d" % (i, len(problems[i]["read"]), len(problems[i]["nodes"]), problems[i]["flags"])
        assert ra["status"][i] == rb["status"][i], ctx
        assert ra["score"][i] == rb["score"][i], ctx
        if ra["status"][i] != 0 or ra["score"][i] <= 0:
            continue
        for f in ("end_node", "end_offset", "end_read"):
            assert ra[f][i] == rb[f][i], (f, ctx)
        if problems[i]["flags"] & capi.VGK_GSSW_TRACEBACK:
            assert capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob), ctx
    return ra


MODES = (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED)


def long_reads(lib, n_each, rng_seed, lens=(1025, 2300)):
    rng = np.random.default_rng(rng_seed)
    problems = []
    for mode in MODES:
        for _ in range(n_each):
            L = int(rng.integers(lens[0], lens[1]))
            problems.append(long_problem(rng, mode, L, int(rng.integers(20, 60)), int(rng.integers(40, 160)), with_n=0.05))
    problems += [long_problem(rng, capi.VGK_XDROP_PINNED, 1500, 12, 200, traceback=False)]
    # short problems beside them in the same call: the packed kernels take those
    problems += [random_problem(rng, mode=m) for m in MODES * 5]
    order = rng.permutation(len(problems))
    problems = [problems[k] for k in order]
    problems += [bubble_chain_problem(rng, m, int(rng.integers(1025, 2000)), 45, 60) for m in MODES]
    res = compare(lib, problems)
    assert (res["score"] > 200).sum() >= n_each


def several_strips(lib):
    """Reads of more than 4096 rows run in strips of 256 lanes x 16 rows, the last row of a strip carried through HBM; 2049-4096 rows
    are one strip of 16 rows per lane."""
    rng = np.random.default_rng(20260925)
    problems = [long_problem(rng, m, L, max(60, L // 40), 120) for m in MODES for L in (2049, 4096, 4097, 4500, 9000)]
    problems.append(long_problem(rng, capi.VGK_XDROP_PINNED, 5000, 6, 900, sub=0.02, indel=0.01))     # long nodes, a clean tail
    problems.append(long_problem(rng, capi.VGK_GSSW_LOCAL, 4300, 1, 3000))                           # one node
    problems += [bubble_chain_problem(rng, m, 5600, 110, 70) for m in MODES]                       # two strips, alignments across the boundary
    problems.append(bubble_chain_problem(rng, capi.VGK_XDROP_PINNED, 9500, 220, 60, sub=0.02, indel=0.01))   # three strips
    res = compare(lib, problems)
    assert (res["score"] > 3000).sum() >= 4 and (res["score"] > 4200).sum() >= 1


def wide_scores(lib):
    """Short reads whose scores do not fit the packed kernels' 11 bits (VGK_EUNSUPPORTED of vgk_gssw_pack) take the wide route too;
    past gssw's / dozeu's own int16 limit both sides say VGK_EOVERFLOW."""
    rng = np.random.default_rng(77)
    sc = capi.Scoring.simple(20, 9, 12, 3, 10)
    problems = [random_problem(rng, mode=m, max_read=400, max_nodes=20, max_node_len=40) for m in MODES * 40]
    res = compare(lib, problems, scoring=sc)
    assert (res["score"] > 2047).sum() > 10
    big = capi.Scoring.simple(100, 9, 12, 3, 10)
    nodes = ["ACGT" * 100]
    p = {"read": "ACGT" * 100, "nodes": nodes, "preds": [[]], "flags": capi.VGK_GSSW_LOCAL | capi.VGK_GSSW_TRACEBACK, "pinning": None}
    ra, _ = capi.Engine(big, lib=lib).align_call(problem_set([p]))
    rb, _ = capi.Engine(big, lib=ORACLE_LIB).align(problem_set([p]))
    assert ra["status"][0] == rb["status"][0] == capi.VGK_EOVERFLOW


def quality_adjusted(lib):
    from qualadj import qual_adj_tables
    tables = qual_adj_tables(1, 4, 5)
    rng = np.random.default_rng(5)
    problems = []
    for mode in MODES * 3:
        p = long_problem(rng, mode, int(rng.integers(1100, 1600)), 30, 120, with_n=0.05)
        p["qual"] = rng.choice(np.array([2, 5, 10, 20, 30, 40], dtype=np.uint8), size=len(p["read"]))
        problems.append(p)
    compare(lib, problems, qual_adj=tables)


def test_emulated_wide_long_reads(emu_lib):
    long_reads(emu_lib, 4, 1)


def test_emulated_wide_several_strips(emu_lib):
    several_strips(emu_lib)


def test_emulated_wide_scores(emu_lib):
    wide_scores(emu_lib)


def test_emulated_wide_quality_adjusted(emu_lib):
    quality_adjusted(emu_lib)


@pytest.mark.gpu
def test_hip_wide_long_reads():
    long_reads(ENGINE_LIB, 30, 2)
    long_reads(ENGINE_LIB, 6, 3, lens=(2300, 5200))


@pytest.mark.gpu
def test_hip_wide_several_strips():
    several_strips(ENGINE_LIB)


@pytest.mark.gpu
def test_hip_wide_scores_and_quality_adjusted():
    wide_scores(ENGINE_LIB)
    quality_adjusted(ENGINE_LIB)
