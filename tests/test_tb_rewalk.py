"""The traceback that does not tax the fill (vg_amd/csrc/gssw_device.hpp, TB_REWALK): the fill keeps boundary rows and checkpoints instead of
4-bit codes, the codes are computed again in a band around the end cell's diagonal by the fill's own lane code, a walk that leaves its band
is redone by the on-demand form.  Results must be those of the stored-codes form and of the oracle bit for bit, on chains (where the band
serves nearly every read), on SNP / indel bubbles and on random DAGs and trees (where many walks leave the band: VGAMD_TB_REWALK=1 forces the
mode that the packers would not choose there), in every rows-per-lane geometry, all three modes, score-only problems beside them."""
import os
import subprocess

import numpy as np
import pytest

from gen import BASES, problem_set, random_problem
from test_gssw_wide import bubble_chain_problem
from util import ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvgamd_emu.so")
MODES = (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED)


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


class env:
    def __init__(self, **kv): self.kv = kv
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def same(ra, oa, rb, ob, problems, what):
    for i in range(len(problems)):
        ctx = "%s, problem %d" % (what, i)
        assert ra["status"][i] == rb["status"][i] and ra["score"][i] == rb["score"][i], ctx
        if ra["status"][i] != 0 or ra["score"][i] <= 0:
            continue
        for f in ("end_node", "end_offset", "end_read"):
            assert ra[f][i] == rb[f][i], (f, ctx)
        if problems[i]["flags"] & capi.VGK_GSSW_TRACEBACK:
            assert capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob), ctx


def three_ways(lib, problems, scoring=None):
    """forced TB_REWALK = forced TB_CODES = the oracle"""
    ps = problem_set(problems); sc = scoring or capi.Scoring.simple()
    with env(VGAMD_TB_REWALK="1", VGAMD_TB_CODES=None):
        ra, oa = capi.Engine(sc, lib=lib).align(ps)
    with env(VGAMD_TB_CODES="1", VGAMD_TB_REWALK=None):
        rc, oc = capi.Engine(sc, lib=lib).align(ps)
    rb, ob = capi.Engine(sc, lib=ORACLE_LIB).align(ps)
    same(ra, oa, rb, ob, problems, "rewalk vs oracle")
    same(rc, oc, rb, ob, problems, "codes vs oracle")
    return ra


def chain_problem(rng, mode, read_len, n_nodes, node_len, sub=0.04, indel=0.02, traceback=True):
    nodes = ["".join(BASES[i] for i in rng.integers(0, 4, int(rng.integers(max(1, node_len // 2), node_len + 1)))) for _ in range(n_nodes)]
    preds = [[]] + [[v - 1] for v in range(1, n_nodes)]
    ref = "".join(nodes)
    start = 0 if mode == capi.VGK_XDROP_PINNED else int(rng.integers(0, max(1, len(ref) - read_len)))
    out = []
    for c in ref[start:start + read_len + 20]:
        r = rng.random()
        if r < sub: out.append(BASES[int(rng.integers(0, 4))])
        elif r < sub + indel / 2: continue
        elif r < sub + indel: out.append(BASES[int(rng.integers(0, 4))]); out.append(c)
        else: out.append(c)
    read = "".join(out)[:read_len] or "A"
    p = {"read": read, "nodes": nodes, "preds": preds, "flags": mode | (capi.VGK_GSSW_TRACEBACK if traceback else 0), "pinning": None}
    if mode == capi.VGK_XDROP_PINNED: p["max_gap"] = 40
    if mode == capi.VGK_GSSW_PINNED: p["pinning"] = [0] * (n_nodes - 1) + [1]
    return p


def chains_and_bubbles(lib):
    rng = np.random.default_rng(11)
    problems = [chain_problem(rng, m, int(rng.integers(20, 260)), int(rng.integers(1, 20)), 32) for m in MODES * 60]
    problems += [chain_problem(rng, capi.VGK_GSSW_LOCAL, 150, 13, 32, indel=0.15) for _ in range(40)]      # gaps: walks that drift out of their band
    problems += [chain_problem(rng, capi.VGK_GSSW_LOCAL, 150, 13, 32, traceback=False) for _ in range(10)]
    problems += [bubble_chain_problem(rng, m, int(rng.integers(60, 400)), 12, 40) for m in MODES * 15]
    # the packers' own choice (stored codes unless asked otherwise)
    ps = problem_set(problems); sc = capi.Scoring.simple()
    with env(VGAMD_TB_REWALK=None, VGAMD_TB_CODES=None):
        ra, oa = capi.Engine(sc, lib=lib).align(ps)
    rb, ob = capi.Engine(sc, lib=ORACLE_LIB).align(ps)
    same(ra, oa, rb, ob, problems, "default mode vs oracle")
    res = three_ways(lib, problems)
    assert (res["score"] > 50).sum() > 150


def dags_in_every_geometry(lib):
    rng = np.random.default_rng(12)
    problems = [random_problem(rng, max_nodes=12, max_node_len=24, max_read=200, with_n=0.05, mode=m) for m in MODES * 40]
    problems += [random_problem(rng, traceback=False) for _ in range(20)]
    try:
        for k in (16, 19, 20, 24):
            os.environ["VGAMD_ROWS_PER_LANE"] = str(k)
            three_ways(lib, problems)
        three_ways(lib, problems[:60], capi.Scoring.simple(3, 5, 7, 2, 9))      # scores too large for the x8 build
    finally:
        os.environ.pop("VGAMD_ROWS_PER_LANE", None)


def long_reads_many_checkpoints(lib):
    rng = np.random.default_rng(13)
    problems = [chain_problem(rng, m, int(rng.integers(600, 1024)), 40, 32, sub=0.03, indel=0.01) for m in MODES * 4]
    problems += [random_problem(rng, max_nodes=30, max_node_len=60, max_read=900, mode=m) for m in MODES * 4]
    three_ways(lib, problems)


def test_emulated_chains_and_bubbles(emu_lib):
    chains_and_bubbles(emu_lib)


def test_emulated_dags_in_every_geometry(emu_lib):
    dags_in_every_geometry(emu_lib)


def test_emulated_long_reads(emu_lib):
    long_reads_many_checkpoints(emu_lib)


@pytest.mark.gpu
def test_hip_chains_and_bubbles():
    chains_and_bubbles(ENGINE_LIB)


@pytest.mark.gpu
def test_hip_dags_in_every_geometry():
    dags_in_every_geometry(ENGINE_LIB)


@pytest.mark.gpu
def test_hip_long_reads():
    long_reads_many_checkpoints(ENGINE_LIB)
