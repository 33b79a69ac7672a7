"""configs[4] as reads: a long read cut at its anchors, every stretch through WFAExtender, what it gives up on through
BandedGlobalAligner between the two anchors (vg_amd/pipeline.py chain_stage).  The engine's WFA declines problems its tables cannot
hold (VGK_ETOOBIG) and the oracle's does not: both routes must reach the same optimal score for every stretch."""
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, pipeline, workloads


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def run(lib, n_reads, read_len, seed, sv):
    wl = workloads.LongReadWorkload(n_reads, seed=seed, graph_bp=150_000, read_len=read_len, sv_fraction=sv)
    outs = []
    for which in (lib, ORACLE_LIB):
        eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=which)
        outs.append(pipeline.chain_stage(eng, eng.haplo_index(wl.nodes, wl.threads), wl))
    a, b = outs
    both = (a["wfa"]["status"] == 0) & (b["wfa"]["status"] == 0)
    for f in ("ok", "score", "length", "n_edits", "path_len"):
        assert (a["wfa"][f][both] == b["wfa"][f][both]).all(), f
    assert (a["segment_score"] == b["segment_score"]).all()
    assert (a["chain_score"] == b["chain_score"]).all()
    # the reads follow a haplotype: nearly every base scores
    assert (a["chain_score"] >= 0.93 * (wl.anchor_bases + np.bincount(wl.read_of, weights=np.diff(wl.ws.seq_off), minlength=n_reads))).all()
    return wl, a


def test_chain_stage_equals_the_oracles(emu_lib):
    wl, a = run(emu_lib, 5, 2500, 4, 0.0)
    assert wl.n > 40 and (a["wfa"]["ok"] != 0).mean() > 0.95
    modes = wl.ws.array["mode"]
    assert (modes == capi.WFA_PREFIX).sum() == 5 and (modes == capi.WFA_SUFFIX).sum() == 5


@pytest.mark.gpu
def test_chain_stage_on_the_gpu_with_fallbacks_equals_the_oracles(monkeypatch):
    monkeypatch.setenv("VGAMD_WFA_LARGE_POINTS", "16384")       # (round 4's table size: links that outgrow it take the fallback this test is about)
    wl, a = run(ENGINE_LIB, 60, 15000, 5, 0.02)
    assert len(a["failed"]) > 10 and (a["banded"]["status"] == 0).all()


def fallbacks_picked_from_flat_connects(lib, n_reads, read_len, seed):
    """chain_stage with the connects' subgraphs kept flat (LongReadWorkload.prepare_connects + BandedSet.select) = chain_stage assembling the
    fallback batch problem by problem; a small WFA point budget makes sure there ARE fallbacks"""
    outs = []
    for flat in (False, True):
        wl = workloads.LongReadWorkload(n_reads, seed=seed, graph_bp=150_000, read_len=read_len, sv_fraction=0.05)
        if flat:
            wl.prepare_connects()
        eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
        eng.wfa_set_point_budget(64)
        outs.append(pipeline.chain_stage(eng, eng.haplo_index(wl.nodes, wl.threads), wl))
    a, b = outs
    assert len(a["failed"]) >= 3 and (a["failed"] == b["failed"]).all()
    assert a["banded"].tobytes() == b["banded"].tobytes() and a["banded_ops"].tobytes() == b["banded_ops"].tobytes()
    assert (a["chain_score"] == b["chain_score"]).all()


def test_fallback_batch_picked_from_flat_connects(emu_lib):
    fallbacks_picked_from_flat_connects(emu_lib, 5, 5000, 9)


@pytest.mark.gpu
def test_fallback_batch_picked_from_flat_connects_on_the_gpu():
    fallbacks_picked_from_flat_connects(ENGINE_LIB, 40, 15000, 10)


# ---- the stage in the host shim (vg_amd/host/chain_stage.cpp): one call, the local graphs extracted inside it -------------------------
def native_stage(lib, n_reads, read_len, seed, sv, budgets=None, threads=3):
    """ChainStage (C++: WFA, then align_sequence_between_consistently for what WFA declines) on `lib` and on the oracle -> both outputs"""
    wl = workloads.LongReadWorkload(n_reads, seed=seed, graph_bp=150_000, read_len=read_len, sv_fraction=sv)
    outs = []
    for which in (lib, ORACLE_LIB):
        cs = pipeline.ChainStage(wl, lib=which)
        if budgets and which != ORACLE_LIB:
            cs.set_point_budgets(*budgets)
        outs.append(cs.run(threads=threads))
        cs.close()
    return wl, outs[0], outs[1]


def check_native(wl, a, b, min_declined):
    assert a["stats"]["failed"] == 0 and a["stats"]["no_graph"] == 0 and a["stats"]["too_big"] == 0, a["stats"]
    assert a["stats"]["declined"] >= min_declined and a["stats"]["between"] == a["stats"]["declined"]
    # a link either WFA or the graph between its anchors answers; where both engines took the WFA route the scores are equal; a link that took
    # the DP route scores at least what the haplotype-bound WFA of the oracle found for it (the local graph holds every haplotype's walk)
    same_route = a["link_source"] == b["link_source"]
    assert (a["link_score"][same_route] == b["link_score"][same_route]).all()
    assert (a["link_score"][~same_route] >= b["link_score"][~same_route]).all()
    assert (a["chain_score"] >= b["chain_score"]).all()
    links = np.bincount(wl.read_of, weights=np.diff(wl.ws.seq_off), minlength=wl.n_reads)
    assert (a["chain_score"] >= 0.8 * (wl.anchor_bases + links)).all()              # (a fifth of the connects may carry a 25-60 bp insertion)


def test_native_chain_stage_equals_the_oracles(emu_lib):
    wl, a, b = native_stage(emu_lib, 5, 2500, 4, 0.0)
    assert (a["chain_score"] == b["chain_score"]).all() and (a["link_score"] == b["link_score"]).all()


def test_native_chain_stage_with_declined_links(emu_lib):
    wl, a, b = native_stage(emu_lib, 4, 2500, 6, 0.2, budgets=(48, 48))
    check_native(wl, a, b, 4)
    assert (a["link_source"][wl.ws.array["mode"] != capi.WFA_CONNECT] != 2).all()         # declined tails went through pinned X-drop, none is left unaligned


def test_native_chain_stage_equals_the_python_pipeline(emu_lib):
    """the same reads through round 2's Python glue (its own stand-in for the graph between two anchors: a run of the topological order)"""
    wl, a, _ = native_stage(emu_lib, 5, 2500, 4, 0.1, budgets=(48, 48))          # (a point budget, so that there are declined links at all)
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=emu_lib)
    eng.wfa_set_point_budget(48)
    old = pipeline.chain_stage(eng, eng.haplo_index(wl.nodes, wl.threads), wl)
    assert len(old["failed"]) == a["stats"]["declined"] > 0
    assert (old["chain_score"] == a["chain_score"]).all()


@pytest.mark.gpu
def test_native_chain_stage_on_the_gpu(monkeypatch):
    # with the tables as shipped (262 144 points per link) no link is declined for its points: the stage's chain scores are the oracle stage's
    wl, a, b = native_stage(ENGINE_LIB, 60, 15000, 5, 0.02, threads=8)
    check_native(wl, a, b, 0)
    shipped = a["stats"]["declined"]
    assert ((a["chain_score"] == b["chain_score"]) | (np.bincount(wl.read_of, weights=(a["link_source"] != b["link_source"]), minlength=wl.n_reads) > 0)).all()
    # with round 4's table size links with long insertions outgrow it and take the DP route
    monkeypatch.setenv("VGAMD_WFA_LARGE_POINTS", "16384")
    wl, a, b = native_stage(ENGINE_LIB, 60, 15000, 5, 0.02, threads=8)
    check_native(wl, a, b, 10)
    assert a["stats"]["declined"] > shipped
    monkeypatch.delenv("VGAMD_WFA_LARGE_POINTS")
    wl, a, b = native_stage(ENGINE_LIB, 40, 15000, 11, 0.02, budgets=(64, 64), threads=8)
    check_native(wl, a, b, 40)
