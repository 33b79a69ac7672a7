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
def test_chain_stage_on_the_gpu_with_fallbacks_equals_the_oracles():
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
    fallbacks_picked_from_flat_connects(emu_lib, 8, 5000, 9)


@pytest.mark.gpu
def test_fallback_batch_picked_from_flat_connects_on_the_gpu():
    fallbacks_picked_from_flat_connects(ENGINE_LIB, 40, 15000, 10)
