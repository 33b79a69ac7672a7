"""A paired-end slice of configs[3] on one device (vg_amd/pipeline.py: paired_stage; vg_amd/host/rescue_stage.cpp = the rescue half,
MinimizerMapper::attempt_rescue, src/minimizer_mapper.cpp:3264-3440): both mates through seeding / extension / tails, the mate without a
full-length extension rescued from its partner's position by the seeded two-pass X-drop + fix-ups.  The engine's stage (emulated kernels
here, HIP under -m gpu) must give the oracle stage's per-pair scores and rescued alignments, and rescue must actually recover the hard
mates: a rescued mate lies where the pair was sampled from."""
import os
import subprocess

import numpy as np
import pytest

from util import ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, pipeline, workloads

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvgamd_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def run(lib, wl, device, resident=False, request_table=None):
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    graph = (wl.node_len, wl.seq)
    index = eng.haplo_index(graph, wl.threads); mindex = eng.minimizer_index(graph, wl.threads)
    aligner = pipeline.HostAlignerHandle(lib)
    olen = np.repeat(wl.node_len, 2)
    rg = aligner.rescue_graph(wl) if resident else None
    timing = {}
    out = pipeline.paired_stage(eng, index, mindex, wl, aligner, oriented_len=olen, device=device, resident=rg, want_ops=True, request_table=request_table, timing=timing)
    out["timing_keys"] = set(timing)
    if rg is not None:
        rg.close()
    aligner.close()
    return out


def same_rescues(got, want, what):
    assert (got["rescued"] == want["rescued"]).all() and (got["requests"] == want["requests"]).all()
    bad = np.nonzero((got["rescue"] != want["rescue"]).any(axis=1))[0]
    assert len(bad) == 0, "%s: rescue %s: %s vs %s (request %s)" % (what, bad[:5], got["rescue"][bad[:5]], want["rescue"][bad[:5]], got["requests"][bad[:5]])
    assert (got["rescue_ops_begin"] == want["rescue_ops_begin"]).all(), what
    assert (got["rescue_ops"].view(np.uint64) == want["rescue_ops"].view(np.uint64)).all(), what
    assert (got["pair_score"] == want["pair_score"]).all()


def check(lib, n_pairs, device):
    wl = workloads.PairedWorkload(n_pairs, ref_len=300_000, seed=5, hard=0.25)
    got = run(lib, wl, device)
    want = run(ORACLE_LIB, wl, False)
    assert (got["read_score"] == want["read_score"]).all()
    same_rescues(got, want, "reference-shaped path on the engine vs on the oracle")
    # the rescue half on the RESIDENT graph (rescue_resident.cpp: extension windows, flat fix-ups): every rescued alignment, op by op — on the
    # engine under test and over the oracle's own extension windows
    res_got = run(lib, wl, device, resident=True)
    same_rescues(res_got, want, "resident path on the engine vs the reference-shaped path on the oracle")
    # (that run took its request table from vgk_rescue_requests — one lane per pair over the sets in HBM; the same with the table made by host
    # threads over the fetched sets, vg_amd/host/rescue_requests.cpp)
    if device:
        assert "rescue requests (device table + the mates' reads)" in res_got["timing_keys"]
        host_table = run(lib, wl, device, resident=True, request_table="host")
        assert "rescue requests (host threads)" in host_table["timing_keys"]
        same_rescues(host_table, want, "resident path, request table on host threads, vs the reference-shaped path on the oracle")
    same_rescues(run(ORACLE_LIB, wl, False, resident=True), want, "resident path on the oracle vs the reference-shaped path on the oracle")
    c = res_got["rescue_counts"]
    assert c["first_pass"] + c["scans"] > 0 and c["second_pass"] > 0
    assert len(got["rescue_ops"]) > 3 * len(got["rescued"]) * 0.5
    # rescue does its job: most pairs with a hard second mate are rescued, to a positive score, near where the mate came from
    hard = set(int(i) for i in wl.truth["hard"])
    rescued_pairs = set(int(r) // 2 for r in got["rescued"])
    assert len(hard & rescued_pairs) > 0.6 * len(hard)
    ok = got["rescue"][:, 1] == 0
    assert ok.mean() > 0.95 and (got["rescue"][ok, 0] > 60).mean() > 0.8
    # the rescued alignment starts inside its rescue subgraph
    fn = got["rescue"][ok, 2]; lo = got["requests"][ok, 0]; hi = got["requests"][ok, 1]
    has = fn >= 0
    assert ((fn[has] >= lo[has]) & (fn[has] < hi[has])).all()
    return got


def test_paired_stage_on_the_emulated_kernels(emu_lib):
    got = check(emu_lib, 300, True)
    assert len(got["rescued"]) > 30


@pytest.mark.gpu
def test_paired_stage_on_hip():
    check(ENGINE_LIB, 4000, True)
