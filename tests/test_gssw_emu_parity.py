"""CPU-only differential test of the HIP kernel *logic*: the lane code of
vg_amd/csrc/gssw_device.hpp + the packing layer vgk_api.cpp run under the
lock-step wavefront emulator (tests/emu) must agree bit for bit with the oracle
through the same C ABI.  The same comparison runs against the real HIP library
on the GPU in test_gssw_gpu_parity.py."""
import os
import subprocess

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


def compare(lib_a, lib_b, problems, scoring=None, ops_per=0):
    ps = problem_set(problems)
    sc = scoring or capi.Scoring.simple()
    ra, oa = capi.Engine(sc, lib=lib_a).align(ps, ops_per)
    rb, ob = capi.Engine(sc, lib=lib_b).align(ps, ops_per)
    for i in range(ps.n):
        ctx = "problem %d: %r" % (i, problems[i])
        assert ra["status"][i] == rb["status"][i], ctx
        assert ra["score"][i] == rb["score"][i], ctx
        if ra["status"][i] != 0 or ra["score"][i] <= 0:
            continue
        for f in ("end_node", "end_offset", "end_read"):
            assert ra[f][i] == rb[f][i], (f, ctx)
        if problems[i]["flags"] & capi.VGK_GSSW_TRACEBACK:
            assert capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob), ctx
    return ra


def test_emulated_kernel_matches_oracle_random(emu_lib):
    rng = np.random.default_rng(1234)
    problems = [random_problem(rng) for _ in range(600)]
    res = compare(emu_lib, ORACLE_LIB, problems)
    assert (res["score"] > 0).sum() > 400


def test_emulated_kernel_matches_oracle_with_n_and_score_only(emu_lib):
    rng = np.random.default_rng(99)
    problems = [random_problem(rng, with_n=0.3) for _ in range(200)]
    problems += [random_problem(rng, traceback=False) for _ in range(100)]
    compare(emu_lib, ORACLE_LIB, problems)


def test_emulated_kernel_matches_oracle_long_reads_and_scoring(emu_lib):
    rng = np.random.default_rng(7)
    problems = [random_problem(rng, max_nodes=20, max_node_len=40, max_read=400) for _ in range(40)]
    compare(emu_lib, ORACLE_LIB, problems, capi.Scoring.simple(2, 3, 5, 2, 7))
    problems = [random_problem(rng, max_nodes=6, max_node_len=8, max_read=40) for _ in range(100)]
    compare(emu_lib, ORACLE_LIB, problems, capi.Scoring.simple(1, 4, 6, 1, 0))


def test_emulated_xdrop_pinned_matches_oracle(emu_lib):
    rng = np.random.default_rng(4242)
    problems = [random_problem(rng, mode=capi.VGK_XDROP_PINNED) for _ in range(600)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED, with_n=0.2, max_read=300, max_node_len=40) for _ in range(100)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED, traceback=False) for _ in range(100)]
    res = compare(emu_lib, ORACLE_LIB, problems)
    assert (res["score"] > 0).sum() > 300
    # mixed batch: all three modes side by side in the same wavefronts
    mixed = [random_problem(rng, mode=m) for m in (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED) * 100]
    compare(emu_lib, ORACLE_LIB, mixed)


def test_emulated_quality_adjusted_contexts_match_oracle(emu_lib):
    from qualadj import qual_adj_tables
    tables = qual_adj_tables(1, 4, 5)
    assert tables[0][25 * 40 + 0] == 1 and tables[0][25 * 40 + 1] == -4 and tables[0][25 * 2 + 0] == 0   # Q40 ~ base scores, Q2 = 0
    rng = np.random.default_rng(31337)
    problems = []
    for mode in (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED) * 150:
        p = random_problem(rng, mode=mode, with_n=0.1)
        q = rng.choice(np.array([2, 5, 10, 20, 30, 40], dtype=np.uint8), size=len(p["read"]))
        p["qual"] = q
        problems.append(p)
    ps = problem_set(problems)
    sc = capi.Scoring.simple()
    ra, oa = capi.Engine(sc, lib=emu_lib, qual_adj=tables).align(ps)
    rb, ob = capi.Engine(sc, lib=ORACLE_LIB, qual_adj=tables).align(ps)
    rc, _ = capi.Engine(sc, lib=ORACLE_LIB).align(ps)                 # plain scoring must differ somewhere
    assert (rb["score"] != rc["score"]).sum() > 50
    for i in range(ps.n):
        assert ra["status"][i] == rb["status"][i] == 0 and ra["score"][i] == rb["score"][i], i
        if ra["score"][i] > 0:
            assert capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob), i


def every_lane_geometry(lib):
    """Each rows-per-lane instantiation (VGAMD_ROWS_PER_LANE forces one for every read) against the oracle, all three modes mixed;
    19 is the one whose last traceback dword holds three rows."""
    rng = np.random.default_rng(4242)
    problems = [random_problem(rng, max_nodes=10, max_node_len=24, max_read=160, with_n=0.05) for _ in range(120)]
    problems += [random_problem(rng, max_nodes=8, max_node_len=16, max_read=60, mode=capi.VGK_GSSW_PINNED) for _ in range(60)]
    problems += [random_problem(rng, max_nodes=8, max_node_len=16, max_read=60, mode=capi.VGK_XDROP_PINNED) for _ in range(60)]
    try:
        for k in (16, 19, 20, 24):
            os.environ["VGAMD_ROWS_PER_LANE"] = str(k)
            compare(lib, ORACLE_LIB, problems)
            compare(lib, ORACLE_LIB, problems[:60], capi.Scoring.simple(3, 5, 7, 2, 9))      # scores too large for the x8 build
    finally:
        os.environ.pop("VGAMD_ROWS_PER_LANE", None)


def test_emulated_kernel_matches_oracle_in_every_lane_geometry(emu_lib):
    every_lane_geometry(emu_lib)


# The tracebacks of a batch of local alignments run as two kernels (GsswParams::walk_passes): every read by diagonal runs alone — the scores
# along the diagonal subtracted from H at the run's end must arrive exactly at 0 or at a saved last-column H — then the reads that needed a
# code.  The same alignments as the oracle's and as the one-kernel walk's.
def test_two_pass_walk_matches_oracle_and_the_one_pass_walk(emu_lib, monkeypatch):
    rng = np.random.default_rng(77)
    problems = [random_problem(rng, mode=capi.VGK_GSSW_LOCAL, max_nodes=12, max_node_len=20, max_read=120, with_n=0.03) for _ in range(1300)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED) for _ in range(100)] + [random_problem(rng, mode=capi.VGK_GSSW_PINNED) for _ in range(100)]
    for sc in (None, capi.Scoring.simple(1, 1, 1, 1, 5), capi.Scoring.simple(2, 3, 5, 2, 0)):          # (equal mismatch and gap costs: ties at nearly every cell)
        two = compare(emu_lib, ORACLE_LIB, problems, sc)
        monkeypatch.setenv("VGAMD_WALK_ONE_PASS", "1")
        one = compare(emu_lib, ORACLE_LIB, problems, sc)
        monkeypatch.delenv("VGAMD_WALK_ONE_PASS")
        assert (two["score"] == one["score"]).all() and (two["n_ops"] == one["n_ops"]).all()


# The speculative fill (GsswParams::spec_fill): a batch of one geometry fills every read WITHOUT traceback codes, settles the alignments that
# are one diagonal run from their end cells, lays the rest out as wavefronts of their own, fills those again with codes and walks them.
def speculative_fill_equals_the_plain_one(lib, n, monkeypatch, match=1):
    from vg_amd import workloads
    wl = workloads.LinearWorkload(n, seed=43, sub_rate=0.02, indel_rate=0.004)      # uniform reads: one bucket; a third of them with an indel
    sc = capi.Scoring.simple(match, 4, 6, 1, 5)
    ro, oo = capi.Engine(sc, lib=ORACLE_LIB).align(wl, 0)
    eng = capi.Engine(sc, lib=lib)
    ra, oa = eng.align(wl, 0)                                                        # packed on the host (vgk_gssw_pack)
    graph = eng.graph(*wl.graph_arrays())
    with eng.pack_windows(graph, wl.windows(), 0) as b:                              # packed on the device (vgk_gssw_pack_windows)
        b.run(); b.sync(); rw, ow = b.fetch()
    monkeypatch.setenv("VGAMD_NO_SPEC_FILL", "1")
    rp, op = capi.Engine(sc, lib=lib).align(wl, 0)
    monkeypatch.delenv("VGAMD_NO_SPEC_FILL")
    for name, (r, o) in (("host-packed", (ra, oa)), ("device-packed", (rw, ow)), ("plain", (rp, op))):
        for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            assert (r[f] == ro[f]).all(), (name, f)
        for i in range(wl.n):
            assert capi.cigar_string(r[i], o) == capi.cigar_string(ro[i], oo), (name, i)
    return int((ro["score"] > 0).sum())


def test_speculative_fill_matches_oracle(emu_lib, monkeypatch):
    assert speculative_fill_equals_the_plain_one(emu_lib, 1200, monkeypatch) > 1100


# The first fill's column key maximum (gssw_device.hpp, K3; GsswParams::key3): one v_pk_maximum3_f16 per two rows while every key of the batch stays below
# 0x7c00, i.e. scores up to 990.  A match worth 6 takes 150-base reads to 910 (keys up to 0x71c0: the top of the range, three-input maximum taken); a match
# worth 7 to 1 060: the packers must leave it off, and the batch still speculates.
@pytest.mark.parametrize("match", [6, 7])
def test_speculative_fill_with_scores_at_the_key_maximums_limit(emu_lib, monkeypatch, match):
    assert speculative_fill_equals_the_plain_one(emu_lib, 1100, monkeypatch, match=match) > 1000


# The corners of the speculation: a batch in which NO read misses (the second fill covers nothing), one in which nearly every read does (the
# refilled wavefronts are as many as the batch's own), the smallest batch that speculates (1 024 reads; one fewer does not), and the same batch
# run three times (the miss list, the new wavefronts and their count are rebuilt by every run).  Under the emulator vgk_batch_kernel_ms(batch, 3) is the number of
# wavefronts laid out again so far.
@pytest.mark.parametrize("n,sub_rate,indel_rate", [(1100, 0.0, 0.0), (1100, 0.05, 0.05), (1024, 0.02, 0.004), (1023, 0.02, 0.004)])
def test_speculative_fill_corner_batches_and_reruns(emu_lib, n, sub_rate, indel_rate):
    from vg_amd import workloads
    wl = workloads.LinearWorkload(n, seed=5, sub_rate=sub_rate, indel_rate=indel_rate)
    sc = capi.Scoring.simple(1, 4, 6, 1, 5)
    ro, oo = capi.Engine(sc, lib=ORACLE_LIB).align(wl, 0)
    eng = capi.Engine(sc, lib=emu_lib)
    eng.set_speculation(1)                                             # (the mechanics, whatever the feedback would say after the first run)
    graph = eng.graph(*wl.graph_arrays())
    with eng.pack_windows(graph, wl.windows(), 0) as b:
        refilled = []
        for _ in range(3):
            b.run(); b.sync(); refilled.append(b.kernel_ms(3))
        r, o = b.fetch()
    steps = np.diff([0.0] + refilled)
    assert (steps == steps[0]).all()                                  # every run lays the same reads out again
    if indel_rate == 0.0:
        assert steps[0] <= 2                                           # exact copies: (nearly) every alignment is one diagonal run
    if n == 1023:
        assert steps[0] == 0                                           # below the threshold: one fill, with codes
    elif indel_rate > 0.0:
        assert steps[0] > 0
    if indel_rate >= 0.05:
        assert steps[0] >= n // 16 * 0.9                               # 150 bases at 5 % indels: hardly a read without one
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        assert (r[f] == ro[f]).all(), f
    for i in range(wl.n):
        assert capi.cigar_string(r[i], o) == capi.cigar_string(ro[i], oo), i


# Speculation with feedback (vgk_ctx::SpecPolicy): whether a batch that CAN speculate does is decided per run from the miss counts of the
# context's earlier speculative runs.  A stream of reads full of indels turns it off after the first batch, is probed after VGAMD_SPEC_PROBE_EVERY
# runs (the wait doubling while the probes fail), and a stream that turns clean turns it on again; every batch equals the oracle whichever way it ran.
def speculation_follows_the_miss_counts(emu_lib, monkeypatch, n=1100):
    from vg_amd import workloads
    monkeypatch.setenv("VGAMD_SPEC_PROBE_EVERY", "2")
    sc = capi.Scoring.simple(1, 4, 6, 1, 5)
    noisy = workloads.LinearWorkload(n, seed=5, sub_rate=0.05, indel_rate=0.05)
    clean = workloads.LinearWorkload(n, seed=6, sub_rate=0.0, indel_rate=0.0)
    ora = capi.Engine(sc, lib=ORACLE_LIB)
    want = {id(noisy): ora.align(noisy, 0), id(clean): ora.align(clean, 0)}
    eng = capi.Engine(sc, lib=emu_lib)
    graphs = {id(noisy): eng.graph(*noisy.graph_arrays()), id(clean): eng.graph(*clean.graph_arrays())}

    def one(wl):
        with eng.pack_windows(graphs[id(wl)], wl.windows(), 0) as b:
            b.run(); b.sync(); spec = b.speculated()
            r, o = b.fetch()
        ro, oo = want[id(wl)]
        for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            assert (r[f] == ro[f]).all(), f
        tot = int(ro["n_ops"].sum())
        assert (o[:tot].view(np.uint64) == oo[:tot].view(np.uint64)).all()
        return spec

    assert eng.speculation_state()["on"]
    ran = [one(noisy) for _ in range(10)]
    # optimistic start; off after the first count; two plain runs; a probe that fails (wait 4); four plain runs; the next probe (wait 8)
    assert ran == [True, False, False, True, False, False, False, False, True, False], ran
    st = eng.speculation_state()
    assert not st["on"] and st["turned_off"] == 1 and st["observed"] == 3 and st["last_miss"] > 0.5 and st["probe_interval"] == 8
    ran = [one(clean) for _ in range(10)]
    assert ran[:7] == [False] * 7 and ran[7:] == [True] * 3, ran          # the probe after the eight-run wait finds hardly a miss: on again
    st = eng.speculation_state()
    assert st["on"] and st["turned_on"] == 1 and st["last_miss"] < 0.1
    # the two overrides
    eng.set_speculation(2); assert one(clean) is False
    eng.set_speculation(1); assert one(noisy) is True and one(noisy) is True
    eng.set_speculation(0)
    # a resident batch run again and again without a fetch in between: its own last run is the evidence
    eng2 = capi.Engine(sc, lib=emu_lib)
    with eng2.pack_windows(eng2.graph(*noisy.graph_arrays()), noisy.windows(), 0) as b:
        ran = []
        for _ in range(4):
            b.run(); b.sync(); ran.append(b.speculated())
        r, o = b.fetch()
    assert ran == [True, False, False, True]
    assert (r["score"] == want[id(noisy)][0]["score"]).all()
    # ... and a plain run right behind a speculative one of the same resident batch: the speculative run moved its missed reads' descriptors to
    # their second wavefronts (refill_layout_one); the plain run puts them back first (refill_restore_one) and finds every read's codes
    eng3 = capi.Engine(sc, lib=emu_lib)
    with eng3.pack_windows(eng3.graph(*noisy.graph_arrays()), noisy.windows(), 0) as b:
        b.run(); b.sync(); assert b.speculated()
        b.run(); b.sync(); assert not b.speculated()
        r, o = b.fetch()
    ro, oo = want[id(noisy)]
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        assert (r[f] == ro[f]).all(), f
    tot = int(ro["n_ops"].sum())
    assert (o[:tot].view(np.uint64) == oo[:tot].view(np.uint64)).all()


def test_speculation_follows_the_miss_counts_of_earlier_batches(emu_lib, monkeypatch):
    speculation_follows_the_miss_counts(emu_lib, monkeypatch)


# ... and over DAGs with bubbles, several predecessors per node and predecessor lists in both orders (the linear workload above has chains
# only): reads of 86-94 bases fall into one lanes-per-pair geometry and the graphs are of similar widths, so the batch speculates; the second fill of a LOCAL read stops behind its
# end cell's column (refill_layout_one), which the other modes in the batch must not.
def speculative_fill_over_random_dags(lib, n_problems, seed=4242):
    from gen import random_dag, random_walk_read
    rng = np.random.default_rng(seed)
    problems = []
    while len(problems) < n_problems:
        nodes, preds = random_dag(rng, int(rng.integers(9, 15)), 22)
        if not 110 <= sum(len(x) for x in nodes) <= 150:
            continue
        read = random_walk_read(rng, nodes, preds, int(rng.integers(86, 95)), sub=0.03, indel=0.004 if rng.random() < 0.7 else 0.03)
        if not 86 <= len(read) <= 94:
            continue
        mode = capi.VGK_GSSW_LOCAL
        pinning = None
        if len(problems) % 9 == 8:                                       # a minority of the batch in the other modes
            mode = capi.VGK_GSSW_PINNED if rng.random() < 0.5 else capi.VGK_XDROP_PINNED
        p = {"read": read, "nodes": nodes, "preds": preds, "flags": mode | capi.VGK_GSSW_TRACEBACK, "pinning": None}
        if mode == capi.VGK_GSSW_PINNED:
            has_succ = [False] * len(nodes)
            for pr in preds:
                for q in pr:
                    has_succ[q] = True
            p["pinning"] = [0 if h else 1 for h in has_succ]
        if mode == capi.VGK_XDROP_PINNED:
            p["max_gap"] = int(rng.integers(0, 60))
        problems.append(p)
    ps = problem_set(problems)
    for sc in (capi.Scoring.simple(1, 4, 6, 1, 5), capi.Scoring.simple(1, 1, 1, 1, 0)):
        ro, oo = capi.Engine(sc, lib=ORACLE_LIB).align(ps, 0)
        eng = capi.Engine(sc, lib=lib)
        with eng.pack(ps, 0) as b:
            b.run(); b.sync()
            assert b.speculated() and b.kernel_ms(3) > 0, "the batch did not speculate (or no read missed)"
            r, o = b.fetch()
        for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            assert (r[f] == ro[f]).all(), f
        tot = int(ro["n_ops"].sum())
        bad = np.nonzero(o[:tot].view(np.uint64) != oo[:tot].view(np.uint64))[0]
        assert len(bad) == 0, (len(bad), np.searchsorted(np.cumsum(ro["n_ops"]), bad[:4], side="right"))


def test_speculative_fill_over_random_dags(emu_lib):
    speculative_fill_over_random_dags(emu_lib, 1150)


# ... and over WIDE graphs (380-420 columns, nodes of up to 48 bases, predecessors from anywhere before): a run that crosses into a predecessor
# far back in the column stream fetches its column block again (walk_diag_one), several times along one read; a few graphs are put first in
# the batch, where a block that would start before the arena is refused.
def first_pass_crosses_far_predecessors(lib, n_problems, seed=777):
    from gen import random_dag, random_walk_read
    rng = np.random.default_rng(seed)
    problems = []
    while len(problems) < n_problems:
        nodes, preds = random_dag(rng, int(rng.integers(10, 18)), 48, p_chain=0.45)
        if not 380 <= sum(len(x) for x in nodes) <= 420:
            continue
        read = random_walk_read(rng, nodes, preds, int(rng.integers(86, 95)), sub=0.02, indel=0.002)
        if not 86 <= len(read) <= 94:
            continue
        problems.append({"read": read, "nodes": nodes, "preds": preds, "flags": capi.VGK_GSSW_LOCAL | capi.VGK_GSSW_TRACEBACK, "pinning": None})
    ps = problem_set(problems)
    for sc in (capi.Scoring.simple(1, 4, 6, 1, 5), capi.Scoring.simple(2, 2, 3, 1, 0)):
        ro, oo = capi.Engine(sc, lib=ORACLE_LIB).align(ps, 0)
        eng = capi.Engine(sc, lib=lib)
        with eng.pack(ps, 0) as b:
            b.run(); b.sync()
            assert b.speculated()
            refilled = b.kernel_ms(3)                                  # (the emulator: wavefronts laid out again; HIP: the second fill's ms)
            r, o = b.fetch()
        st = eng.speculation_state()
        assert st["observed"] == 1 and 0 < st["last_miss"] < 0.8        # the batch speculated, and most reads were settled by runs
        assert refilled > 0
        for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            assert (r[f] == ro[f]).all(), f
        tot = int(ro["n_ops"].sum())
        bad = np.nonzero(o[:tot].view(np.uint64) != oo[:tot].view(np.uint64))[0]
        assert len(bad) == 0, (len(bad), np.searchsorted(np.cumsum(ro["n_ops"]), bad[:4], side="right"))


def test_first_pass_crosses_far_predecessors(emu_lib):
    first_pass_crosses_far_predecessors(emu_lib, 1100)
