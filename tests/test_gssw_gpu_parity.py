"""GPU parity: the HIP engine (vg_amd/libvgamd.so) against the CPU oracle through
the same C ABI, bit-exact in score, end cell and CIGAR.  Also the reference's own
unit-test vectors driven through the C++ host shim bound to the HIP engine."""
import numpy as np
import pytest

from gen import problem_set, random_problem
from test_golden_gssw_oracle import _cases, run_group
from test_gssw_emu_parity import compare
from util import ENGINE_LIB, ORACLE_LIB
from vg_amd import capi

pytestmark = pytest.mark.gpu


def test_engine_reports_a_gfx950_device():
    name, cus, mem = capi.Engine(lib=ENGINE_LIB).device_info()
    assert cus > 0 and mem > 0, (name, cus, mem)


def test_hip_matches_oracle_random_small():
    rng = np.random.default_rng(1234)
    problems = [random_problem(rng) for _ in range(3000)]
    res = compare(ENGINE_LIB, ORACLE_LIB, problems)
    assert (res["score"] > 0).sum() > 2000


def test_hip_matches_oracle_with_n_score_only_and_scoring():
    rng = np.random.default_rng(99)
    problems = [random_problem(rng, with_n=0.3) for _ in range(500)]
    problems += [random_problem(rng, traceback=False) for _ in range(300)]
    compare(ENGINE_LIB, ORACLE_LIB, problems)
    problems = [random_problem(rng, max_nodes=20, max_node_len=40, max_read=400) for _ in range(200)]
    compare(ENGINE_LIB, ORACLE_LIB, problems, capi.Scoring.simple(2, 3, 5, 2, 7))
    problems = [random_problem(rng, max_nodes=6, max_node_len=8, max_read=40) for _ in range(500)]
    compare(ENGINE_LIB, ORACLE_LIB, problems, capi.Scoring.simple(1, 4, 6, 1, 0))


def test_hip_matches_oracle_read_length_edges():
    rng = np.random.default_rng(5)
    for L in (1, 2, 15, 16, 17, 31, 32, 33, 150, 255, 256, 257, 1000, 1024):
        problems = []
        for _ in range(6):
            p = random_problem(rng, max_nodes=12, max_node_len=64, max_read=L)
            problems.append(p)
        compare(ENGINE_LIB, ORACLE_LIB, problems)


def test_hip_rejects_what_it_cannot_do():
    eng = capi.Engine(lib=ENGINE_LIB)
    p = {"read": "A" * 1025, "nodes": ["ACGT"], "preds": [[]], "flags": capi.VGK_GSSW_TRACEBACK, "pinning": None}
    with pytest.raises(capi.VgkError):
        eng.align(problem_set([p]))


def test_reference_unit_test_vectors_through_host_shim_on_hip():
    cases = [c for c in _cases("ref_aligner.json", {"align"}) if len(c["args"]) == 2 and c["args"][1] is True]
    assert run_group(cases, ENGINE_LIB) >= 40
    cases = _cases("ref_pinned_alignment.json", {"align_pinned"})
    assert run_group(cases, ENGINE_LIB) >= 400


def test_hip_xdrop_pinned_matches_oracle_and_mixes_with_gssw_modes():
    rng = np.random.default_rng(4242)
    problems = [random_problem(rng, mode=capi.VGK_XDROP_PINNED) for _ in range(2000)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED, with_n=0.2, max_read=300, max_node_len=40) for _ in range(300)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED, traceback=False) for _ in range(200)]
    res = compare(ENGINE_LIB, ORACLE_LIB, problems)
    assert (res["score"] > 0).sum() > 1000
    mixed = [random_problem(rng, mode=m) for m in (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED) * 300]
    compare(ENGINE_LIB, ORACLE_LIB, mixed)


def test_reference_xdrop_unit_test_vectors_through_host_shim_on_hip():
    from test_golden_gssw_oracle import run_xdrop_group
    ncase, nexp = run_xdrop_group(ENGINE_LIB)
    assert ncase >= 15 and nexp >= 80


def test_reference_seeded_xdrop_unit_test_vectors_through_host_shim_on_hip():
    from test_golden_gssw_oracle import run_seeded_xdrop_group
    ncase, nexp = run_seeded_xdrop_group(ENGINE_LIB)
    assert ncase >= 7 and nexp >= 20


def test_hip_quality_adjusted_contexts_match_oracle_and_reference_vectors():
    from qualadj import qual_adj_tables
    from test_golden_gssw_oracle import run_qual_adj_group
    tables = qual_adj_tables(1, 4, 5)
    rng = np.random.default_rng(31337)
    problems = []
    for mode in (capi.VGK_GSSW_LOCAL, capi.VGK_GSSW_PINNED, capi.VGK_XDROP_PINNED) * 400:
        p = random_problem(rng, mode=mode, with_n=0.1)
        p["qual"] = rng.choice(np.array([2, 5, 10, 20, 30, 40], dtype=np.uint8), size=len(p["read"]))
        problems.append(p)
    ps = problem_set(problems)
    sc = capi.Scoring.simple()
    ra, oa = capi.Engine(sc, lib=ENGINE_LIB, qual_adj=tables).align(ps)
    rb, ob = capi.Engine(sc, lib=ORACLE_LIB, qual_adj=tables).align(ps)
    for i in range(ps.n):
        assert ra["status"][i] == rb["status"][i] == 0 and ra["score"][i] == rb["score"][i], i
        if ra["score"][i] > 0:
            assert capi.cigar_string(ra[i], oa) == capi.cigar_string(rb[i], ob), i
    ncase, nexp = run_qual_adj_group(ENGINE_LIB)
    assert ncase >= 12 and nexp >= 90


def test_hip_matches_oracle_in_every_lane_geometry():
    from test_gssw_emu_parity import every_lane_geometry
    every_lane_geometry(ENGINE_LIB)


@pytest.mark.gpu
@pytest.mark.parametrize("match", [6, 7])
def test_hip_speculative_fill_with_scores_at_the_key_maximums_limit(monkeypatch, match):
    # (v_pk_maximum3_f16 on keys up to 0x71c0 with a match worth 6; with 7 the keys would pass 0x7c00 and the packers leave the three-input maximum off)
    from test_gssw_emu_parity import speculative_fill_equals_the_plain_one
    assert speculative_fill_equals_the_plain_one(ENGINE_LIB, 8000, monkeypatch, match=match) > 7500


@pytest.mark.gpu
def test_hip_two_kernel_traceback_and_speculative_fill_match_oracle(monkeypatch):
    """batches of local alignments: the tracebacks as two kernels (diagonal runs settled from the end cells; the rest by their codes) and, for
    batches of one geometry, the fill without codes first and the missed reads filled again — equal to the oracle, host- and device-packed"""
    from test_gssw_emu_parity import speculative_fill_equals_the_plain_one
    assert speculative_fill_equals_the_plain_one(ENGINE_LIB, 20000, monkeypatch) > 19000
    rng = np.random.default_rng(78)
    problems = [random_problem(rng, mode=capi.VGK_GSSW_LOCAL, max_nodes=12, max_node_len=20, max_read=140, with_n=0.03) for _ in range(4000)]
    problems += [random_problem(rng, mode=capi.VGK_XDROP_PINNED) for _ in range(300)] + [random_problem(rng, mode=capi.VGK_GSSW_PINNED) for _ in range(300)]
    for sc in (None, capi.Scoring.simple(1, 1, 1, 1, 5), capi.Scoring.simple(2, 3, 5, 2, 0)):
        compare(ENGINE_LIB, ORACLE_LIB, problems, sc)


# The speculative fill over graphs that are not chains, on the device (emulator-only until now): bubbles, several
# predecessors per node in both orders, a minority of the batch in the other modes; wide graphs whose diagonal runs cross into predecessors far
# back in the column stream (walk_diag_one's re-fetched column block); and the LOCAL windows of a resident VARIATION graph with reads of one
# length — one lane geometry, so the device-packed batch speculates.
def test_hip_speculative_fill_over_random_dags():
    from test_gssw_emu_parity import speculative_fill_over_random_dags
    speculative_fill_over_random_dags(ENGINE_LIB, 6000, seed=4243)


def test_hip_first_pass_crosses_far_predecessors():
    from test_gssw_emu_parity import first_pass_crosses_far_predecessors
    first_pass_crosses_far_predecessors(ENGINE_LIB, 4000, seed=778)


def test_hip_speculative_fill_over_windows_of_a_variation_graph():
    from test_windows import speculative_windows_of_a_variation_graph
    speculative_windows_of_a_variation_graph(ENGINE_LIB, n_nodes=6000, n_problems=20000)


def test_hip_speculation_follows_the_miss_counts_of_earlier_batches(monkeypatch):
    from test_gssw_emu_parity import speculation_follows_the_miss_counts
    speculation_follows_the_miss_counts(ENGINE_LIB, monkeypatch, n=20000)
