"""Pin the gssw oracle (oracle/vgo_gssw.c) + host logic (vg_amd/host) against the
reference's own known-answer unit tests, transcribed into tests/golden/ by
tests/golden/extract_reference_tests.py from src/unittest/aligner.cpp and
src/unittest/pinned_alignment.cpp.  CPU only: the host shim is bound to the
oracle library explicitly here (the product binds the HIP library)."""
import pytest

from util import HostAligner, ORACLE_LIB, check_expectations, load_golden, run_align_xdrop


def _cases(fname, calls):
    return [c for c in load_golden(fname) if c["call"] in calls and not c["qual_adj"]]


def _run(case, engine_lib):
    al = HostAligner(engine_lib, tuple(case["scores"]))
    if case["call"] == "align":
        tb = case["args"][1] if len(case["args"]) > 1 else True
        return al.run(case["nodes"], case["edges"], case["read"], "align" if tb is True else "align_score")
    if case["call"] == "align_pinned":
        return al.run(case["nodes"], case["edges"], case["read"], "align_pinned", pin_left=bool(case["args"][1]))
    raise AssertionError(case["call"])


def run_group(cases, engine_lib):
    """Cases from one source location may refer to each other's scores (aln1/aln2)."""
    by_source = {}
    for c in cases:
        by_source.setdefault(c["source"], []).append(c)
    n = 0
    for src, group in by_source.items():
        alns = {c["aln"]: _run(c, engine_lib) for c in group}
        scores = {k: v["score"] for k, v in alns.items()}
        for c in group:
            check_expectations(c, alns[c["aln"]], scores)
            n += len(c["expect"])
    return n


def test_oracle_matches_reference_aligner_unit_tests():
    cases = [c for c in _cases("ref_aligner.json", {"align"}) if len(c["args"]) == 2 and c["args"][1] is True]
    assert len(cases) >= 20
    assert run_group(cases, ORACLE_LIB) >= 40


def test_oracle_matches_reference_pinned_alignment_unit_tests():
    cases = _cases("ref_pinned_alignment.json", {"align_pinned"})
    assert len(cases) >= 28
    assert run_group(cases, ORACLE_LIB) >= 400


def _run_xdrop(case, engine_lib):
    al = HostAligner(engine_lib, tuple(case["scores"]))
    args = case["args"]          # [graph, pin_left, xdrop, max_gap?]
    max_gap = args[3] if len(args) > 3 and isinstance(args[3], int) else 40
    return al.run(case["nodes"], case["edges"], case["read"], "align_pinned_xdrop", pin_left=bool(args[1]), max_alt_alns=max_gap)


def xdrop_pinned_cases():
    return [c for c in load_golden("ref_xdrop_aligner.json")
            if c["call"] == "align_pinned" and not c["qual_adj"] and len(c["args"]) >= 3 and c["args"][2] is True]


def run_xdrop_group(engine_lib):
    n = 0
    cases = xdrop_pinned_cases()
    for c in cases:
        aln = _run_xdrop(c, engine_lib)
        check_expectations(c, aln, {c["aln"]: aln["score"]})
        n += len(c["expect"])
    return len(cases), n


def test_oracle_matches_reference_xdrop_pinned_unit_tests():
    ncase, nexp = run_xdrop_group(ORACLE_LIB)
    assert ncase >= 15 and nexp >= 80


def seeded_xdrop_cases():
    return [c for c in load_golden("ref_xdrop_aligner.json")
            if c["call"] == "align_xdrop" and not c["qual_adj"] and isinstance(c["args"][1], dict) and c["nodes"]]


def run_seeded_xdrop_group(engine_lib):
    n = 0
    cases = seeded_xdrop_cases()
    for c in cases:
        al = HostAligner(engine_lib, tuple(c["scores"]))
        args = c["args"]           # [graph, {mems}, reverse_complemented, max_gap?]
        max_gap = args[3] if len(args) > 3 and isinstance(args[3], int) else 40
        aln = run_align_xdrop(al, c["nodes"], c["edges"], c["read"], args[1]["mems"], bool(args[2]), max_gap)
        check_expectations(c, aln, {c["aln"]: aln["score"]})
        n += len(c["expect"])
    return len(cases), n


def test_oracle_matches_reference_seeded_xdrop_unit_tests():
    ncase, nexp = run_seeded_xdrop_group(ORACLE_LIB)
    assert ncase >= 7 and nexp >= 20


def qual_adj_cases():
    out = []
    for f in ("ref_pinned_alignment.json", "ref_xdrop_aligner.json"):
        out += [c for c in load_golden(f) if c["qual_adj"] and c["call"] == "align_pinned" and c["quality"]]
    return out


def run_qual_adj_group(engine_lib):
    n = 0
    cases = qual_adj_cases()
    for c in cases:
        al = HostAligner(engine_lib, tuple(c["scores"]), qual_adj=True)
        args = c["args"]           # [graph, pin_left, xdrop?, max_gap?]
        xdrop = len(args) > 2 and args[2] is True
        if xdrop:
            max_gap = args[3] if len(args) > 3 and isinstance(args[3], int) else 40
            aln = al.run(c["nodes"], c["edges"], c["read"], "align_pinned_xdrop", pin_left=bool(args[1]), max_alt_alns=max_gap, quality=c["quality"])
        else:
            aln = al.run(c["nodes"], c["edges"], c["read"], "align_pinned", pin_left=bool(args[1]), quality=c["quality"])
        check_expectations(c, aln, {c["aln"]: aln["score"]})
        n += len(c["expect"])
    return len(cases), n


def test_oracle_matches_reference_quality_adjusted_unit_tests():
    ncase, nexp = run_qual_adj_group(ORACLE_LIB)
    assert ncase >= 12 and nexp >= 90


# ---- X-drop with dozeu's band restated (vgk_xdrop_band_align; PARITY-UNPINNED) -------------------------------------------------------
def run_xdrop_cases_with_band(engine_lib):
    """Every reference X-drop case (pinned and seeded, src/unittest/xdrop_aligner.cpp) through the shim with the banded extension
    switched on: the reference's answers are those of an extension whose band contains the optimum, so they must not change."""
    import ctypes
    import util
    h = util.host()
    h.vgh_aligner_set_xdrop_band.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = 0
    for c in xdrop_pinned_cases():
        al = HostAligner(engine_lib, tuple(c["scores"])); h.vgh_aligner_set_xdrop_band(al.ptr, 1)
        args = c["args"]; max_gap = args[3] if len(args) > 3 and isinstance(args[3], int) else 40
        aln = al.run(c["nodes"], c["edges"], c["read"], "align_pinned_xdrop", pin_left=bool(args[1]), max_alt_alns=max_gap)
        check_expectations(c, aln, {c["aln"]: aln["score"]}); n += 1
    for c in seeded_xdrop_cases():
        al = HostAligner(engine_lib, tuple(c["scores"])); h.vgh_aligner_set_xdrop_band(al.ptr, 1)
        args = c["args"]; max_gap = args[3] if len(args) > 3 and isinstance(args[3], int) else 40
        aln = run_align_xdrop(al, c["nodes"], c["edges"], c["read"], args[1]["mems"], bool(args[2]), max_gap)
        check_expectations(c, aln, {c["aln"]: aln["score"]}); n += 1
    return n


def test_reference_xdrop_cases_hold_with_the_band_on_the_oracle():
    assert run_xdrop_cases_with_band(ORACLE_LIB) >= 22
