"""k-best pinned alignments (Aligner::align_pinned_multi).

gssw's own multi-traceback is not in the reference snapshot; what pins the behaviour are the nine property sections of
src/unittest/pinned_alignment.cpp:1951-2530, transcribed by hand below (line numbers in the comments) and driven through the
C++ host shim exactly like the reference drives vg's Aligner.  The HIP engine is then compared with the oracle on random
problems, alternate by alternate."""
import os
import numpy as np
import pytest

import util
from gen import problem_set, random_problem
from vg_amd import capi

DIAMOND = ([(1, "ACGTAGTCTGAA"), (2, "CA"), (3, "TT"), (4, "TGACGTACGTTA")], [(1, 2), (1, 3), (2, 4), (3, 4)])
DIAMOND_READ = "ACGTAGTCTGACATGACGTACGTTA"


def mappings(aln):
    return [(m["position"]["node_id"], m["position"].get("offset", 0),
             [(e.get("from_length", 0), e.get("to_length", 0), e.get("sequence", "")) for e in m["edit"]]) for m in aln["path"]["mapping"]]


def is_pinned(aln, pinned_id, pinned_len, pin_left):
    m = mappings(aln)
    if pin_left:
        return m[0][1] == 0 and m[0][0] == pinned_id
    return sum(e[0] for e in m[-1][2]) == pinned_len and m[-1][0] == pinned_id


def multi(al, nodes, edges, read, pin_left, n, quality=None):
    return al.run(nodes, edges, read, "align_pinned_multi", pin_left=pin_left, max_alt_alns=n, quality=quality)


def matches(aln, nodes_expected, edits_expected):
    m = mappings(aln)
    return [x[0] for x in m] == nodes_expected and [x[2] for x in m] == edits_expected


def reference_property_cases(engine_lib):
    al = util.HostAligner(engine_lib)
    nodes, edges = DIAMOND
    # descending score order, every alternate pinned (:1951)
    alts = multi(al, nodes, edges, DIAMOND_READ, False, 20)
    assert len(alts) > 1
    assert all(is_pinned(a, 4, 12, False) for a in alts)
    scores = [a["score"] for a in alts]
    assert scores == sorted(scores, reverse=True) and scores[-1] > 0
    # the first alternate is the single pinned alignment (:1998)
    single = al.run(nodes, edges, DIAMOND_READ, "align_pinned", pin_left=False)
    two = multi(al, nodes, edges, DIAMOND_READ, False, 2)
    assert two[0]["score"] == single["score"] and mappings(two[0]) == mappings(single)
    # both optimal alignments of a deletion in a homodimer, pinned left (:2041)
    two = multi(al, nodes, edges, DIAMOND_READ, True, 2)
    assert all(is_pinned(a, 1, 12, True) for a in two)
    first = ([1, 2, 4], [[(11, 11, ""), (1, 0, "")], [(2, 2, "")], [(12, 12, "")]])
    second = ([1, 2, 4], [[(10, 10, ""), (1, 0, ""), (1, 1, "")], [(2, 2, "")], [(12, 12, "")]])
    assert any(matches(a, *first) for a in two) and any(matches(a, *second) for a in two)
    # an alternate that follows another node sequence (:2142)
    snp = ([(1, "ACGTAGTCTGAA"), (2, "C"), (3, "T"), (4, "TGACGTACGTTA")], edges)
    alts = multi(al, snp[0], snp[1], "ACGTAGTCTGAACTGACGTACGTTA", True, 20)
    assert all(is_pinned(a, 1, 12, True) for a in alts)
    assert any(mappings(a)[1][0] != mappings(alts[0])[1][0] for a in alts)
    # no alternates when none scores positively (:2192)
    assert len(multi(al, [(1, "CA")], [], "A", False, 100)) == 1
    # alternates that branch from another alternate at a node boundary (:2214)
    g6 = ([(1, "AAAAAAAA"), (2, "GGG"), (3, "G"), (4, "C"), (5, "T"), (6, "G"), (7, "AAAAAA")],
          [(1, 2), (1, 3), (3, 4), (3, 5), (4, 6), (5, 6), (2, 7), (6, 7)])
    alts = multi(al, g6[0], g6[1], "AAAAAAAAGGGAAAAAA", False, 3)
    assert all(is_pinned(a, 7, 6, False) for a in alts)
    via_c = ([1, 3, 4, 6, 7], [[(8, 8, "")], [(1, 1, "")], [(1, 1, "G")], [(1, 1, "")], [(6, 6, "")]])
    via_t = ([1, 3, 5, 6, 7], [[(8, 8, "")], [(1, 1, "")], [(1, 1, "G")], [(1, 1, "")], [(6, 6, "")]])
    assert any(matches(a, *via_c) for a in alts) and any(matches(a, *via_t) for a in alts)
    # alternates that branch from another alternate inside a node (:2331)
    g7 = ([(1, "AAAAAAAAAA"), (2, "CGGC"), (3, "CGGT"), (4, "AAAAAAAAAA")], edges)
    alts = multi(al, g7[0], g7[1], "AAAAAAAAAACGGGCAAAAAAAAAA", False, 10)
    assert all(is_pinned(a, 4, 10, False) for a in alts)
    ins_first = ([1, 3, 4], [[(10, 10, "")], [(1, 1, ""), (0, 1, "G"), (2, 2, ""), (1, 1, "C")], [(10, 10, "")]])
    ins_second = ([1, 3, 4], [[(10, 10, "")], [(2, 2, ""), (0, 1, "G"), (1, 1, ""), (1, 1, "C")], [(10, 10, "")]])
    assert any(matches(a, *ins_first) for a in alts) and any(matches(a, *ins_second) for a in alts)
    # no duplicates among 5000 alternates of a low-complexity read (:2495)
    al0 = util.HostAligner(engine_lib, scores=(1, 4, 6, 1, 0))
    g9 = ([(1, "CCCCCCCCCTCCCCCCCCCCTCCCCCCCCCCGACCCCCCCCCCC"), (2, "CCCCCCCCCCACCCCCCCCCCACCCCCCCCCCTCCCA"), (3, "CCCCCACCCCCCCCGTCCCCCCCCCCCA"),
           (4, "CCCCCCCCCCCCGCCCCCCCCCCGCCCCCCCCC")], edges)
    read9 = "CCCCCCCTCCCCCCCCCCTCCCCCCCCCCGACCCCCCCCCCCCCCCCCCCCCACCCCCCCCCCACCCCCCCCCCTCCCACCCCCCCCCCCCGCCCCCCCCCCGCCCCCCCCC"
    alts = multi(al0, g9[0], g9[1], read9, False, 5000)
    seen = set()
    for a in alts:
        key = repr(mappings(a))
        assert key not in seen
        seen.add(key)
    assert len(alts) > 100
    return len(alts)


def quality_adjusted_case(engine_lib):
    # matches over node boundaries keep offsets >= 0 and the whole read (:2455)
    al = util.HostAligner(engine_lib, qual_adj=True)
    g8 = ([(1, "T"), (2, "C"), (3, "A"), (4, "CCCTGCTAGTCTGGAGTTGATCAAGGAACCTGTCT")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    qual = [ord(c) - 33 for c in "<<<''"]
    alts = multi(al, g8[0], g8[1], "CCCGG", True, 100, quality=qual)
    assert alts
    for a in alts:
        assert all(off >= 0 for _, off, _ in mappings(a))
        assert sum(e[1] for _, _, ed in mappings(a) for e in ed) == 5


def test_oracle_has_the_reference_properties():
    assert reference_property_cases(util.ORACLE_LIB) > 100
    quality_adjusted_case(util.ORACLE_LIB)


# ---- random problems: the engine against the oracle, alternate by alternate ---------------------------------------------

def compare_engines(lib, seeds, n_problems=40, max_alt=30, on_device=True):
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    total = 0
    for s in seeds:
        rng = np.random.default_rng(s)
        problems = [random_problem(rng, max_nodes=8, max_node_len=10, max_read=40, mode=capi.VGK_GSSW_PINNED) for _ in range(n_problems)]
        ps = problem_set(problems)
        ra, ca, oa = ora.align_multi(ps, max_alt)
        rb, cb, ob = eng.align_multi(ps, max_alt)
        if on_device:              # the alternates were enumerated by the kernel (gssw_multi_device.hpp), none by a host thread
            assert eng.multi_host_walks == 0, (s, eng.multi_host_walks)
        else:
            assert eng.multi_host_walks > 0
        assert (ca == cb).all(), (s, ca, cb)
        for i in range(ps.n):
            for k in range(int(ca[i])):
                a, b = ra[i, k], rb[i, k]
                assert a["score"] == b["score"] and a["status"] == b["status"] == 0 and a["first_offset"] == b["first_offset"], (s, i, k, a, b)
                assert capi.cigar_string(a, oa) == capi.cigar_string(b, ob), (s, i, k)
            if ca[i]:
                assert list(ra[i, :ca[i]]["score"]) == sorted(ra[i, :ca[i]]["score"], reverse=True)
                rendered = [(int(ra[i, k]["first_offset"]), capi.cigar_string(ra[i, k], oa)) for k in range(int(ca[i]))]
                assert len(set(rendered)) == len(rendered), (s, i, rendered)          # no alignment twice
            total += int(ca[i])
        # the first alternate is what the single-alignment entry point returns
        rs, os_ = eng.align(ps)
        for i in range(ps.n):
            if cb[i]:
                assert rs["score"][i] == rb[i, 0]["score"] and capi.cigar_string(rs[i], os_) == capi.cigar_string(rb[i, 0], ob), (s, i)
            else:
                assert rs["score"][i] <= 0 or rb[i, 0]["status"] != 0
    return total


def test_emulated_pinned_multi_matches_oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    assert reference_property_cases(util.EMU_LIB) > 100
    quality_adjusted_case(util.EMU_LIB)
    assert compare_engines(util.EMU_LIB, range(500, 509)) > 600          # (the emulator steps every lane: nine seeds here, forty on the MI355X)
    # what the kernel's tables cannot hold goes to host threads under the same rules: more alternates than a lane's slot pool ...
    assert compare_engines(util.EMU_LIB, range(520, 523), max_alt=80, on_device=False) > 300
    # ... and (forced) every problem
    os.environ["VGAMD_MULTI_HOST_WALK"] = "1"
    try:
        assert compare_engines(util.EMU_LIB, range(523, 526), on_device=False) > 200
    finally:
        del os.environ["VGAMD_MULTI_HOST_WALK"]


@pytest.mark.gpu
def test_hip_pinned_multi_matches_oracle():
    assert reference_property_cases(util.ENGINE_LIB) > 100
    quality_adjusted_case(util.ENGINE_LIB)
    assert compare_engines(None, range(600, 640), n_problems=100) > 15000


def declined_by_the_kernel_case(lib):
    """A node with 17 predecessors is more than a lane's source table holds: the kernel answers VGK_ETOOBIG for that problem alone and a
    host thread walks it (its matrices are the only ones copied back); the neighbours stay on the device; all equal the oracle."""
    rng = np.random.default_rng(77)
    wide = {"read": "ACGTACGTTG", "nodes": ["ACGTA", "ACGTT", "ACGAA", "ACTTA", "AGGTA", "CCGTA", "ACGTC", "ACGGA", "TCGTA", "ACGTG", "AAGTA",
                                            "ACCTA", "ACGCA", "GCGTA", "ACGTA", "ATGTA", "ACGAT", "CGTTG"],
            "preds": [[] for _ in range(17)] + [list(range(17))], "flags": capi.VGK_GSSW_PINNED | capi.VGK_GSSW_TRACEBACK, "pinning": [0] * 17 + [1]}
    problems = [random_problem(rng, max_nodes=8, max_node_len=10, max_read=40, mode=capi.VGK_GSSW_PINNED) for _ in range(6)]
    problems.insert(3, wide)
    ps = problem_set(problems)
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    ra, ca, oa = ora.align_multi(ps, 20)
    rb, cb, ob = eng.align_multi(ps, 20)
    assert eng.multi_host_walks == 1
    assert (ca == cb).all() and ca[3] > 5
    for i in range(ps.n):
        for k in range(int(ca[i])):
            assert ra[i, k]["score"] == rb[i, k]["score"] and ra[i, k]["first_offset"] == rb[i, k]["first_offset"]
            assert capi.cigar_string(ra[i, k], oa) == capi.cigar_string(rb[i, k], ob), (i, k)


def test_emulated_kernel_hands_what_it_cannot_hold_to_a_host_thread():
    declined_by_the_kernel_case(util.EMU_LIB)


@pytest.mark.gpu
def test_hip_kernel_hands_what_it_cannot_hold_to_a_host_thread():
    declined_by_the_kernel_case(None)
