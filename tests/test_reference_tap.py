"""Known answers the reference holds outside its Catch unit tests, plus the small parity rows of SURVEY.md §8:
  * test/t/04_vg_align.t — the four `vg align` pins whose inputs are self-contained .vg files (first-mapping node ids under
    lenient scoring / Ns / lower-case nodes, and the score 274 that must not saturate at 255), transcribed by
    tests/golden/extract_tap_tests.py;
  * src/unittest/aligner.cpp:450-500 — Aligner::align(alignment, graph, topological_order) on both strands of a cyclic graph (row a7);
  * configs[0] — test/tiny/tiny.gfa, 500 x 50 bp reads, perfect reads score 50 + 2 * 5;
  * src/alignment_scorer.cpp:264-271 — longest_detectable_gap (row a20).
Each runs on the oracle here and on the HIP engine in the gpu tests."""
import ctypes
import json

import numpy as np
import pytest

import util
from util import ENGINE_LIB, ORACLE_LIB, HostAligner, check_expectations, load_golden
from vg_amd import capi


# ---- test/t/04_vg_align.t ---------------------------------------------------------------------------------------------------
def run_tap(engine_lib):
    n = 0
    for c in load_golden("ref_tap_align.json"):
        aln = HostAligner(engine_lib, tuple(c["scores"])).run(c["nodes"], c["edges"], c["read"], "align")
        check_expectations(c, aln)
        n += len(c["expect"])
    return n


def test_vg_align_tap_pins_on_the_oracle():
    assert run_tap(ORACLE_LIB) == 4


def test_score_above_255_is_not_saturated_named_case():
    """04_vg_align.t:30 "alignment score does not overflow at 255 when using 8x16bit vectors": 274 with 2/2/3/1/0."""
    c = [c for c in load_golden("ref_tap_align.json") if c["source"].endswith(":30")][0]
    for lib in (ORACLE_LIB, util.EMU_LIB):
        assert HostAligner(lib, tuple(c["scores"])).run(c["nodes"], c["edges"], c["read"], "align")["score"] == 274


@pytest.mark.gpu
def test_vg_align_tap_pins_on_hip():
    assert run_tap(ENGINE_LIB) == 4


# ---- Aligner::align(alignment, graph, topological_order): src/unittest/aligner.cpp:450-500 ---------------------------------
def align_order(engine_lib, read):
    h = util.host()
    h.vgh_align_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                  ctypes.c_char_p, ctypes.c_size_t]
    al = HostAligner(engine_lib, (1, 4, 6, 1, 0))                   # set_alignment_scores(1, 4, 6, 1, 0)  (:475)
    g = h.vgh_graph_create()
    try:
        seqs = ["AAAA", "GATT", "ACAT", "AAAA"]                     # four nodes, ids 1..4 (:455-459)
        for i, s in enumerate(seqs):
            assert h.vgh_graph_add_node(g, i + 1, s.encode()) == 0
        for i in range(4):                                          # "Make the graph a cycle" (:462-464)
            assert h.vgh_graph_add_edge(g, i + 1, (i + 1) % 4 + 1) == 0
        order = [2 * 2, 2 * 3, 2 * 3 + 1, 2 * 2 + 1]                # handles[1], handles[2], flip(handles[2]), flip(handles[1]) (:471-473)
        arr = (ctypes.c_int64 * 4)(*order)
        buf = ctypes.create_string_buffer(1 << 16)
        assert h.vgh_align_order(al.ptr, g, read.encode(), arr, 4, buf, len(buf)) == 0, h.vgh_last_error().decode()
        return json.loads(buf.value.decode()), order
    finally:
        h.vgh_graph_destroy(g)


def check_mapping(m, oriented, offset, length):                      # check_mapping (:440-448)
    assert m["position"]["node_id"] == oriented >> 1
    assert bool(m["position"].get("is_reverse", False)) == bool(oriented & 1)
    assert m["position"].get("offset", 0) == offset
    assert len(m["edit"]) == 1
    assert m["edit"][0]["from_length"] == length and m["edit"][0]["to_length"] == length and not m["edit"][0].get("sequence")


def subgraph_sections(engine_lib):
    aln, order = align_order(engine_lib, "ATTACA")                    # "Align to forward strand" (:479-488)
    maps = aln["path"]["mapping"]
    assert len(maps) == 2
    check_mapping(maps[0], order[0], 1, 3); check_mapping(maps[1], order[1], 0, 3)
    aln, order = align_order(engine_lib, "TGTAAT")                    # "Align to reverse strand" (:490-499)
    maps = aln["path"]["mapping"]
    assert len(maps) == 2
    check_mapping(maps[0], order[2], 1, 3); check_mapping(maps[1], order[3], 0, 3)


def test_aligner_can_align_to_a_subgraph_on_the_oracle():
    subgraph_sections(ORACLE_LIB)


@pytest.mark.gpu
def test_aligner_can_align_to_a_subgraph_on_hip():
    subgraph_sections(ENGINE_LIB)


# ---- configs[0]: test/tiny/tiny.gfa, 500 x 50 bp ------------------------------------------------------------------------------
def tiny_problems(n=500, seed=1337, sub=0.01, indel=0.002):
    """50 bp walks from random source-to-sink paths of the 15-node tiny graph (every such walk is 50 bp),
    substitutions 1 %, indels 0.2 % (stand-in for `vg sim -l 50 -s 1337 -e 0.01 -i 0.002`).  -> (problem dicts, perfect flags)"""
    tg = load_golden("tiny_graph.json")
    ids = [nid for nid, _ in tg["nodes"]]
    seq = dict((nid, s) for nid, s in tg["nodes"])
    succ = {i: [] for i in ids}; pred = {i: [] for i in ids}
    for a, b in tg["edges"]:
        succ[a].append(b); pred[b].append(a)
    order = sorted(ids)                                              # ids of tiny.gfa are already topological
    assert all(a < b for a, b in tg["edges"])
    index = {nid: k for k, nid in enumerate(order)}
    nodes = [seq[i] for i in order]
    preds = [[index[p] for p in pred[i]] for i in order]
    rng = np.random.default_rng(seed)
    problems, perfect = [], []
    for _ in range(n):
        v = order[0]; walk = []
        while True:
            walk.append(seq[v])
            if not succ[v]:
                break
            v = succ[v][int(rng.integers(0, len(succ[v])))]
        ref = "".join(walk)
        assert len(ref) == 50
        out, clean = [], True
        for ch in ref:
            r = rng.random()
            if r < sub:
                alt = "ACGT"[int(rng.integers(0, 4))]; clean &= alt == ch; out.append(alt)
            elif r < sub + indel / 2:
                clean = False
            elif r < sub + indel:
                clean = False; out.append("ACGT"[int(rng.integers(0, 4))]); out.append(ch)
            else:
                out.append(ch)
        problems.append(dict(read="".join(out) or "A", nodes=nodes, preds=preds, flags=capi.VGK_GSSW_LOCAL | capi.VGK_GSSW_TRACEBACK, pinning=None))
        perfect.append(clean)
    return problems, np.array(perfect)


def config0(engine_lib):
    problems, perfect = tiny_problems()
    ps = capi.ProblemSet.from_lists(problems)
    ro, oo = capi.Engine(lib=ORACLE_LIB).align(ps)
    assert perfect.sum() > 250
    assert (ro["score"][perfect] == 60).all()                        # 50 matches + the full-length bonus at both ends
    assert (ro["score"][~perfect] <= 60).all() and (ro["score"][~perfect] < 60).mean() > 0.8     # (a substitution at a SNP site can spell the other allele)
    re_, oe = capi.Engine(lib=engine_lib).align(ps)
    for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        assert (re_[f] == ro[f]).all(), f
    assert (oe.view(np.uint64) == oo.view(np.uint64)).all()
    # the same reads as windows of the resident graph (every window is the whole 15-node graph)
    eng = capi.Engine(lib=engine_lib)
    node_len = [len(s) for s in problems[0]["nodes"]]
    pred_off = np.concatenate([[0], np.cumsum([len(p) for p in problems[0]["preds"]])])
    g = eng.graph(node_len, np.frombuffer("".join(problems[0]["nodes"]).encode(), dtype=np.uint8), pred_off, [q for p in problems[0]["preds"] for q in p])
    ws = capi.WindowSet(ps.reads, ps.read_off, np.zeros(ps.n, np.uint32), np.full(ps.n, len(node_len)), ps.flags, cols=np.full(ps.n, 50))
    rw, ow = eng.align_windows(g, ws)
    assert (rw["score"] == ro["score"]).all() and (ow.view(np.uint64) == oo.view(np.uint64)).all()
    return int(perfect.sum())


def test_config0_tiny_graph_on_the_emulated_engine():
    assert config0(util.EMU_LIB) > 250


@pytest.mark.gpu
def test_config0_tiny_graph_on_hip():
    assert config0(ENGINE_LIB) > 250


# ---- longest_detectable_gap (src/alignment_scorer.cpp:264-271) -------------------------------------------------------------------
def test_longest_detectable_gap_matches_the_reference_formula():
    h = util.host()
    h.vgh_longest_detectable_gap.restype = ctypes.c_int64
    h.vgh_longest_detectable_gap.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    for scores in ((1, 4, 6, 1, 5), (2, 3, 5, 2, 7), (1, 4, 6, 1, 0)):
        m, _, go, ge, bonus = scores
        al = HostAligner(ORACLE_LIB, scores)
        for L in (1, 2, 50, 150, 151, 1000):
            for pos in sorted({0, 1, L // 3, L // 2, L - 1, L}):
                overhang = min(pos, L - pos)
                gap = (m * overhang + bonus - go) // ge + 1 if (m * overhang + bonus - go) >= 0 else -((-(m * overhang + bonus - go)) // ge) + 1   # C++ division truncates
                want = gap if gap >= 0 and overhang > 0 else 0
                assert h.vgh_longest_detectable_gap(al.ptr, L, pos) == want, (scores, L, pos)
    al = HostAligner(ORACLE_LIB, (1, 4, 6, 1, 5))
    assert h.vgh_longest_detectable_gap(al.ptr, 150, 75) == 75       # "L=150 mid-read => 75"
    # the tails workload sizes its windows with it (vg_amd/workloads.py) — same numbers as the scorer's
    from vg_amd import workloads
    for t in (1, 5, 40, 75, 121):
        assert workloads.longest_detectable_gap(2 * t, t, 1, 6, 1, 5) == h.vgh_longest_detectable_gap(al.ptr, 2 * t, t)


# ---- host arithmetic the aligner interface carries: MappingQualityCalculator (GSSWAligner::mapq_calc, src/aligner.hpp:148) and the scorer's
# score_contiguous_alignment — the reference's own unit tests, src/unittest/aligner.cpp:347-442 ----
def test_mapping_quality_estimation_is_robust():
    """src/unittest/aligner.cpp:371-436: the element chosen by maximum_mapping_quality_exact / _approx"""
    import ctypes
    from util import host
    h = host()
    h.vgh_maximum_mapping_quality.restype = ctypes.c_double
    h.vgh_maximum_mapping_quality.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    for approx in (0, 1):
        for scores, ok in (([10.0], {0}), ([0.0], {0}), ([-10.0], {0}), ([1.0, 5.0, 2.0, 5.0, 4.0], {1, 3})):
            idx = ctypes.c_int64(-1)
            q = h.vgh_maximum_mapping_quality((ctypes.c_double * len(scores))(*scores), len(scores), approx, ctypes.byref(idx))
            assert idx.value in ok and q >= 0.0, (approx, scores, idx.value, q)
    # the approximation and the exact value agree when one score stands far above the rest: 10 / ln 10 x the gap to the runner-up
    s = [40.0, 10.0, 3.0]
    q = [h.vgh_maximum_mapping_quality((ctypes.c_double * 3)(*s), 3, a, None) for a in (0, 1)]
    assert abs(q[0] - q[1]) < 0.1 and abs(q[1] - 30.0 * 10.0 / np.log(10.0)) < 1e-9


def test_full_length_bonus_is_applied_to_both_ends_by_rescoring():
    """src/unittest/aligner.cpp:347-369: score_contiguous_alignment of the test's alignment is 129 without a bonus, 139 with 5 at either end"""
    import ctypes
    from util import HostAligner, ORACLE_LIB, host
    seq = "ACCCCGTCTCTACTAAAAATACAAAAATTAGCCGGGTGTGGTGGCATGCACCTGTAATCCCAGCTACTGGGCATGCTGAGGTAGCAGAATCGCTTGAACCCAGGAGGAACCGGTTGCAGTGAGCCGAGATTGTGCCACTCCACTCCAG"
    # (mapping, from_length, to_length, carries a sequence) as in the test's JSON: ten all-match mappings, one pure deletion node, then
    # deletion / match / insertion "CCG" / match, then a match
    edits = [(0, 4, 4, 0), (1, 1, 1, 0), (2, 3, 3, 0), (3, 1, 1, 0), (4, 32, 32, 0), (5, 32, 32, 0), (6, 8, 8, 0), (7, 1, 1, 0), (8, 24, 24, 0), (9, 1, 0, 0),
             (10, 2, 0, 0), (10, 3, 3, 0), (10, 0, 3, 1), (10, 27, 27, 0), (11, 9, 9, 0)]
    assert sum(e[2] for e in edits) == len(seq)
    h = host()
    h.vgh_score_contiguous_alignment.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    flat = (ctypes.c_int64 * (4 * len(edits)))(*[x for e in edits for x in e])
    assert h.vgh_score_contiguous_alignment(HostAligner(ORACLE_LIB, (1, 4, 6, 1, 0)).ptr, seq.encode(), flat, len(edits)) == 129
    assert h.vgh_score_contiguous_alignment(HostAligner(ORACLE_LIB, (1, 4, 6, 1, 5)).ptr, seq.encode(), flat, len(edits)) == 139


def test_aligner_carries_a_mapping_quality_calculator():
    """GSSWAligner::mapq_calc: built from the scorer's match / mismatch and the log base recovered from the matrix (1/4 scoring at 50 % GC)"""
    import ctypes
    from util import HostAligner, ORACLE_LIB, host
    h = host()
    h.vgh_log_base.restype = ctypes.c_double; h.vgh_log_base.argtypes = [ctypes.c_void_p]
    h.vgh_compute_mapping_quality.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    al = HostAligner(ORACLE_LIB)
    lam = h.vgh_log_base(al.ptr)
    assert abs(0.25 * (np.exp(lam) + 3 * np.exp(-4 * lam)) - 1.0) < 1e-9 and lam > 0          # the partition function of a log-odds matrix is 1
    s = [150.0, 140.0]
    exact = h.vgh_compute_mapping_quality(al.ptr, (ctypes.c_double * 2)(*s), 2, 0, 0); fast = h.vgh_compute_mapping_quality(al.ptr, (ctypes.c_double * 2)(*s), 2, 1, 0)
    assert fast == int(10.0 / np.log(10.0) * lam * 10.0) and abs(exact - fast) <= 1
    # a lead the doubles cannot express any more is the largest quality there is (the reference's isinf branch, mapping_quality_calculator.cpp:66)
    s = [150.0, 100.0]
    assert h.vgh_compute_mapping_quality(al.ptr, (ctypes.c_double * 2)(*s), 2, 0, 0) == 2**31 - 1
