"""giraffe's alignment between / beyond anchors (vg_amd/host/chain_alignment.hpp: align_sequence_between, its orientation-independent
form, with_dagified_local_graph, longest_detectable_gap_in_range) pinned on the reference's own known-answer tests —
src/unittest/minimizer_mapper.cpp:254-880, transcribed into tests/golden/ref_minimizer_mapper.json by
tests/golden/extract_minimizer_mapper_tests.py (every REQUIRE kept as an expression over the alignment).

CPU: the host shim bound to the oracle library and to the emulated kernels; `-m gpu`: bound to the HIP engine.  The batched form
(ChainConnector: many requests, the local graphs on host threads, one engine flush) must answer every request exactly as the direct
call does."""
import ctypes
import json
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT, HostAligner, host, load_golden

CASES = load_golden("ref_minimizer_mapper.json")["cases"]


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def _bind():
    h = host()
    if getattr(h, "_chain_bound", False):
        return h
    h.vgh_bigraph_create.restype = ctypes.c_void_p
    h.vgh_bigraph_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_bigraph_add_node.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    h.vgh_bigraph_add_edge.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    P = ctypes.POINTER(ctypes.c_int64)
    h.vgh_align_sequence_between.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, P, P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_char_p, ctypes.c_size_t]
    h.vgh_dagified_local_graph.argtypes = [ctypes.c_void_p, P, P, ctypes.c_int64, ctypes.c_char_p, ctypes.c_size_t]
    h.vgh_longest_detectable_gap_in_range.restype = ctypes.c_int64
    h.vgh_longest_detectable_gap_in_range.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    h.vgh_connector_create.restype = ctypes.c_void_p
    h.vgh_connector_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    h.vgh_connector_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_connector_add.argtypes = [ctypes.c_void_p, ctypes.c_char_p, P, P, ctypes.c_int64, ctypes.c_int64]
    h.vgh_connector_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_size_t]
    h._chain_bound = True
    return h


class BiGraph:
    def __init__(self, graph):
        self.h = _bind()
        self.ptr = self.h.vgh_bigraph_create()
        for nid, seq in graph["nodes"]:
            assert self.h.vgh_bigraph_add_node(self.ptr, nid, seq.encode()) == 0, self.h.vgh_last_error()
        for a, a_start, b, b_end in graph["edges"]:
            assert self.h.vgh_bigraph_add_edge(self.ptr, a, int(a_start), b, int(b_end)) == 0, self.h.vgh_last_error()
        self.length = {nid: len(seq) for nid, seq in graph["nodes"]}

    def __del__(self):
        if getattr(self, "ptr", None):
            self.h.vgh_bigraph_destroy(self.ptr)


def _pos(p):
    return (ctypes.c_int64 * 3)(*(p if p else (0, 0, 0)))


def _annotate(aln):
    for m in aln["path"]["mapping"]:
        m["n_edits"] = len(m["edit"])
    return aln


def align_between(aligner, graph, read, left, right, max_path_length, max_gap_length, consistently=False, max_dp_cells=-1):
    h = _bind()
    buf = ctypes.create_string_buffer(1 << 22)
    rc = h.vgh_align_sequence_between(aligner.ptr, graph.ptr, read.encode(), _pos(left), _pos(right), max_path_length, max_gap_length, int(consistently),
                                      max_dp_cells, buf, len(buf))
    if rc != 0:
        raise RuntimeError("rc %d: %s" % (rc, h.vgh_last_error().decode()))
    out = json.loads(buf.value.decode())
    return _annotate(out["alignment"]), out["did_align"]


def gap_in_range(aligner, length, begin, end):
    return _bind().vgh_longest_detectable_gap_in_range(aligner.ptr, length, begin, end)


def _numbers(case, aligner):
    """the call's max_path_length and max_gap_length (one case computes them from the read, :647-648)"""
    L = len(case["sequence"])
    gap = case["max_gap_length"]
    if gap == "gap_in_range":
        gap = gap_in_range(aligner, L, 0, L)
    path = case["max_path_length"]
    if path == "len+gap":
        path = L + gap
    return path, gap


def run_direct_cases(engine_lib):
    aligner = HostAligner(engine_lib)
    n = 0
    for case in CASES:
        if case["call"] != "align_sequence_between":
            continue
        graph = BiGraph(case["graph"])
        path, gap = _numbers(case, aligner)
        # ("can align a long tail", :682: a 4.4 kbp tail through pinned X-drop — beyond the packed kernels' 1024 rows, so the engine runs
        #  it on its wide route, vg_amd/csrc/gssw_wide_device.hpp)
        aln, did = align_between(aligner, graph, case["sequence"], case["left"], case["right"], path, gap)
        assert did
        for req in case["requires"]:
            assert eval(req, {"aln": aln, "max": max, "len": len}), "%s [%s]: %s\n%s" % (case["source"], case["name"], req, json.dumps(aln)[:600])
            n += 1
    return n


def same_alignment(a, b):
    """require_alignments_equal (:712-728): node ids, offsets, edits (not the strand flag)"""
    ma, mb = a["path"]["mapping"], b["path"]["mapping"]
    assert len(ma) == len(mb), (a, b)
    for x, y in zip(ma, mb):
        assert x["position"]["node_id"] == y["position"]["node_id"] and x["position"]["offset"] == y["position"]["offset"], (a, b)
        assert [(e["from_length"], e["to_length"], e["sequence"]) for e in x["edit"]] == [(e["from_length"], e["to_length"], e["sequence"]) for e in y["edit"]], (a, b)


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


def flip_alignment(aln, length):
    """reverse_complement_alignment in python, for the consistency test"""
    out = {"path": {"mapping": []}}
    for m in reversed(aln["path"]["mapping"]):
        used = sum(e["from_length"] for e in m["edit"])
        p = m["position"]
        out["path"]["mapping"].append({"position": {"node_id": p["node_id"], "is_reverse": not p["is_reverse"], "offset": length[p["node_id"]] - used - p["offset"]},
                                       "edit": [{"from_length": e["from_length"], "to_length": e["to_length"], "sequence": revcomp(e["sequence"])} for e in reversed(m["edit"])]})
    return out


def run_consistency_case(engine_lib):
    aligner = HostAligner(engine_lib)
    case = next(c for c in CASES if c["call"] == "align_sequence_between_consistently")
    graph = BiGraph(case["graph"])
    for seq in case["sequence"]:
        fwd, _ = align_between(aligner, graph, seq, case["left"], case["right"], case["max_path_length"], case["max_gap_length"], consistently=True)
        rev, _ = align_between(aligner, graph, revcomp(seq), case["rev_left"], case["rev_right"], case["max_path_length"], case["max_gap_length"], consistently=True)
        same_alignment(flip_alignment(rev, graph.length), fwd)
        assert sum(e["to_length"] for m in fwd["path"]["mapping"] for e in m["edit"]) == len(seq)
    return len(case["sequence"])


def run_dagified_case():
    h = _bind()
    case = next(c for c in CASES if c["call"] == "with_dagified_local_graph")
    graph = BiGraph(case["graph"])
    buf = ctypes.create_string_buffer(1 << 20)
    assert h.vgh_dagified_local_graph(graph.ptr, _pos(case["left"]), _pos(case["right"]), case["max_path_length"], buf, len(buf)) == 0, h.vgh_last_error()
    d = json.loads(buf.value.decode())
    base = {n[0]: (n[2], bool(n[3])) for n in d["nodes"]}
    length = {n[0]: len(n[1]) for n in d["nodes"]}
    env = {"head_tip_bases": [base[t[0]] for t in d["tips"] if not t[1]], "n_tips": len(d["tips"]), "left_anchor_is_tip": d["left_anchor"] in d["tips"],
           "left_anchor_length": length[d["left_anchor"][0]], "all": all}
    for req in case["requires"]:
        assert eval(req, env), (req, d)
    # the stick stays a stick: the cut anchor, then the two nodes the 50 bases reach (4 + 14 + 14 + 14 bases)
    assert [n[1] for n in d["nodes"]] == ["TACA", "GATTACAGATTACA", "GATTACAGATTACA"] and d["edges"] == [[1, 2], [2, 3]]


def test_longest_detectable_gap_in_range():
    case = next(c for c in CASES if c["call"] == "longest_detectable_gap_in_range")
    aligner = HostAligner(ORACLE_LIB)
    L = len(case["sequence"])
    env = {name: gap_in_range(aligner, L, lo, hi) for name, (lo, hi) in case["ranges"].items()}
    for req in case["requires"]:
        assert eval(req, env), (req, env)
    assert env["whole_sequence_gap"] == (1 * (L // 2) + 5 - 6) // 1 + 1            # src/alignment_scorer.cpp:264-271 at the middle


def test_reference_cases_on_the_oracle():
    assert len([c for c in CASES if c["call"] == "align_sequence_between"]) >= 12
    assert run_direct_cases(ORACLE_LIB) >= 100


def test_reference_cases_on_the_emulated_kernels(emu_lib):
    assert run_direct_cases(emu_lib) >= 100


def test_consistent_alignments_on_the_oracle():
    assert run_consistency_case(ORACLE_LIB) == 5


def test_dagified_local_graph_without_extraneous_tips():
    run_dagified_case()


# ---- beyond the reference's vectors: graphs with cycles and reversing edges, and the batched form ----------------------------------------
def random_bigraph(rng, n_nodes, p_back=0.1, p_rev=0.1):
    nodes = [[i + 1, "".join(rng.choice(list("ACGT"), int(rng.integers(1, 12))))] for i in range(n_nodes)]
    edges = []
    for i in range(1, n_nodes):
        edges.append([i, False, i + 1, False])                                  # a backbone, so that anchors connect
        if i + 2 <= n_nodes and rng.random() < 0.4:
            edges.append([i, False, i + 2, False])
        if rng.random() < p_back:
            edges.append([i + 1, False, int(rng.integers(1, i + 1)), False])    # a cycle
        if rng.random() < p_rev:
            edges.append([i, False, int(rng.integers(1, n_nodes + 1)), True])   # onto a reverse strand
    return {"nodes": nodes, "edges": edges}


def walk(rng, graph, start, steps):
    """a random walk along forward-to-forward edges from node `start`: the bases it spells and its nodes"""
    succ = {}
    for a, ar, b, br in graph["edges"]:
        if not ar and not br:
            succ.setdefault(a, []).append(b)
    seq = dict(graph["nodes"])
    path = [start]
    while len(path) < steps and succ.get(path[-1]):
        path.append(int(rng.choice(succ[path[-1]])))
    return path, "".join(seq[v] for v in path)


def score_of(aln, graph, read, scores=(1, 4, 6, 1)):
    """the score of an alignment re-computed from its edits (no full-length bonus: global / as reported otherwise)"""
    match, mismatch, go, ge = scores
    s = 0
    for m in aln["path"]["mapping"]:
        for e in m["edit"]:
            f, t = e["from_length"], e["to_length"]
            if f == t:
                s += f * match if e["sequence"] == "" else -mismatch * f
            elif f == 0 or t == 0:
                s -= go + (max(f, t) - 1) * ge
    return s


def connector_equals_direct(engine_lib, n_graphs=12, per_graph=10, seed=5):
    h = _bind()
    rng = np.random.default_rng(seed)
    aligner = HostAligner(engine_lib)
    total = aligned = 0
    for _ in range(n_graphs):
        graph_def = random_bigraph(rng, int(rng.integers(6, 30)))
        graph = BiGraph(graph_def)
        seq = dict(graph_def["nodes"])
        conn = h.vgh_connector_create(aligner.ptr, graph.ptr, -1)
        requests = []
        for _ in range(per_graph):
            a = int(rng.integers(1, len(seq)))
            path, bases = walk(rng, graph_def, a, int(rng.integers(2, 8)))
            kind = rng.random()
            off_a = int(rng.integers(0, len(seq[path[0]]) + 1)); off_b = int(rng.integers(0, len(seq[path[-1]]) + 1))
            inner = bases[off_a:len(bases) - (len(seq[path[-1]]) - off_b)] if len(path) > 1 else ""
            read = "".join(c if rng.random() > 0.05 else "ACGT"[int(rng.integers(0, 4))] for c in inner) or "A"
            left = [path[0], False, off_a]; right = [path[-1], False, off_b]
            if kind < 0.2:
                right = None
            elif kind < 0.4:
                left = None
            requests.append((read, left, right, len(read) + 20, 10))
            h.vgh_connector_add(conn, read.encode(), _pos(left), _pos(right), len(read) + 20, 10)
        buf = ctypes.create_string_buffer(1 << 22)
        ms = (ctypes.c_double * 3)()
        assert h.vgh_connector_run(conn, 3, ms, buf, len(buf)) == 0, h.vgh_last_error()
        got = json.loads(buf.value.decode())
        h.vgh_connector_destroy(conn)
        for (read, left, right, mp, mg), g in zip(requests, got):
            total += 1
            try:
                aln, did = align_between(aligner, graph, read, left, right, mp, mg)
            except RuntimeError as e:
                assert g["status"] != 0, (str(e), g)
                continue
            assert g["status"] == 0 and g["did_align"] == did, g
            assert _annotate(g["alignment"])["path"] == aln["path"] and g["alignment"]["score"] == aln["score"], (g["alignment"], aln)
            aligned += 1
            # whatever route it took, the alignment consumes the whole read, and between two anchors its score is the score of its edits
            assert sum(e["to_length"] for m in aln["path"]["mapping"] for e in m["edit"]) == len(read)
            if left and right and aln["path"]["mapping"]:
                assert aln["score"] == score_of(aln, graph_def, read), (aln, read)
    assert aligned > total // 2
    return total


def test_connector_answers_like_the_direct_call_on_the_oracle():
    assert connector_equals_direct(ORACLE_LIB) == 120


def test_connector_answers_like_the_direct_call_on_the_emulated_kernels(emu_lib):
    connector_equals_direct(emu_lib, n_graphs=6, seed=6)


@pytest.mark.gpu
def test_reference_cases_on_the_gpu():
    assert run_direct_cases(ENGINE_LIB) >= 100
    assert run_consistency_case(ENGINE_LIB) == 5


@pytest.mark.gpu
def test_connector_answers_like_the_direct_call_on_the_gpu():
    connector_equals_direct(ENGINE_LIB, n_graphs=20, per_graph=25, seed=7)
