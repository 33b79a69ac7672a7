"""The graph algorithms behind N1 against the reference's own known-answer tests of them.

vg_amd/host/local_graph.cpp states extract_connecting_graph, extract_extending_graph, extract_containing_graph (behaviour of
src/algorithms/extract_*.cpp) and — from their contracts, libhandlegraph being an empty submodule of the snapshot — dagify, dagify_from,
split_strands, find_tips.  tests/golden/ref_graph_algorithms.json holds what src/unittest/vg_algorithms.cpp and src/unittest/dagify.cpp
REQUIRE of them, transcribed by tests/golden/extract_graph_algorithm_tests.py; every case runs through the host shim's C entry
vgh_graph_algorithm.  Host-only code: nothing here needs a GPU.
"""
import ctypes
import itertools
import json
import os

import numpy as np
import pytest

import util

CASES = json.load(open(os.path.join(util.ROOT, "tests", "golden", "ref_graph_algorithms.json")))["cases"]


def host():
    lib = ctypes.CDLL(util.HOST_LIB)
    lib.vgh_bigraph_create.restype = ctypes.c_void_p
    lib.vgh_bigraph_destroy.argtypes = [ctypes.c_void_p]
    lib.vgh_bigraph_add_node.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
    lib.vgh_bigraph_add_edge.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    lib.vgh_graph_algorithm.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.vgh_last_error.restype = ctypes.c_char_p
    return lib


def run(lib, graph, call, args):
    g = lib.vgh_bigraph_create()
    try:
        for nid, seq in graph["nodes"]:
            assert lib.vgh_bigraph_add_node(g, nid, seq.encode()) == 0
        for a, a_start, b, b_end in graph["edges"]:
            assert lib.vgh_bigraph_add_edge(g, a, int(a_start), b, int(b_end)) == 0
        arr = np.array(args, dtype=np.int64)
        buf = ctypes.create_string_buffer(1 << 20)
        rc = lib.vgh_graph_algorithm(g, call.encode(), arr.ctypes.data, len(arr), buf, len(buf))
        assert rc == 0, lib.vgh_last_error()
        return json.loads(buf.value.decode())
    finally:
        lib.vgh_bigraph_destroy(g)


def successors(out):
    """oriented node (id, is_reverse) -> the oriented nodes its right side leads to"""
    nxt = {}
    for a, a_start, b, b_end in out["edges"]:
        nxt.setdefault((a, bool(a_start)), []).append((b, bool(b_end)))
        nxt.setdefault((b, not b_end), []).append((a, not a_start))
    return nxt


@pytest.mark.parametrize("case", [c for c in CASES if c["call"].startswith("extract")], ids=lambda c: c["source"].split("/")[-1])
def test_extraction_has_the_reference_properties(case):
    out = run(host(), case["graph"], case["call"], case["args"])
    f = case["facts"]
    nodes = out["nodes"]                                    # [id, sequence, source id, source is_reverse]
    full = {nid: seq for nid, seq in case["graph"]["nodes"]}
    if "n_nodes" in f:
        assert len(nodes) == f["n_nodes"], (case["source"], nodes)
    if "n_edges" in f:
        assert len(out["edges"]) == f["n_edges"], (case["source"], out["edges"])
    if "only_sequence" in f:
        assert [n[1] for n in nodes] == [f["only_sequence"]]
    sources = [n[2] for n in nodes]
    if "sources_within" in f:
        assert set(sources) <= set(f["sources_within"]), (case["source"], sources)
        if f.get("n_nodes") == len(f["sources_within"]):
            assert sorted(sources) == f["sources_within"]
    if "retained" in f:
        assert set(f["retained"]) <= set(sources)
    if f.get("no_duplicates"):
        assert len(set(sources)) == len(sources)
    if "n_retained" in f:
        assert len(set(sources)) == f["n_retained"], (case["source"], nodes)
    for n in nodes:
        want = f.get("sequence_of", {}).get(str(n[2]))
        if want is not None:
            assert n[1] in want, (case["source"], n, want)
        elif "other_sequence" in f and "sequence_of" in f:
            assert n[1] == f["other_sequence"], (case["source"], n)
        # whatever the section says, an extracted node carries a piece of the node it stands for
        assert n[1] in full[n[2]], (case["source"], n)
    for s in f.get("sequences_include", []):
        assert s in [n[1] for n in nodes], (case["source"], s, nodes)
    if "node_ids" in f:
        assert sorted(n[0] for n in nodes) == f["node_ids"]
    # every edge joins nodes of the extracted graph
    ids = {n[0] for n in nodes}
    assert all(e[0] in ids and e[2] in ids for e in out["edges"])


def test_strict_connecting_graph_leaves_only_the_two_anchor_tips():
    """src/unittest/vg_algorithms.cpp:1086-1116 ("a cool loop"): at least the two tips without strict_max_len, exactly the two with it."""
    case = [c for c in CASES if c["source"].endswith(":1086")][0]
    loose = run(host(), case["graph"], "extract_connecting", case["args"][:-1] + [0])
    strict = run(host(), case["graph"], "extract_connecting", case["args"][:-1] + [1])
    assert len(loose["tips"]) >= 2 and len(strict["tips"]) == 2


@pytest.mark.parametrize("case", [c for c in CASES if c["call"] == "dagify"], ids=lambda c: c["source"].split("/")[-1])
def test_dagify_has_the_reference_properties(case):
    """src/unittest/dagify.cpp:22-395: acyclic, the stated number of copies (6 / 8 / 6 nodes — this pins how far a cycle is unrolled),
    one copy of the nodes outside the cycle, and every listed walk present with its nodes' orientations."""
    out = run(host(), case["graph"], "dagify", case["args"])
    f = case["facts"]
    assert out["acyclic"] and len(out["nodes"]) == f["n_nodes"], out["nodes"]
    full = {nid: seq for nid, seq in case["graph"]["nodes"]}
    assert all(n[1] == full[n[2]] and n[3] == 0 for n in out["nodes"])           # copies keep the node's forward sequence
    src = {n[0]: n[2] for n in out["nodes"]}
    nxt = successors(out)
    for walk in f["walks"]:
        found = False
        for n in out["nodes"]:
            for rev in (False, True):
                stack = [[(n[0], rev)]]
                while stack and not found:
                    w = stack.pop()
                    if len(w) == len(walk):
                        found = all(src[a[0]] == b[0] and a[1] == bool(b[1]) for a, b in zip(w, walk))
                        continue
                    stack.extend(w + [x] for x in nxt.get(w[-1], []))
        assert found, (case["source"], walk)


@pytest.mark.parametrize("case", [c for c in CASES if c["call"] == "dagify_from"], ids=lambda c: c["source"].split("/")[-1])
def test_dagify_from_avoids_extraneous_tips(case):
    """src/unittest/dagify.cpp:424-515"""
    out = run(host(), case["graph"], "dagify_from", case["args"])
    f = case["facts"]
    assert out["acyclic"]
    src = {n[0]: n[2] for n in out["nodes"]}
    heads = [t for t in out["tips"] if not t[1]]
    tails = [[t[0], 0] for t in out["tips"] if t[1]]                            # (a reverse tip handle is a tail, read forward)
    for side, tips in (("heads", heads), ("tails", tails)):
        want = f[side]
        if "count" in want:
            assert len(tips) == want["count"], (side, tips)
        assert all(src[t[0]] in want["sources"] for t in tips), (side, tips, src)
    assert len(out["starts"]) == 1
    s = out["starts"][0]
    if f["start_is"] == "head":
        assert [s[0], s[1]] in [[h[0], 0] for h in heads]
    else:
        assert s[1] == 1 and [s[0], 0] in tails


def test_dagify_random_graphs_both_ways_of_splitting_strands():
    """src/unittest/dagify.cpp:397-422 in spirit: random bidirected graphs with reversing edges, strands split, dagified to 15 bases —
    acyclic, and every walk of up to 15 bases of the split graph is a walk of the result."""
    rng = np.random.default_rng(11)
    lib = host()
    for trial in range(60):
        n = int(rng.integers(2, 7))
        graph = {"nodes": [[i + 1, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 6))))] for i in range(n)], "edges": []}
        for _ in range(int(rng.integers(1, 2 * n))):
            graph["edges"].append([int(rng.integers(1, n + 1)), bool(rng.random() < 0.25), int(rng.integers(1, n + 1)), bool(rng.random() < 0.25)])
        split = run(lib, graph, "split_strands", [])
        assert split["single_stranded"] and len(split["nodes"]) == 2 * n
        sgraph = {"nodes": [[x[0], x[1]] for x in split["nodes"]], "edges": split["edges"]}
        dag = run(lib, sgraph, "dagify", [15])
        assert dag["acyclic"]
        src = {x[0]: x[2] for x in dag["nodes"]}; length = {x[0]: len(x[1]) for x in dag["nodes"]}
        s_next = successors(split); d_next = successors(dag)
        slen = {x[0]: len(x[1]) for x in split["nodes"]}
        # walks of the split graph, as sequences of node ids, while the bases after the first node stay under 15
        def walks_from(nxt, start, lens):
            out, stack = set(), [([start], 0)]
            while stack:
                w, spent = stack.pop()
                out.add(tuple(w))
                for x in nxt.get((w[-1], False), []):
                    if spent < 15 and len(w) < 7:
                        stack.append((w + [x[0]], spent + lens[x[0]]))
            return out
        want = set().union(*[walks_from(s_next, x[0], slen) for x in split["nodes"]])
        have = set()
        for x in dag["nodes"]:
            have |= {tuple(src[v] for v in w) for w in walks_from(d_next, x[0], length)}
        assert want <= have, (trial, sorted(want - have)[:3])


# ---- `vg map`'s side of N1: Mapper::align_to_graph over the same algorithms (vg_amd/host/cluster_alignment.hpp) -----------------------

def _align_to_graph(aligner, graph, read, do_flip=False, traceback=True, pinned=False, pin_left=False, banded=False, keep_bonuses=True):
    lib = host()
    lib.vgh_align_to_graph.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    g = lib.vgh_bigraph_create()
    try:
        for nid, seq in graph["nodes"]:
            assert lib.vgh_bigraph_add_node(g, nid, seq.encode()) == 0
        for a, a_start, b, b_end in graph["edges"]:
            assert lib.vgh_bigraph_add_edge(g, a, int(a_start), b, int(b_end)) == 0
        flags = (1 if do_flip else 0) | (2 if traceback else 0) | (4 if pinned else 0) | (8 if pin_left else 0) | (16 if banded else 0) | (32 if keep_bonuses else 0)
        buf = ctypes.create_string_buffer(1 << 20)
        assert lib.vgh_align_to_graph(aligner.ptr, g, read.encode(), flags, buf, len(buf)) == 0, lib.vgh_last_error()
        return json.loads(buf.value.decode())
    finally:
        lib.vgh_bigraph_destroy(g)


def _walk(aln):
    return [(m["position"]["node_id"], bool(m["position"].get("is_reverse", False))) for m in aln["path"]["mapping"]]


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def align_to_graph_cases(engine_lib):
    al = util.HostAligner(engine_lib)
    # a cycle (a self loop) is unrolled far enough for the read: three turns of node 2 come back as three mappings on node 2
    cyc = {"nodes": [[1, "GATTACA"], [2, "CATTAG"], [3, "AGAGAGAG"], [4, "CCC"]], "edges": [[1, False, 2, False], [2, False, 2, False], [2, False, 3, False], [2, False, 4, False]]}
    read = "GATTACA" + "CATTAG" * 3 + "AGAGAGAG"
    a = _align_to_graph(al, cyc, read)
    assert _walk(a) == [(1, False), (2, False), (2, False), (2, False), (3, False)]
    assert a["score"] == len(read) + 10                                     # every base matches, a full-length bonus at either end
    assert _align_to_graph(al, cyc, read, keep_bonuses=False)["score"] == len(read)
    # the read's reverse complement with do_flip: the same walk on the reverse strand, read backwards
    b = _align_to_graph(al, cyc, _revcomp(read), do_flip=True)
    assert b["score"] == a["score"] and _walk(b) == [(n, True) for n, _ in reversed(_walk(a))]
    # banded global and pinned through the same preparation
    c = _align_to_graph(al, cyc, "GATTACA" + "CATTAG" * 2 + "CCC", banded=True)
    assert _walk(c) == [(1, False), (2, False), (2, False), (4, False)]
    d = _align_to_graph(al, cyc, "CATTAGAGAGAGAG", pinned=True, pin_left=False)
    assert _walk(d)[-1] == (3, False)
    # a reversing edge: the strands are split apart, and a walk that turns around comes back with its orientations
    rev = {"nodes": [[1, "GATTACAGG"], [2, "CCATTAGCA"], [3, "TTGACGTTG"]], "edges": [[1, False, 2, False], [2, False, 3, True]]}
    r = _align_to_graph(al, rev, "GATTACAGG" + "CCATTAGCA" + _revcomp("TTGACGTTG"))
    assert _walk(r) == [(1, False), (2, False), (3, True)] and r["score"] == 27 + 10
    r2 = _align_to_graph(al, rev, _revcomp("GATTACAGG" + "CCATTAGCA" + _revcomp("TTGACGTTG")))
    assert _walk(r2) == [(3, False), (2, True), (1, True)] and r2["score"] == r["score"]
    # an acyclic one-strand graph goes through unchanged: the same answer as the aligner called on it directly
    dag = {"nodes": [[1, "GATTACA"], [2, "C"], [3, "T"], [4, "GGGACCA"]], "edges": [[1, False, 2, False], [1, False, 3, False], [2, False, 4, False], [3, False, 4, False]]}
    e = _align_to_graph(al, dag, "TTACATGGGA")
    assert _walk(e) == [(1, False), (3, False), (4, False)] and e["path"]["mapping"][0]["position"]["offset"] == 2


def test_align_to_graph_on_the_oracle_engine():
    align_to_graph_cases(util.ORACLE_LIB)


def test_align_to_graph_on_the_emulated_kernels():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    align_to_graph_cases(util.EMU_LIB)


@pytest.mark.gpu
def test_align_to_graph_on_the_hip_engine():
    align_to_graph_cases(util.ENGINE_LIB)


def test_cluster_subgraph_reaches_as_far_as_a_detectable_alignment_can():
    """cluster_subgraph_containing (src/cluster.cpp:3832-3851): from a seed at read offset b, forward longest_detectable_gap(end) + (L - b)
    bases and backward longest_detectable_gap(b) + b bases — on a chain of 10-base nodes exactly the nodes within those distances."""
    lib = host()
    lib.vgh_cluster_subgraph.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    al = util.HostAligner(util.ORACLE_LIB)
    n = 60
    g = lib.vgh_bigraph_create()
    try:
        for i in range(1, n + 1):
            assert lib.vgh_bigraph_add_node(g, i, b"ACGTACGTAC") == 0
        for i in range(1, n):
            assert lib.vgh_bigraph_add_edge(g, i, 0, i + 1, 0) == 0
        L, b, e = 50, 20, 32                                              # a 12-base seed at read offset 20, on node 30 offset 4
        seeds = np.array([b, e, 30, 0, 4], dtype=np.int64)
        buf = ctypes.create_string_buffer(1 << 20)
        assert lib.vgh_cluster_subgraph(al.ptr, g, L, seeds.ctypes.data, 1, buf, len(buf)) == 0, lib.vgh_last_error()
        got = sorted(x[0] for x in json.loads(buf.value.decode())["nodes"])
        gap = lambda pos: max(0, (1 * min(pos, L - pos) + 5 - 6) // 1 + 1) if min(pos, L - pos) > 0 else 0      # src/alignment_scorer.cpp:264-271 with 1/4/6/1/5
        forward = gap(e) + (L - b); backward = gap(b) + b
        start = 29 * 10 + 4                                                # the seed's first base along the chain
        want = [i for i in range(1, n + 1) if (i - 1) * 10 < start + forward and i * 10 > start - backward]
        assert got == want, (got, want)
    finally:
        lib.vgh_bigraph_destroy(g)
