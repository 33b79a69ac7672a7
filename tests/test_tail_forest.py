"""Tail forests (vgk_tail_forest): the haplotype-consistent subgraphs giraffe aligns read tails to
(MinimizerMapper::get_tail_forest / dfs_gbwt, src/minimizer_mapper.cpp:5745-5860, :5909-6013).

The reference holds no known-answer test for this path, so three constructions that share nothing are compared:
  * this file's: the TRIE OF THREAD CONTINUATIONS, built from the explicit thread lists — the threads (in either orientation) that
    pass through the start state, cut behind it, merged node by node; children entered in descending node order (the reference's
    stack pops the last follow_paths edge first), expanded while the bases used stay below the walk distance;
  * the oracle's (oracle/vgo_tail.c): dfs_gbwt restated over search states of the oracle's haplotype index;
  * the engine's (vg_amd/csrc/tail_device.hpp): the same walk on the device (here: the lock-step emulator; on the MI355X in the
    gpu tests), twice, with the forest left in HBM as one resident graph.
Then the forest graph is used the way giraffe uses it — every tree a left-pinned X-drop problem (get_best_alignment_against_any_tree,
:5626-5741) — through vgk_gssw_pack_windows, against the oracle on the same windows and against explicit per-tree graphs."""
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, workloads
from test_windows import assert_same


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


# ---- the independent construction ------------------------------------------------------------------------------------------------
def both_orientations(threads):
    out = []
    for t in threads:
        t = [int(x) for x in t]
        out.append(t)
        out.append([x ^ 1 for x in reversed(t)])
    return out


def continuations_of_node(all_threads, node):
    """what follows every visit of `node` (the state of get_state(handle): all its visits)"""
    return [t[k + 1:] for t in all_threads for k, x in enumerate(t) if x == node]


def continuations_of_path(all_threads, path):
    """what follows every traversal of `path` (the forward state of an extension along it)"""
    out = []
    m = len(path)
    for t in all_threads:
        for k in range(len(t) - m + 1):
            if t[k:k + m] == path:
                out.append(t[k + m:])
    return out


def trie_forest(lens, node, conts, offset, walk):
    """-> [(parent, node, length)] in entry order; parent = index in this list or -1"""
    out = []
    if not conts and conts is not None:
        return out

    def visit(v, cs, used, parent, is_root):
        remaining = lens[v] - offset if is_root else lens[v]
        hidden = is_root and remaining == 0
        me = parent
        if not hidden:
            out.append((parent, v, remaining))
            me = len(out) - 1
        else:
            me = -1
        used += remaining
        if used < walk:
            for w in sorted({c[0] for c in cs if c}, reverse=True):
                visit(w, [c[1:] for c in cs if c and c[0] == w], used, me, False)

    visit(node, conts, 0, -1, True)
    return out


def visit_counts(all_threads, n_oriented):
    c = np.zeros(n_oriented, dtype=np.int64)
    for t in all_threads:
        for x in t:
            c[x] += 1
    return c


def small_workload(seed, n_haplotypes=6, graph_bp=3000):
    wl = workloads.GaplessWorkload(8, seed=seed, graph_bp=graph_bp, n_haplotypes=n_haplotypes, snp_every=25, indel_every=120)
    lens = np.repeat(np.array([len(s) for s in wl.nodes]), 2)
    return wl, lens, both_orientations(wl.threads)


def whole_node_problems(rng, lens, counts, n):
    probs = []
    for _ in range(n):
        o = int(rng.integers(0, len(lens)))
        while counts[o] == 0:
            o = int(rng.integers(0, len(lens)))
        off = int(rng.integers(0, lens[o] + 1)) if rng.random() < 0.8 else int(lens[o])
        probs.append((o, 0, int(counts[o]) - 1, off, int(rng.integers(1, 160))))
    return probs


def check_forest(eng, index, probs, expected):
    res, forest = eng.tail_forest(index, probs)
    parent, node, length = forest.fetch()
    at = 0
    for i, exp in enumerate(expected):
        r = res[i]
        assert r["status"] == 0, (i, r)
        assert r["first_node"] == at and r["n_nodes"] == len(exp), (i, r, len(exp))
        got = [(int(parent[at + k]) - at if parent[at + k] >= 0 else -1, int(node[at + k]), int(length[at + k])) for k in range(len(exp))]
        assert got == exp, "problem %d %s: forest differs\n got %s\n exp %s" % (i, probs[i], got[:12], exp[:12])
        assert r["n_trees"] == sum(1 for e in exp if e[0] < 0)
        assert r["bases"] == sum(e[2] for e in exp)
        at += len(exp)
    assert forest.size == at
    return res, forest, (parent, node, length)


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_forest_of_all_visits_equals_the_trie_of_thread_continuations(lib_name, emu_lib):
    lib = ORACLE_LIB if lib_name == "oracle" else emu_lib
    for seed in (1, 2):
        wl, lens, allt = small_workload(seed)
        counts = visit_counts(allt, len(lens))
        rng = np.random.default_rng(seed)
        probs = whole_node_problems(rng, lens, counts, 150)
        expected = [trie_forest(lens, p[0], continuations_of_node(allt, p[0]), p[3], p[4]) for p in probs]
        eng = capi.Engine(lib=lib)
        index = eng.haplo_index(wl.nodes, wl.threads)
        check_forest(eng, index, probs, expected)
        assert any(len(e) > 8 for e in expected) and any(sum(1 for x in e if x[0] < 0) > 1 for e in expected)      # branching trees, and forests behind a skipped root


def extension_tail_problems(wl, lens, allt, eng, index, rng, n):
    """Tails the way giraffe gets them: gapless extensions that stop short of a read end (reads with an insertion in the middle),
    the walk started from the extension's search state."""
    nodes = wl.nodes
    fw = [np.frombuffer("".join(nodes[o >> 1] if not (o & 1) else revcomp(nodes[o >> 1]) for o in t).encode(), dtype=np.uint8) for t in allt]
    starts = [np.concatenate([[0], np.cumsum([lens[o] for o in t])]) for t in allt]
    problems = []
    for _ in range(n):
        ti = int(rng.integers(0, len(allt)))
        L = 120
        if len(fw[ti]) < L + 40:
            continue
        a = int(rng.integers(20, len(fw[ti]) - L - 20))
        rd = fw[ti][a:a + L].copy()
        cut = int(rng.integers(40, 80))
        rd = np.concatenate([rd[:cut], np.frombuffer(b"ACGTACG"[:int(rng.integers(2, 6))], dtype=np.uint8), rd[cut:]])[:L]
        # a seed in the first 30 bases, at its true position
        ro = int(rng.integers(0, 30)); g = a + ro
        k = int(np.searchsorted(starts[ti], g, side="right") - 1)
        problems.append(dict(read=rd.tobytes().decode(), seeds=[(allt[ti][k], ro - (g - int(starts[ti][k])))], trim=False))
    res, ext, enodes, _ = eng.gapless_extend(index, problems)
    tails, expected, meta = [], [], []
    for i, r in enumerate(res):
        for e in ext[r["ext_begin"]:r["ext_begin"] + r["n_ext"]]:
            path = [int(x) for x in enodes[e["path_begin"]:e["path_begin"] + e["path_len"]]]
            L = len(problems[i]["read"])
            matched = int(e["read_end"]) - int(e["read_begin"])
            if not e["right_full"]:
                end_off = int(e["offset"]) + matched - int(sum(lens[o] for o in path[:-1]))
                tail_len = L - int(e["read_end"])
                walk = workloads.longest_detectable_gap(L, tail_len, 1, 6, 1, 5) + tail_len
                st = e["state"]
                assert int(st[0]) == path[-1]
                tails.append((int(st[0]), int(st[1]), int(st[2]), end_off, walk))
                expected.append(trie_forest(lens, path[-1], continuations_of_path(allt, path), end_off, walk))
                meta.append((i, True, int(e["read_end"]), tail_len))
            if not e["left_full"]:
                first = path[0] ^ 1
                off = int(lens[first]) - int(e["offset"])
                tail_len = int(e["read_begin"])
                walk = workloads.longest_detectable_gap(L, tail_len, 1, 6, 1, 5) + tail_len
                st = e["state"]
                assert int(st[3]) == first
                tails.append((int(st[3]), int(st[4]), int(st[5]), off, walk))
                expected.append(trie_forest(lens, first, continuations_of_path(allt, [x ^ 1 for x in reversed(path)]), off, walk))
                meta.append((i, False, int(e["read_begin"]), tail_len))
    return problems, tails, expected, meta


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def tree_windows(res, parent, reads, meta, problems, tails):
    """one left-pinned X-drop window problem per TREE: the tail sequence (reverse-complemented for a left tail, :5660) against the
    tree's run of nodes"""
    seqs, first, count, gaps, owner = [], [], [], [], []
    for i, r in enumerate(res):
        if r["n_nodes"] == 0:
            continue
        ri, right, pos, tail_len = meta[i]
        rd = problems[ri]["read"]
        seq = rd[pos:] if right else revcomp(rd[:pos])
        lo, hi = int(r["first_node"]), int(r["first_node"] + r["n_nodes"])
        roots = [v for v in range(lo, hi) if parent[v] < 0] + [hi]
        for a, b in zip(roots[:-1], roots[1:]):
            seqs.append(np.frombuffer(seq.encode(), dtype=np.uint8)); first.append(a); count.append(b - a)
            gaps.append(workloads.longest_detectable_gap(len(rd), tail_len, 1, 6, 1, 5)); owner.append(i)
    read_off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    return np.concatenate(seqs), read_off, np.array(first), np.array(count), np.array(gaps), owner


def run_tails_through_forests(lib, seed, n_reads):
    wl, lens, allt = small_workload(seed, graph_bp=6000)
    sc = capi.Scoring.simple(1, 4, 6, 1, 5)
    eng = capi.Engine(sc, lib=lib); ora = capi.Engine(sc, lib=ORACLE_LIB)
    index = eng.haplo_index(wl.nodes, wl.threads); oindex = ora.haplo_index(wl.nodes, wl.threads)
    rng = np.random.default_rng(seed)
    problems, tails, expected, meta = extension_tail_problems(wl, lens, allt, eng, index, rng, n_reads)
    assert len(tails) > n_reads // 2
    res, forest, (parent, node, length) = check_forest(eng, index, tails, expected)
    ores, oforest, _ = check_forest(ora, oindex, tails, expected)
    reads, read_off, first, count, gaps, owner = tree_windows(res, parent, None, meta, problems, tails)
    flags = capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK
    cols = np.array([int(length[a:a + k].sum()) for a, k in zip(first, count)])
    ws = capi.WindowSet(reads, read_off, first, count, flags, gaps, cols=cols)
    ra, oa = eng.align_windows(forest.graph, ws, 0)
    rb, ob = ora.align_windows(oforest.graph, ws, 0)
    assert_same(ra, oa, rb, ob, "tails against their trees: engine vs oracle")
    # the same problems as explicit per-tree graphs (bases from the node strings, the root's behind its cut)
    plist = []
    for w, (a, k) in enumerate(zip(first, count)):
        trim = int(res[owner[w]]["root_trim"])
        seqs = []
        for v in range(a, a + k):
            o = int(node[v]); s = wl.nodes[o >> 1] if not (o & 1) else revcomp(wl.nodes[o >> 1])
            seqs.append(s[trim:] if v == a and parent[v] < 0 and trim else s)
            assert len(seqs[-1]) == length[v]
        plist.append(dict(read=bytes(reads[read_off[w]:read_off[w + 1]]).decode(), nodes=seqs,
                          preds=[[int(parent[v]) - a] if parent[v] >= a else [] for v in range(a, a + k)], flags=flags, pinning=None, max_gap=int(gaps[w])))
    rp, op = eng.align(capi.ProblemSet.from_lists(plist), 0)
    assert_same(ra, oa, rp, op, "tails against their trees: windows of the forest graph vs explicit graphs")
    assert (ra["status"] == 0).all()
    # a tail follows a haplotype behind the insertion: most of it aligns
    tl = np.diff(read_off)
    assert (ra["score"] >= tl - 12).mean() > 0.7, (ra["score"][:10], tl[:10])
    return len(tails), len(first)


def test_tails_of_gapless_extensions_through_their_forests(emu_lib):
    n_tails, n_trees = run_tails_through_forests(emu_lib, 5, 120)
    assert n_trees >= n_tails * 0.9


def test_tail_problem_errors(emu_lib):
    wl, lens, allt = small_workload(3)
    counts = visit_counts(allt, len(lens))
    o = int(np.argmax(counts > 0))
    for lib in (emu_lib, ORACLE_LIB):
        eng = capi.Engine(lib=lib)
        index = eng.haplo_index(wl.nodes, wl.threads)
        probs = [(len(lens) + 3, 0, 0, 0, 50),                       # no such node
                 (o, 0, int(counts[o]) - 1, int(lens[o]) + 1, 50),      # cut behind the node
                 (o, 0, int(counts[o]), 0, 50),                         # range beyond the node's visits
                 (o, 1, 0, 0, 50),                                      # empty state: no trees, no error
                 (o, 0, int(counts[o]) - 1, 0, 50)]
        res, forest = eng.tail_forest(index, probs)
        assert list(res["status"][:3]) == [-1, -1, -1] or (res["status"][:3] < 0).all()
        assert res["status"][3] == 0 and res["n_nodes"][3] == 0
        assert res["status"][4] == 0 and res["n_nodes"][4] > 0 and res["first_node"][4] == 0
        assert forest.size == res["n_nodes"][4]
        r0, f0 = eng.tail_forest(index, [])
        assert f0.size == 0


@pytest.mark.gpu
def test_forests_on_the_gpu_equal_the_trie_of_thread_continuations():
    wl, lens, allt = small_workload(11, n_haplotypes=8, graph_bp=20000)
    counts = visit_counts(allt, len(lens))
    rng = np.random.default_rng(11)
    probs = whole_node_problems(rng, lens, counts, 3000)
    expected = [trie_forest(lens, p[0], continuations_of_node(allt, p[0]), p[3], p[4]) for p in probs]
    eng = capi.Engine(lib=ENGINE_LIB)
    check_forest(eng, eng.haplo_index(wl.nodes, wl.threads), probs, expected)


@pytest.mark.gpu
def test_tails_of_gapless_extensions_through_their_forests_on_the_gpu():
    n_tails, n_trees = run_tails_through_forests(ENGINE_LIB, 12, 1500)
    assert n_tails > 700


def test_a_walk_deeper_than_the_kernels_stack_is_declined(emu_lib):
    """600 one-base nodes in a row and a walk of 1000 bases: the path alone is deeper than the kernel's 512 frames -> VGK_ETOOBIG for
    that tail only; the oracle (a stack that grows) answers it; the other tails of the batch are untouched."""
    nodes = ["ACGT"[i % 4] for i in range(600)] + ["ACGTACGTAC"]
    threads = [[2 * i for i in range(601)]]
    probs = [(0, 0, 0, 0, 40), (0, 0, 0, 0, 1000), (2 * 10, 0, 0, 0, 25)]
    eng = capi.Engine(lib=emu_lib); ora = capi.Engine(lib=ORACLE_LIB)
    res, forest = eng.tail_forest(eng.haplo_index(nodes, threads), probs)
    ores, oforest = ora.tail_forest(ora.haplo_index(nodes, threads), probs)
    assert list(ores["status"]) == [0, 0, 0] and ores["n_nodes"][1] == 601
    assert res["status"][1] == -7 and res["n_nodes"][1] == 0
    for i in (0, 2):
        assert res["status"][i] == 0 and res["n_nodes"][i] == ores["n_nodes"][i] == (40 if i == 0 else 25)
    assert res["first_node"][2] == res["n_nodes"][0] and forest.size == 65
    p, n, l = forest.fetch()
    assert list(n[:40]) == [2 * i for i in range(40)] and list(p[:40]) == [-1] + list(range(39))


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_forests_over_run_length_records_and_nodes_with_many_edges(lib_name, emu_lib):
    """300 haplotypes (the index stores the visit bodies run-length encoded) and a site with six alleles (more than the four edges the
    one-pass count handles): the other two branches of the kernel's follow step"""
    lib = ORACLE_LIB if lib_name == "oracle" else emu_lib
    rng = np.random.default_rng(31)
    # a chain with three multi-allelic sites: backbone nodes B_i, between them 6 / 2 / 5 alternative nodes
    nodes, sites, at = [], [], 0
    for n_alt in (6, 2, 5):
        nodes.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, 20))); b = len(nodes) - 1
        alts = []
        for _ in range(n_alt):
            nodes.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 6))))); alts.append(len(nodes) - 1)
        sites.append((b, alts))
    nodes.append("ACGTACGTACGTACGTTTGA"); last = len(nodes) - 1
    threads = []
    for _ in range(300):
        t = []
        for b, alts in sites:
            t += [2 * b, 2 * alts[int(rng.integers(0, len(alts)))]]
        threads.append(t + [2 * last])
    lens = np.repeat(np.array([len(s) for s in nodes]), 2)
    allt = both_orientations(threads)
    counts = visit_counts(allt, len(lens))
    probs = whole_node_problems(rng, lens, counts, 200)
    expected = [trie_forest(lens, p[0], continuations_of_node(allt, p[0]), p[3], p[4]) for p in probs]
    eng = capi.Engine(lib=lib)
    check_forest(eng, eng.haplo_index(nodes, threads), probs, expected)
    assert max(len(e) for e in expected) >= 14                       # the whole bubble structure behind the first backbone node


def shim_tail_forests(engine_lib):
    """MinimizerMapper::get_tail_forest as the host shim offers it (GaplessExtender::get_tail_forest, reference-shaped: a vector of
    (parent, handle) trees with their root trim) on the toy graph of the reference's extender tests: the partial extension of
    "trim right flank" (src/unittest/gbwt_extender.cpp:1082) has a tail on either side; the trees must be the tries of the threads'
    continuations."""
    import ctypes, json
    import util
    from test_gapless import TOY_NODES, SHORT_PATH, ALT_PATH
    h = util.host()
    h.vgh_gapless_create.restype = ctypes.c_void_p
    h.vgh_gapless_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    h.vgh_gapless_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_gapless_tail_forest.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    al = util.HostAligner(engine_lib)
    g = h.vgh_graph_create()
    try:
        for i, s in enumerate(TOY_NODES):
            assert h.vgh_graph_add_node(g, i + 1, s.encode()) == 0
        for a, b in [(1, 2), (1, 4), (1, 6), (2, 3), (2, 4), (3, 5), (4, 5), (5, 6), (6, 7), (6, 8), (7, 9), (8, 9)]:
            assert h.vgh_graph_add_edge(g, a, b) == 0
        threads = [SHORT_PATH, ALT_PATH, SHORT_PATH]
        flat = [2 * n for t in threads for n in t]; off = np.concatenate([[0], np.cumsum([len(t) for t in threads])])
        x = h.vgh_gapless_create(al.ptr, g, (ctypes.c_int64 * len(flat))(*flat), (ctypes.c_int32 * len(off))(*[int(v) for v in off]), len(threads))
        assert x, h.vgh_last_error().decode()
        try:
            read = "xAGGGTxAx"; seeds = [((4, False, 2), 4)]
            sd = [v for (node_id, rev, offset), ro in seeds for v in (node_id, int(rev), offset, ro)]
            out = {}
            for left in (0, 1):
                buf = ctypes.create_string_buffer(1 << 16)
                rc = h.vgh_gapless_tail_forest(x, read.encode(), (ctypes.c_int64 * len(sd))(*sd), len(seeds), 1, 0, left, buf, len(buf))
                assert rc == 0, h.vgh_last_error().decode()
                out[left] = json.loads(buf.value.decode())
        finally:
            h.vgh_gapless_destroy(x)
    finally:
        h.vgh_graph_destroy(g)
    # expectation: the extension covers read [1, 6) on nodes 2, 4, 5 (the reference's REQUIREs); oriented node = 2 * (id - 1) + is_reverse
    lens = np.repeat(np.array([len(s) for s in TOY_NODES]), 2)
    allt = both_orientations([[2 * (n - 1) for n in t] for t in threads])
    path = [2 * (2 - 1), 2 * (4 - 1), 2 * (5 - 1)]
    L = len(read)
    for left, tail_len, p in ((0, L - 6, path), (1, 1, [q ^ 1 for q in reversed(path)])):
        gap = workloads.longest_detectable_gap(L, tail_len, 1, 6, 1, 5)
        assert out[left]["gap"] == gap and not out[left]["left_full"] and not out[left]["right_full"]
        conts = continuations_of_path(allt, p)
        got = out[left]["trees"]
        # rebuild the expectation with the cut the shim used: the root trim it reports (0 when the root was skipped)
        for cut in range(0, int(lens[p[-1]]) + 1):
            exp = trie_forest(lens, p[-1], conts, cut, tail_len + gap)
            trees, cur = [], None
            for par, node, ln in exp:
                if par < 0:
                    cur = []; trees.append(cur); base = len([1 for t in trees[:-1] for _ in t])
                cur.append([par - base if par >= 0 else -1, (node >> 1) + 1, node & 1])
            if [t["tree"] for t in got] == trees and all(t["root_trim"] == (cut if cut < lens[p[-1]] else 0) for t in got):
                break
        else:
            raise AssertionError("no cut on node %d reproduces the shim's forest %s" % (p[-1], got))
        assert len(got) >= 1


def test_host_shim_get_tail_forest_on_the_oracle():
    shim_tail_forests(ORACLE_LIB)


@pytest.mark.gpu
def test_host_shim_get_tail_forest_on_hip():
    shim_tail_forests(ENGINE_LIB)
