"""vgk_chain_stitch — one Path per read out of the pieces of its chain (WFAAlignment::to_path + append_path + simplify(path, false)):
the oracle (oracle/vgo_chain.c, a literal object-by-object restatement) against the reference's own simplify vectors, and the engine's
streaming kernel (vg_amd/csrc/chain_device.hpp) against the oracle on random piece lists — emulated here, on the GPU under -m gpu."""
import os
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT, load_golden
from vg_amd import capi
from test_wfa import random_wfa_case

M, X, I, D = capi.WFA_MATCH, capi.WFA_MISMATCH, capi.WFA_INSERTION, capi.WFA_DELETION


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu", "oracle"], cwd=ROOT)
    return EMU_LIB


def run(n):
    return (int(n[1]) << 2) | int(n[0])


def flat(res, maps, edits, r):
    """read r's result as [(node, offset, [(kind, len)...])...]"""
    out = []
    for k in range(int(res["mapping_begin"][r]), int(res["mapping_begin"][r]) + int(res["n_mappings"][r])):
        m = maps[k]
        out.append((int(m["node"]), int(m["offset"]), [(int(e) & 3, int(e) >> 2) for e in edits[int(m["edit_begin"]):int(m["edit_begin"]) + int(m["n_edits"])]]))
    return out


def path_pieces(paths):
    """paths: per read a list of mappings (oriented node | None, offset, [(kind, len)...]) -> one PATH piece per read + the arrays"""
    pieces = np.zeros(len(paths), dtype=capi.CHAIN_PIECE_DT); mappings = []; edits = []
    for r, p in enumerate(paths):
        pieces[r]["kind"] = capi.PIECE_PATH; pieces[r]["path_begin"] = len(mappings); pieces[r]["path_len"] = len(p)
        for node, off, ed in p:
            mappings.append((capi.WFA_NO_NODE if node is None else node, off, len(edits), len(ed)))
            edits += [run(e) for e in ed]
    return pieces, np.arange(len(paths) + 1, dtype=np.uint64), np.array(mappings, dtype=capi.CHAIN_MAPPING_DT), np.array(edits, dtype=np.uint32)


def golden_paths():
    out = []
    for c in load_golden("ref_simplify.json")["cases"]:
        p = []
        for m in c["mappings"]:
            ed = []
            for f, t, has_seq in m["edits"]:
                ed.append((M if f == t and not has_seq else X if f == t else I if f == 0 else D, max(f, t)))
            p.append((2 * (m["node_id"] - 1) + int(m["is_reverse"]) if m["node_id"] else None, m["offset"], ed))
        out.append((c, p))
    return out


@pytest.mark.parametrize("which", ["oracle", "emu"])
def test_reference_simplify_vectors(which, emu_lib):
    """src/unittest/path.cpp:21-45, src/unittest/alignment.cpp:57-102 (tests/golden/ref_simplify.json)"""
    eng = capi.Engine(lib=ORACLE_LIB if which == "oracle" else emu_lib)
    idx = eng.haplo_index(["A" * 20] * 70, [list(range(0, 140, 2))])
    cases = golden_paths()
    pieces, off, mappings, edits = path_pieces([p for _, p in cases])
    res, om, oe = eng.chain_stitch(idx, pieces, off, mappings=mappings, edits=edits)
    for r, (c, _) in enumerate(cases):
        got = flat(res, om, oe, r); e = c["expect"]
        assert res["status"][r] == 0 and len(got) == e["mapping_size"], (c["source"], got)
        if e["node_ids"]:
            assert [g[0] // 2 + 1 for g in got] == e["node_ids"], c["source"]
        for k, n in e["edit_sizes"].items():
            assert len(got[int(k)][2]) == n, (c["source"], got)
    # what the vectors do not state but the rules give: the insertion of node 67's mapping moves onto node 68's, its deletion stays
    assert flat(res, om, oe, 0) == [(134, 0, [(M, 1), (I, 4)]), (132, 0, [(D, 3)]), (130, 0, [(M, 17)])]
    assert flat(res, om, oe, 2)[0][0] == capi.WFA_NO_NODE and flat(res, om, oe, 2)[0][2] == [(I, 793 + 18 + 161)]


def hand_cases():
    """(pieces of one read as a path, expected) — each names the rule of src/path.cpp it exercises"""
    return [
        # :1382-1394 two mappings that continue each other on one node join, and their runs merge again
        ([(0, 3, [(M, 4)]), (0, 7, [(M, 2), (X, 1)]), (0, 10, [(X, 1), (M, 3)])], [(0, 3, [(M, 6), (X, 2), (M, 3)])]),
        # :1371-1380 a mapping without a position takes the previous one's node and from_length(*l) as its offset: it joins only when l starts at 0
        ([(0, 0, [(M, 5)]), (None, 0, [(I, 3)]), (2, 0, [(M, 4)])], [(0, 0, [(M, 5), (I, 3)]), (2, 0, [(M, 4)])]),
        ([(0, 2, [(M, 5), (I, 1)]), (None, 0, [(I, 3)]), (2, 0, [(M, 4)])], [(0, 2, [(M, 5), (I, 1), (I, 3)]), (2, 0, [(M, 4)])]),       # appended, NOT merged (:1349)
        # :1361-1369 a leading mapping without a position takes the next one's position
        ([(None, 0, [(I, 7)]), (4, 5, [(M, 4)])], [(4, 5, [(I, 7), (M, 4)])]),
        # :1422-1475 deletions before the first and after the last read base go; the first mapping's offset moves on
        ([(0, 1, [(D, 2), (M, 3)]), (2, 0, [(M, 2), (D, 4)]), (4, 0, [(D, 5)])], [(0, 3, [(M, 3)]), (2, 0, [(M, 2)])]),
        ([(0, 0, [(D, 20)]), (2, 0, [(D, 3), (M, 2), (D, 1)])], [(2, 3, [(M, 2), (D, 1)])]),          # one mapping with read bases: only its leading deletions go (:1453-1470)
        # nothing but deletions: nothing is left
        ([(0, 0, [(D, 20)])], []),
        # zero-length runs are not edits
        ([(0, 0, [(M, 0), (M, 3), (I, 0), (M, 2)])], [(0, 0, [(M, 5)])]),
    ]


@pytest.mark.parametrize("which", ["oracle", "emu"])
def test_simplify_rules_by_hand(which, emu_lib):
    eng = capi.Engine(lib=ORACLE_LIB if which == "oracle" else emu_lib)
    idx = eng.haplo_index(["A" * 20] * 8, [list(range(0, 16, 2))])
    cases = hand_cases()
    pieces, off, mappings, edits = path_pieces([c for c, _ in cases])
    res, om, oe = eng.chain_stitch(idx, pieces, off, mappings=mappings, edits=edits)
    for r, (c, want) in enumerate(cases):
        want = [(capi.WFA_NO_NODE if n is None else n, o, e) for n, o, e in want]
        assert res["status"][r] == 0 and flat(res, om, oe, r) == want, (r, c, flat(res, om, oe, r))
        assert res["to_length"][r] == sum(l for _, _, e in want for k, l in e if k != D) and res["from_length"][r] == sum(l for _, _, e in want for k, l in e if k != I)


def alignment_pieces_case(which_lib):
    """to_path (src/gbwt_extender.cpp:954-1070): ALIGNMENT pieces over nodes of 4, 3, 5 bases"""
    eng = capi.Engine(lib=which_lib)
    idx = eng.haplo_index(["ACGT", "ACG", "ACGTA"], [[0, 2, 4]])
    nodes = np.array([0, 2, 4, 4], dtype=np.uint32)
    edits = np.array([run(e) for e in [(M, 3), (I, 2), (M, 2), (D, 1), (X, 1), (M, 4),     # piece 0: from node 0 offset 1 across all three nodes
                                       (I, 6),                                                # piece 1: unlocalized insertion
                                       (M, 2),                                                # piece 2: on node 4 (oriented) offset 0 ... does not continue piece 0
                                       (M, 9)]], dtype=np.uint32)                             # piece 3: walks off its path
    pieces = np.zeros(5, dtype=capi.CHAIN_PIECE_DT)
    pieces[0] = (capi.PIECE_ALIGNMENT, 0, 1, 0, 3, 0, 6, 0)
    pieces[1] = (capi.PIECE_ALIGNMENT, 0, 0, 0, 0, 6, 1, 0)
    pieces[2] = (capi.PIECE_ALIGNMENT, 0, 0, 3, 1, 7, 1, 0)
    pieces[3] = (capi.PIECE_ALIGNMENT, 0, 0, 0, 2, 8, 1, 0)
    pieces[4] = (capi.PIECE_LINK, 0, 0, 0, 0, 0, 0, 0)                                        # no vgk_wfa_extend call before: "is not OK"
    off = np.array([0, 3, 4, 5], dtype=np.uint64)
    return eng, idx, pieces, off, nodes, edits


@pytest.mark.parametrize("which", ["oracle", "emu"])
def test_alignment_pieces_become_paths(which, emu_lib):
    eng, idx, pieces, off, nodes, edits = alignment_pieces_case(ORACLE_LIB if which == "oracle" else emu_lib)
    res, om, oe = eng.chain_stitch(idx, pieces, off, nodes=nodes, edits=edits)
    # node 0 (4 bases) from offset 1: M3 fills it; the insertion opens node 2's mapping and moves back (:1345-1352); M2 + D1 fill node 2 (3 bases);
    # X1 M4 fill node 4; the unlocalized insertion follows node 4's mapping, which started at 0: joined (:1371-1394); piece 2 restarts node 4 at 0: its own mapping
    assert flat(res, om, oe, 0) == [(0, 1, [(M, 3), (I, 2)]), (2, 0, [(M, 2), (D, 1)]), (4, 0, [(X, 1), (M, 4), (I, 6)]), (4, 0, [(M, 2)])]
    assert list(res["status"]) == [0, capi.VGK_EINVAL, capi.VGK_EINVAL] and list(res["n_mappings"][1:]) == [0, 0]
    # room for read 0 only / for nothing: VGK_EOPS on the reads that do not fit, sizes still reported
    r2, m2, e2 = eng.chain_stitch(idx, pieces[:3], off[:2], nodes=nodes, edits=edits, mapping_cap=3, edit_cap=100)
    assert r2["status"][0] == capi.VGK_EOPS and r2["n_mappings"][0] == 4


def random_pieces(rng, eng, idx, nodes, wres, n_reads):
    """piece lists mixing the last wfa_extend call's results (LINK), stated alignments (exact-match anchors along real node paths, unlocalized
    insertions) and stated paths — not coherent chains: the rules under test only ever look at two neighbouring mappings"""
    lens = [len(s) for s in nodes]
    pieces, off, pn, pm, pe = [], [0], [], [], []
    ok = [i for i in range(len(wres)) if wres["status"][i] == 0 and wres["ok"][i]]
    for _ in range(n_reads):
        for _ in range(int(rng.integers(0, 9))):
            x = rng.random()
            if x < 0.45 and ok:
                pieces.append((capi.PIECE_LINK, ok[int(rng.integers(0, len(ok)))], 0, 0, 0, 0, 0, 0))
            elif x < 0.6:
                pieces.append((capi.PIECE_ALIGNMENT, 0, 0, 0, 0, len(pe), 1, 0)); pe.append(run((I, int(rng.integers(1, 9)))))
            elif x < 0.8:                                   # an anchor: one match run from inside a node, over one to three nodes in id order
                v = int(rng.integers(0, len(nodes))); o = int(rng.integers(0, lens[v])); k = min(int(rng.integers(1, 4)), len(nodes) - v)
                total = sum(lens[v:v + k]) - o
                n = int(rng.integers(max(1, total - lens[v + k - 1] + 1), total + 1))
                pieces.append((capi.PIECE_ALIGNMENT, 0, o, len(pn), k, len(pe), int(rng.integers(1, 3)), 0)); pn += [2 * (v + j) for j in range(k)]
                if pieces[-1][6] == 2 and n > 1:
                    a = int(rng.integers(1, n)); pe += [run((M, a)), run((M, n - a))]          # unmerged runs of one kind
                else:
                    pieces[-1] = pieces[-1][:6] + (1, 0); pe.append(run((M, n)))
            else:                                           # a stated path of one to three mappings with arbitrary small runs (deletions and insertions at either end)
                k = int(rng.integers(1, 4)); pieces.append((capi.PIECE_PATH, 0, 0, len(pm), k, 0, 0, 0))
                for _ in range(k):
                    ne = int(rng.integers(0, 5)); v = int(rng.integers(0, len(nodes)))
                    pm.append((capi.WFA_NO_NODE if rng.random() < 0.1 else 2 * v + int(rng.integers(0, 2)), int(rng.integers(0, 3)), len(pe), ne))
                    pe += [run((int(rng.integers(0, 4)), int(rng.integers(0, 4)))) for _ in range(ne)]
        off.append(len(pieces))
    return (np.array(pieces, dtype=capi.CHAIN_PIECE_DT) if pieces else np.zeros(0, dtype=capi.CHAIN_PIECE_DT), np.array(off, dtype=np.uint64),
            np.array(pn, dtype=np.uint32), np.array(pm, dtype=capi.CHAIN_MAPPING_DT) if pm else np.zeros(0, dtype=capi.CHAIN_MAPPING_DT), np.array(pe, dtype=np.uint32))


def engine_equals_oracle(lib, seeds, n_reads, forms=(capi.VGK_WFA_FORM_HYBRID if hasattr(capi, "VGK_WFA_FORM_HYBRID") else 0,)):
    checked = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        nodes, threads, wp = random_wfa_case(rng, 80)
        eng = capi.Engine(lib=lib); ora = capi.Engine(lib=ORACLE_LIB)
        ei, oi = eng.haplo_index(nodes, threads), ora.haplo_index(nodes, threads)
        for form in forms:
            eng.lib.vgk_wfa_set_form(eng.h, form)
            wres = eng.wfa_extend(ei, wp)[0]; ores = ora.wfa_extend(oi, wp)[0]
            both = ores.copy(); both["ok"] &= (wres["status"] == 0) & (wres["ok"] != 0)       # (the engine's tables decline a problem now and then: test_wfa.py)
            assert both["ok"].sum() > 0.8 * (ores["ok"] != 0).sum()
            pieces, off, pn, pm, pe = random_pieces(np.random.default_rng(seed + 1000), eng, ei, nodes, both, n_reads)
            a = eng.chain_stitch(ei, pieces, off, nodes=pn, mappings=pm, edits=pe)
            b = ora.chain_stitch(oi, pieces, off, nodes=pn, mappings=pm, edits=pe)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes(), (seed, form)
            checked += int((b[0]["status"] == 0).sum())
    return checked


def test_emulated_kernel_equals_the_oracle(emu_lib):
    assert engine_equals_oracle(emu_lib, range(40), 60, forms=(0, 1, 2)) > 4000


@pytest.mark.gpu
def test_kernel_equals_the_oracle_on_the_gpu():
    assert engine_equals_oracle(ENGINE_LIB, range(100, 160), 400, forms=(0, 2)) > 30000


@pytest.mark.gpu
def test_reference_vectors_and_rules_on_the_gpu():
    test_reference_simplify_vectors("gpu", ENGINE_LIB)
    test_simplify_rules_by_hand("gpu", ENGINE_LIB)
    test_alignment_pieces_become_paths("gpu", ENGINE_LIB)
