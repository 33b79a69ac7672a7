"""What MinimizerMapper::attempt_rescue does to an alignment back from align_xdrop (vg_amd/host/rescue_fixups.cpp), held to the reference's
unit test "MinimizerMapper can fix up alignments with deletions on the ends" (src/unittest/minimizer_mapper.cpp:1130-1179)."""
import ctypes
import json

import util


def fix_end_deletions(sequence, mappings, as_written=False):
    """mappings: [((node id, offset, is_reverse), [(from_length, to_length, has_sequence), ...]), ...] -> the alignment afterwards (dict)"""
    h = util.host()
    fn = h.vgh_fix_dozeu_end_deletions_as_written if as_written else h.vgh_fix_dozeu_end_deletions
    fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                              ctypes.c_char_p, ctypes.c_size_t]
    pos = [x for (p, _) in mappings for x in (p[0], p[1], int(p[2]))]
    eds = [x for m, (_, edits) in enumerate(mappings) for e in edits for x in (m, e[0], e[1], int(e[2]))]
    buf = ctypes.create_string_buffer(1 << 16)
    rc = fn(sequence.encode(), (ctypes.c_int64 * max(1, len(pos)))(*pos), len(mappings),
                                       (ctypes.c_int64 * max(1, len(eds)))(*eds), len(eds) // 4, buf, len(buf))
    assert rc == 0, h.vgh_last_error().decode()
    return json.loads(buf.value.decode())


def test_reference_case_deletions_on_both_ends():
    aln = fix_end_deletions("A", [((1, 3, False), [(2, 0, False)]),                                  # :1140-1147
                                  ((2, 0, False), [(2, 0, False), (1, 1, False), (1, 0, False)]),     # :1149-1161
                                  ((3, 0, False), [(1, 0, False)])])                                   # :1163-1169
    maps = aln["path"]["mapping"]
    assert len(maps) == 1                                                                              # :1173
    p = maps[0]["position"]
    assert p["node_id"] == 2 and p["offset"] == 2 and not p.get("is_reverse", False)                   # :1174-1176
    assert len(maps[0]["edit"]) == 1                                                                   # :1177
    e = maps[0]["edit"][0]
    assert e["from_length"] == 1 and e["to_length"] == 1 and e.get("sequence", "") == ""              # :1178-1180


def test_leading_deletion_where_mapping_and_edit_index_differ():
    """The reference drops the leading edits from `mappings[j]`, j the EDIT index (src/minimizer_mapper.cpp:3541); its own test has i = j = 1.
    Here i = 0, j = 1 (the first mapping opens with a deletion, a second mapping follows): the default takes mapping i — the deletion goes,
    the offset moves past it, every read base is still consumed; the as-written form is kept for comparison and leaves the deletion in."""
    aln_in = [((4, 2, False), [(3, 0, False), (2, 2, False)]), ((5, 0, False), [(1, 0, False), (2, 2, False)])]
    maps = fix_end_deletions("ACGT", aln_in)["path"]["mapping"]
    assert [(m["position"]["node_id"], m["position"]["offset"]) for m in maps] == [(4, 5), (5, 0)]
    assert [[(e["from_length"], e["to_length"]) for e in m["edit"]] for m in maps] == [[(2, 2)], [(1, 0), (2, 2)]]
    assert sum(e["to_length"] for m in maps for e in m["edit"]) == 4
    written = fix_end_deletions("ACGT", aln_in, as_written=True)["path"]["mapping"]
    assert [(e["from_length"], e["to_length"]) for e in written[0]["edit"]] == [(3, 0), (2, 2)]          # the leading deletion is still there
    assert written[0]["position"]["offset"] == 2 + 1                                                      # and the offset moved by the WRONG mapping's edit
    # i = j: both forms are the reference's tested behaviour
    same = [((1, 3, False), [(2, 0, False)]), ((2, 0, False), [(2, 0, False), (1, 1, False), (1, 0, False)])]
    assert fix_end_deletions("A", same) == fix_end_deletions("A", same, as_written=True)


def test_nothing_but_deletions_clears_the_path_and_clean_alignments_are_left_alone():
    assert fix_end_deletions("", [((1, 0, False), [(3, 0, False)]), ((2, 0, False), [(1, 0, False)])])["path"]["mapping"] == []      # (:3534-3537)
    clean = [((5, 1, False), [(2, 2, False), (0, 1, True)]), ((6, 0, True), [(3, 3, False)])]
    maps = fix_end_deletions("ACGTAC", clean)["path"]["mapping"]
    assert [(m["position"]["node_id"], m["position"]["offset"], len(m["edit"])) for m in maps] == [(5, 1, 2), (6, 0, 1)]
    # a deletion inside stays; one at the right end of the last mapping goes (:3552-3564)
    maps = fix_end_deletions("ACGT", [((7, 0, False), [(2, 2, False), (3, 0, False), (2, 2, False), (1, 0, False)])])["path"]["mapping"]
    assert [(e["from_length"], e["to_length"]) for e in maps[0]["edit"]] == [(2, 2), (3, 0), (2, 2)]


# ---- Aligner::align_xdrop_many: the rescue alignments of many reads, their passes side by side (vg_amd/host/aligner.cpp) -------------

def random_rescue_problems(seed, n):
    """a small DAG per read, a read walked through it with a few errors, and (for most) a seed: an exact 10-mer of the read on its node"""
    import numpy as np
    from gen import random_dag, random_walk_read
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        nodes, preds = random_dag(rng, int(rng.integers(2, 8)), 14)
        if any(len(s) == 0 for s in nodes):
            continue
        read = random_walk_read(rng, nodes, preds, int(rng.integers(20, 70)))
        mems = []
        if rng.random() < 0.7:                                           # a seed where a node holds 8 bases of the read verbatim
            for v, s in enumerate(nodes):
                hit = [(o, read.find(s[o:o + 8])) for o in range(0, max(1, len(s) - 7))] if len(s) >= 8 else []
                hit = [(o, q) for o, q in hit if q >= 0]
                if hit:
                    o, q = hit[0]
                    mems = [{"begin": q, "end": q + 8, "nodes": [[v + 1, o, False]]}]
                    break
        out.append({"nodes": [[v + 1, s] for v, s in enumerate(nodes)], "edges": [[p + 1, v + 1] for v, pr in enumerate(preds) for p in pr],
                    "read": read, "mems": mems})
    return out


def xdrop_many(aligner, problems, reverse_complemented, max_gap, fixups):
    h = aligner.h
    h.vgh_align_xdrop_many.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    h.vgh_graph_create.restype = ctypes.c_void_p
    graphs = []
    try:
        for p in problems:
            g = h.vgh_graph_create(); graphs.append(g)
            for nid, seq in p["nodes"]:
                assert h.vgh_graph_add_node(ctypes.c_void_p(g), nid, seq.encode()) == 0
            for a, b in p["edges"]:
                assert h.vgh_graph_add_edge(ctypes.c_void_p(g), a, b) == 0
        flat = [x for p in problems for m in p["mems"] for x in (m["begin"], m["end"], m["nodes"][-1][0], m["nodes"][-1][1], int(m["nodes"][-1][2]))]
        counts = [len(p["mems"]) for p in problems]
        n = len(problems)
        buf = ctypes.create_string_buffer(1 << 22)
        rc = h.vgh_align_xdrop_many(aligner.ptr, (ctypes.c_void_p * n)(*graphs), (ctypes.c_char_p * n)(*[p["read"].encode() for p in problems]),
                                    (ctypes.c_int64 * max(1, len(flat)))(*flat), (ctypes.c_int * n)(*counts), n, int(reverse_complemented), max_gap, int(fixups),
                                    buf, len(buf))
        assert rc == 0, h.vgh_last_error().decode()
        return json.loads(buf.value.decode())
    finally:
        for g in graphs:
            h.vgh_graph_destroy(ctypes.c_void_p(g))


def many_equals_direct(engine_lib, n):
    al = util.HostAligner(engine_lib)
    problems = random_rescue_problems(5, n)
    for rc in (False, True):
        direct = [util.run_align_xdrop(al, p["nodes"], p["edges"], p["read"], p["mems"], rc, 30) for p in problems]
        many = xdrop_many(al, problems, rc, 30, False)
        assert many == direct
        assert sum(1 for a in many if a["path"]["mapping"]) > n // 2
    fixed = xdrop_many(al, problems, False, 30, True)                      # with attempt_rescue's fix-ups behind it
    for a in fixed:
        maps = a["path"]["mapping"]
        if maps:                                                          # no deletion is left at either end, the score is the scorer's own
            assert maps[0]["edit"][0]["to_length"] > 0 and maps[-1]["edit"][-1]["to_length"] > 0
            assert a["score"] > 0


def test_align_xdrop_many_equals_the_direct_calls_on_the_oracle():
    many_equals_direct(util.ORACLE_LIB, 60)


def test_align_xdrop_many_equals_the_direct_calls_on_the_emulated_kernels():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    many_equals_direct(util.EMU_LIB, 12)


import pytest


@pytest.mark.gpu
def test_align_xdrop_many_equals_the_direct_calls_on_hip():
    many_equals_direct(util.ENGINE_LIB, 200)


def test_alignment_batch_defers_the_seeded_xdrop_beside_the_other_calls():
    """AlignmentBatch::align_xdrop: requests wait for flush(), which answers them through align_xdrop_many — in one batch with plain local
    alignments, every slot as the direct call leaves it"""
    h = util.host()
    h.vgh_batch_create.restype = ctypes.c_void_p; h.vgh_batch_create.argtypes = [ctypes.c_void_p]
    h.vgh_batch_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_batch_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int]
    h.vgh_batch_add_slot.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_add_xdrop_slot.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    h.vgh_graph_create.restype = ctypes.c_void_p
    al = util.HostAligner(util.ORACLE_LIB)
    problems = random_rescue_problems(9, 30)
    graphs = []
    b = h.vgh_batch_create(al.ptr)
    try:
        assert h.vgh_batch_reserve(b, 2 * len(problems)) == 0
        for k, p in enumerate(problems):
            g = h.vgh_graph_create(); graphs.append(g)
            for nid, seq in p["nodes"]:
                assert h.vgh_graph_add_node(ctypes.c_void_p(g), nid, seq.encode()) == 0
            for x, y in p["edges"]:
                assert h.vgh_graph_add_edge(ctypes.c_void_p(g), x, y) == 0
            flat = [x for m in p["mems"] for x in (m["begin"], m["end"], m["nodes"][-1][0], m["nodes"][-1][1], int(m["nodes"][-1][2]))]
            assert h.vgh_batch_add_xdrop_slot(b, 2 * k, ctypes.c_void_p(g), p["read"].encode(), (ctypes.c_int64 * max(1, len(flat)))(*flat), len(p["mems"]), 0, 30) == 0
            assert h.vgh_batch_add_slot(b, 2 * k + 1, ctypes.c_void_p(g), p["read"].encode(), 0, 0, 1) == 0, h.vgh_last_error().decode()      # call 0: align
        buf = ctypes.create_string_buffer(1 << 22)
        assert h.vgh_batch_flush(b, buf, len(buf)) == 0, h.vgh_last_error().decode()
        out = json.loads(buf.value.decode())
    finally:
        h.vgh_batch_destroy(b)
        for g in graphs:
            h.vgh_graph_destroy(ctypes.c_void_p(g))
    assert len(out) == 2 * len(problems)
    for k, p in enumerate(problems):
        assert out[2 * k] == util.run_align_xdrop(al, p["nodes"], p["edges"], p["read"], p["mems"], False, 30)
        assert out[2 * k + 1] == al.run(p["nodes"], p["edges"], p["read"], "align")
