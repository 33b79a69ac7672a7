"""What MinimizerMapper::attempt_rescue does to an alignment back from align_xdrop (vg_amd/host/rescue_fixups.cpp), held to the reference's
unit test "MinimizerMapper can fix up alignments with deletions on the ends" (src/unittest/minimizer_mapper.cpp:1130-1179)."""
import ctypes
import json

import util


def fix_end_deletions(sequence, mappings):
    """mappings: [((node id, offset, is_reverse), [(from_length, to_length, has_sequence), ...]), ...] -> the alignment afterwards (dict)"""
    h = util.host()
    h.vgh_fix_dozeu_end_deletions.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                              ctypes.c_char_p, ctypes.c_size_t]
    pos = [x for (p, _) in mappings for x in (p[0], p[1], int(p[2]))]
    eds = [x for m, (_, edits) in enumerate(mappings) for e in edits for x in (m, e[0], e[1], int(e[2]))]
    buf = ctypes.create_string_buffer(1 << 16)
    rc = h.vgh_fix_dozeu_end_deletions(sequence.encode(), (ctypes.c_int64 * max(1, len(pos)))(*pos), len(mappings),
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


def test_nothing_but_deletions_clears_the_path_and_clean_alignments_are_left_alone():
    assert fix_end_deletions("", [((1, 0, False), [(3, 0, False)]), ((2, 0, False), [(1, 0, False)])])["path"]["mapping"] == []      # (:3534-3537)
    clean = [((5, 1, False), [(2, 2, False), (0, 1, True)]), ((6, 0, True), [(3, 3, False)])]
    maps = fix_end_deletions("ACGTAC", clean)["path"]["mapping"]
    assert [(m["position"]["node_id"], m["position"]["offset"], len(m["edit"])) for m in maps] == [(5, 1, 2), (6, 0, 1)]
    # a deletion inside stays; one at the right end of the last mapping goes (:3552-3564)
    maps = fix_end_deletions("ACGT", [((7, 0, False), [(2, 2, False), (3, 0, False), (2, 2, False), (1, 0, False)])])["path"]["mapping"]
    assert [(e["from_length"], e["to_length"]) for e in maps[0]["edit"]] == [(2, 2), (3, 0), (2, 2)]
