"""DozeuPinningOverlay (the graph Aligner::align_pinned(xdrop = true) pins on, src/dozeu_pinning_overlay.cpp) in the host shim, held to the
reference's own unit test: src/unittest/dozeu_pinning_overlay.cpp:16-242, "produces expected topology for small test graph", REQUIRE by
REQUIRE.  The overlay is dumped as JSON by vgh_pinning_overlay (vg_amd/host/host_capi.cpp)."""
import ctypes
import json

import util


def overlay_of(nodes, edges, preserve_sinks):
    h = util.host()
    h.vgh_pinning_overlay.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    g = h.vgh_graph_create()
    try:
        for nid, seq in nodes:
            assert h.vgh_graph_add_node(g, nid, seq.encode()) == 0
        for a, b in edges:
            assert h.vgh_graph_add_edge(g, a, b) == 0
        buf = ctypes.create_string_buffer(1 << 16)
        assert h.vgh_pinning_overlay(g, int(preserve_sinks), buf, len(buf)) == 0, h.vgh_last_error().decode()
        return json.loads(buf.value.decode())
    finally:
        h.vgh_graph_destroy(g)


def test_overlay_topology_of_the_reference_test_graph():
    seqs = {1: "", 2: "CGGTG", 3: "AGAA", 4: "TTG"}                                           # :20-23
    ov = overlay_of(sorted(seqs.items()), [[1, 2], [1, 3], [2, 3], [3, 4]], False)            # :25-30
    assert ov["performed_duplications"]                                                        # :32
    assert ov["node_count"] == 4 and len(ov["handles"]) == 4                                   # :34, :66
    o1 = [x for x in ov["handles"] if x["sequence"] == seqs[3] and x["id"] == 3]               # :39-52
    o2 = [x for x in ov["handles"] if x["sequence"] == seqs[3] and x["id"] not in (1, 2, 3, 4)]
    o3 = [x for x in ov["handles"] if x["sequence"] == seqs[2]]
    o4 = [x for x in ov["handles"] if x["sequence"] == seqs[4]]
    assert len(o1) == len(o2) == len(o3) == len(o4) == 1                                       # :67-70 (and :60-62: nothing else)
    o1, o2, o3, o4 = o1[0], o2[0], o3[0], o4[0]
    assert o3["id"] == 2 and o4["id"] == 4                                                     # :53, :58
    for o, under in ((o1, 3), (o2, 3), (o3, 2), (o4, 4)):                                      # :74-81
        assert o["underlying"] == [under, False] and o["flip_underlying"] == [under, True]
        assert o["flip_flip_is_self"]                                                          # :108-110
    ids = {o["id"] for o in (o1, o2, o3, o4)}
    assert len(ids) == 4                                                                       # :93
    assert all(ov["min_id"] <= i <= ov["max_id"] for i in ids)                                 # :95-98
    assert set(ov["has_node"]) & set(range(1, 50)) == ids                                      # :83-85, :100-107
    fwd = lambda o: [o["id"], False]
    rev = lambda o: [o["id"], True]
    # from o1 (:114-140)
    assert o1["right"] == [fwd(o4)] and o1["left"] == [fwd(o3)] and o1["flip_right"] == [rev(o3)] and o1["flip_left"] == [rev(o4)]
    # from o2 (:144-169): the copy that stands for "entered from the empty source" has nothing to its left
    assert o2["right"] == [fwd(o4)] and o2["left"] == [] and o2["flip_right"] == [] and o2["flip_left"] == [rev(o4)]
    # from o3 (:173-198)
    assert o3["right"] == [fwd(o1)] and o3["left"] == [] and o3["flip_right"] == [] and o3["flip_left"] == [rev(o1)]
    # from o4 (:202-240)
    assert o4["right"] == [] and sorted(o4["left"]) == sorted([fwd(o1), fwd(o2)])
    assert sorted(o4["flip_right"]) == sorted([rev(o1), rev(o2)]) and o4["flip_left"] == []


def test_overlay_of_a_graph_without_empty_nodes_changes_nothing():
    ov = overlay_of([(1, "AC"), (2, "G"), (3, "TT")], [[1, 2], [1, 3], [2, 3]], False)
    assert not ov["performed_duplications"] and ov["node_count"] == 3 and sorted(ov["has_node"]) == [1, 2, 3]
    by = {x["id"]: x for x in ov["handles"]}
    assert sorted(by[1]["right"]) == [[2, False], [3, False]] and sorted(by[3]["left"]) == [[1, False], [2, False]]
