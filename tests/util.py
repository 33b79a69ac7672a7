"""Shared helpers for the test-suite: library handles and golden-case runner."""
import ctypes
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "libvgoracle.so")
HOST_LIB = os.path.join(ROOT, "vg_amd", "libvgamd_host.so")
ENGINE_LIB = os.path.join(ROOT, "vg_amd", "libvgamd.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvgamd_emu.so")      # CPU lock-step emulation of the kernels
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


_host = None


def host():
    global _host
    if _host is None:
        h = ctypes.CDLL(HOST_LIB)
        h.vgh_last_error.restype = ctypes.c_char_p
        h.vgh_graph_create.restype = ctypes.c_void_p
        h.vgh_graph_destroy.argtypes = [ctypes.c_void_p]
        h.vgh_graph_add_node.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
        h.vgh_graph_add_edge.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        h.vgh_aligner_create.restype = ctypes.c_void_p
        h.vgh_aligner_create.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 6
        h.vgh_aligner_destroy.argtypes = [ctypes.c_void_p]
        h.vgh_align.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        h.vgh_qual_adj_aligner_create.restype = ctypes.c_void_p
        h.vgh_qual_adj_aligner_create.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 6
        h.vgh_align_q.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        h.vgh_align_xdrop.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        _host = h
    return _host


class HostAligner:
    """Drives the C++ host shim (vg_amd/host) the way src/unittest/*.cpp drives vg's Aligner."""

    def __init__(self, engine_lib, scores=(1, 4, 6, 1, 5), device=0, qual_adj=False):
        self.h = host()
        make = self.h.vgh_qual_adj_aligner_create if qual_adj else self.h.vgh_aligner_create
        self.ptr = make(engine_lib.encode() if engine_lib else None, device, *scores)
        if not self.ptr:
            raise RuntimeError(self.h.vgh_last_error().decode())

    def __del__(self):
        if getattr(self, "ptr", None):
            self.h.vgh_aligner_destroy(self.ptr)

    def run(self, nodes, edges, read, call, pin_left=False, max_alt_alns=1, quality=None):
        g = self.h.vgh_graph_create()
        try:
            for nid, seq in nodes:
                assert self.h.vgh_graph_add_node(g, nid, seq.encode()) == 0
            for a, b in edges:
                assert self.h.vgh_graph_add_edge(g, a, b) == 0
            buf = ctypes.create_string_buffer(1 << 20 if max_alt_alns <= 100 else 1 << 26)
            code = {"align": 0, "align_score": 1, "align_pinned": 2, "align_pinned_multi": 3, "align_pinned_xdrop": 4,
                    "align_global_banded": 5}[call]
            if quality is not None:
                q = bytes(bytearray(int(x) for x in quality))
                rc = self.h.vgh_align_q(self.ptr, g, read.encode(), q, code, int(pin_left), max_alt_alns, buf, len(buf))
            else:
                rc = self.h.vgh_align(self.ptr, g, read.encode(), code, int(pin_left), max_alt_alns, buf, len(buf))
            if rc != 0:
                raise RuntimeError(self.h.vgh_last_error().decode())
            return json.loads(buf.value.decode())
        finally:
            self.h.vgh_graph_destroy(g)


def run_align_xdrop(aligner, nodes, edges, read, mems, reverse_complemented, max_gap=40):
    """Aligner::align_xdrop through the host shim; mems = [{begin, end, nodes: [[id, offset, is_rev]]}]."""
    h = aligner.h
    g = h.vgh_graph_create()
    try:
        for nid, seq in nodes:
            assert h.vgh_graph_add_node(g, nid, seq.encode()) == 0
        for a, b in edges:
            assert h.vgh_graph_add_edge(g, a, b) == 0
        flat = []
        for m in mems:
            hit = m["nodes"][-1]
            flat += [m["begin"], m["end"], hit[0], hit[1], int(hit[2])]
        arr = (ctypes.c_int64 * max(1, len(flat)))(*flat)
        buf = ctypes.create_string_buffer(1 << 20)
        rc = h.vgh_align_xdrop(aligner.ptr, g, read.encode(), arr, len(mems), int(reverse_complemented), max_gap, buf, len(buf))
        if rc != 0:
            raise RuntimeError(h.vgh_last_error().decode())
        return json.loads(buf.value.decode())
    finally:
        h.vgh_graph_destroy(g)


def check_expectations(case, aln, sibling_scores=None):
    """Assert every transcribed REQUIRE of a golden case against an alignment dict."""
    maps = aln["path"]["mapping"]
    for exp in case["expect"]:
        kind = exp[0]
        ctx = "%s [%s] %r" % (case["source"], case["name"], exp)
        if kind == "score":
            assert aln["score"] == exp[1], ctx
        elif kind == "score_minus":
            assert sibling_scores is not None and exp[1] in sibling_scores, ctx
            assert aln["score"] == sibling_scores[exp[1]] + exp[2], ctx
        elif kind == "mapping_size":
            assert len(maps) == exp[1], ctx
        elif kind == "node_id":
            assert maps[exp[1]]["position"]["node_id"] == exp[2], ctx
        elif kind == "offset":
            assert maps[exp[1]]["position"]["offset"] == exp[2], ctx
        elif kind == "is_reverse":
            assert maps[exp[1]]["position"]["is_reverse"] == exp[2], ctx
        elif kind == "rank":
            assert maps[exp[1]]["rank"] == exp[2], ctx
        elif kind == "edit_size":
            assert len(maps[exp[1]]["edit"]) == exp[2], ctx
        elif kind in ("from_length", "to_length", "sequence"):
            assert maps[exp[1]]["edit"][exp[2]][kind] == exp[3], ctx
        elif kind == "mapping_from_length":
            assert sum(e["from_length"] for e in maps[exp[1]]["edit"]) == exp[2], ctx
        elif kind == "mapping_to_length":
            assert sum(e["to_length"] for e in maps[exp[1]]["edit"]) == exp[2], ctx
        elif kind == "path_from_length":
            assert sum(e["from_length"] for m in maps for e in m["edit"]) == exp[1], ctx
        elif kind == "path_to_length":
            assert sum(e["to_length"] for m in maps for e in m["edit"]) == exp[1], ctx
        else:
            raise AssertionError("unknown expectation " + ctx)
