"""The class shapes around the aligners (vg_amd/host/aligner_client.hpp): AlignerClient (reference src/aligner.hpp:266-316,
src/aligner.cpp:1350-1440 — scores from a matrix stream, the regular / quality-adjusted aligner by get_aligner, parse_matrix's errors)
and XdropAligner / QualAdjXdropAligner (src/dozeu_interface.hpp:259-330: the bonus an argument of the call).  Each must answer exactly
like the Aligner / QualAdjAligner calls it forwards to (which the reference's own test vectors hold: test_golden_gssw_oracle.py)."""
import ctypes
import json

import numpy as np
import pytest

import util
from util import ENGINE_LIB, ORACLE_LIB, HostAligner

MATRIX = "1 -4 -4 -4\n-4 1 -4 -4\n-4 -4 1 -4\n-4 -4 -4 1\n"
NODES = [(1, "ACGTACGTAGCTAGCTAGGA"), (2, "T"), (3, "G"), (4, "CCATCGATCGATTACGGA")]
EDGES = [(1, 2), (1, 3), (2, 4), (3, 4)]
READ = "ACGTACGTAGCTAGCTAGGAGCCATCGATCGTTACGG"


def client_call(lib, what, read, qual=None, adjust=False, pin_left=True, bonus=5, max_gap=40, matrix=MATRIX, go=6, ge=1):
    h = util.host()
    h.vgh_client_align_pinned.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 2 + [ctypes.c_char_p, ctypes.c_size_t]
    g = h.vgh_graph_create()
    try:
        for nid, seq in NODES:
            assert h.vgh_graph_add_node(g, nid, seq.encode()) == 0
        for a, b in EDGES:
            assert h.vgh_graph_add_edge(g, a, b) == 0
        buf = ctypes.create_string_buffer(1 << 18)
        q = bytes(bytearray(qual)) if qual is not None else None
        rc = h.vgh_client_align_pinned(lib.encode(), matrix.encode(), go, ge, bonus, int(adjust), what, g, read.encode(), q, int(pin_left), max_gap, buf, len(buf))
        if rc != 0:
            raise RuntimeError(h.vgh_last_error().decode())
        return json.loads(buf.value.decode())
    finally:
        h.vgh_graph_destroy(g)


def shapes(lib):
    qual = [30] * len(READ)
    plain = HostAligner(lib); qa = HostAligner(lib, qual_adj=True)
    # AlignerClient: the regular aligner unless qualities are present AND adjustment is on (src/aligner.cpp:1360-1364)
    want = plain.run(NODES, EDGES, READ, "align_pinned", pin_left=True)
    assert client_call(lib, 0, READ) == want
    assert client_call(lib, 0, READ, qual=qual, adjust=False)["score"] == want["score"]
    want_q = qa.run(NODES, EDGES, READ, "align_pinned", pin_left=True, quality=qual)
    got_q = client_call(lib, 0, READ, qual=qual, adjust=True)
    assert got_q["score"] == want_q["score"] and got_q["path"] == want_q["path"]
    # XdropAligner / QualAdjXdropAligner: Aligner::align_pinned(..., xdrop = true, max_gap) with the bonus the aligner was built with
    for pin_left in (True, False):
        want_x = plain.run(NODES, EDGES, READ, "align_pinned_xdrop", pin_left=pin_left, max_alt_alns=40)
        assert client_call(lib, 1, READ, pin_left=pin_left, bonus=5, max_gap=40) == want_x
    want_xq = qa.run(NODES, EDGES, READ, "align_pinned_xdrop", pin_left=True, max_alt_alns=40, quality=qual)
    got_xq = client_call(lib, 2, READ, qual=qual, pin_left=True, bonus=5, max_gap=40)
    assert got_xq["score"] == want_xq["score"] and got_xq["path"] == want_xq["path"]
    # a different bonus per call: its own engine context, the score moves by the bonus
    b0 = client_call(lib, 1, READ, bonus=0)["score"]; b9 = client_call(lib, 1, READ, bonus=9)["score"]
    assert b0 <= client_call(lib, 1, READ, bonus=5)["score"] <= b9 and b9 - b0 <= 9


def test_shapes_on_the_oracle():
    shapes(ORACLE_LIB)


def test_parse_matrix_errors():
    with pytest.raises(RuntimeError, match="4x4 whitespace separated"):
        client_call(ORACLE_LIB, 1, READ, matrix="1 -4 -4")
    with pytest.raises(RuntimeError, match=r"range \[-127,127\]"):
        client_call(ORACLE_LIB, 1, READ, matrix=MATRIX.replace("1 -4", "200 -4", 1))


@pytest.mark.gpu
def test_shapes_on_hip():
    shapes(ENGINE_LIB)
