"""AlignmentBatch (the deferred-submission shim): a batch of mixed Aligner calls flushed in one engine launch per
kernel family must fill every Alignment exactly as the direct calls do."""
import ctypes
import json

import pytest

import util

CALLS = {"align": 0, "align_score": 1, "align_pinned": 2, "align_global_banded": 5}


def batch_vs_direct(engine_lib):
    h = util.host()
    h.vgh_batch_create.restype = ctypes.c_void_p
    h.vgh_batch_create.argtypes = [ctypes.c_void_p]
    h.vgh_batch_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_batch_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    # the reference's own unit-test problems, default scoring, no qualities: local, pinned (both ends) and banded calls mixed
    jobs = []
    for fname, calls in (("ref_aligner.json", ("align",)), ("ref_pinned_alignment.json", ("align_pinned",)), ("ref_banded_global_aligner.json", ("align_global_banded",))):
        for c in util.load_golden(fname):
            if c["call"] in calls and not c["qual_adj"] and c["scores"] == [1, 4, 6, 1, 5] and c["nodes"]:
                if c["call"] == "align":
                    jobs.append((c, "align", False, 1))
                elif c["call"] == "align_pinned":
                    args = c["args"]
                    if len(args) > 2 and args[2] is True:
                        continue                                           # xdrop variant: not batched
                    jobs.append((c, "align_pinned", bool(args[1]), 1))
                else:
                    args = c["args"]
                    jobs.append((c, "align_global_banded", args[2] if len(args) > 2 else True, args[1]))
    assert len(jobs) > 50
    al = util.HostAligner(engine_lib)
    direct = []
    for c, call, flag, arg in jobs:
        try:
            direct.append(al.run(c["nodes"], c["edges"], c["read"], call, pin_left=flag, max_alt_alns=arg))
        except RuntimeError as e:
            direct.append(str(e))
    keep = [k for k, d in enumerate(direct) if not isinstance(d, str)]       # banded cases that throw (no alignment in band) stay direct-only
    b = h.vgh_batch_create(al.ptr)
    graphs = []
    try:
        for k in keep:
            c, call, flag, arg = jobs[k]
            g = h.vgh_graph_create(); graphs.append(g)
            for nid, seq in c["nodes"]:
                assert h.vgh_graph_add_node(g, nid, seq.encode()) == 0
            for x, y in c["edges"]:
                assert h.vgh_graph_add_edge(g, x, y) == 0
            assert h.vgh_batch_add(b, g, c["read"].encode(), None, CALLS[call], int(flag), arg) == 0, h.vgh_last_error().decode()
        buf = ctypes.create_string_buffer(1 << 24)
        assert h.vgh_batch_flush(b, buf, len(buf)) == 0, h.vgh_last_error().decode()
        out = json.loads(buf.value.decode())
    finally:
        h.vgh_batch_destroy(b)
        for g in graphs:
            h.vgh_graph_destroy(g)
    assert len(out) == len(keep)
    checked = 0
    for k, got in zip(keep, out):
        assert got == direct[k], (jobs[k][0]["source"], jobs[k][1])
        # ... and what came back through the batch satisfies the reference's own REQUIREs for the case (not merely "equals the
        # direct call"): expectations that refer to a sibling alignment's score are resolved against the direct runs of the group
        c = jobs[k][0]
        siblings = {jobs[j][0]["aln"]: direct[j]["score"] for j in keep if jobs[j][0]["source"] == c["source"]}
        util.check_expectations(c, got, siblings)
        checked += len(c["expect"])
    assert checked > 300
    return len(keep)


def test_alignment_batch_equals_direct_calls_on_the_oracle():
    assert batch_vs_direct(util.ORACLE_LIB) > 45


@pytest.mark.gpu
def test_alignment_batch_equals_direct_calls_on_hip():
    assert batch_vs_direct(util.ENGINE_LIB) > 45
