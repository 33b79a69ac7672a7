"""AlignmentBatch (the deferred-submission shim): a batch of mixed Aligner calls flushed in one engine launch per
kernel family must fill every Alignment exactly as the direct calls do."""
import ctypes
import json

import pytest

import util

CALLS = {"align": 0, "align_score": 1, "align_pinned": 2, "align_pinned_xdrop": 4, "align_global_banded": 5}


def batch_vs_direct(engine_lib, devices=1, max_pending=0):
    """devices > 1: one aligner (engine context) per "device", flushes go to them in turn; max_pending > 0: submissions flush."""
    h = util.host()
    h.vgh_batch_create.restype = ctypes.c_void_p
    h.vgh_batch_create.argtypes = [ctypes.c_void_p]
    h.vgh_batch_create_multi.restype = ctypes.c_void_p
    h.vgh_batch_create_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flushes.argtypes = [ctypes.c_void_p]
    h.vgh_batch_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_batch_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    # the reference's own unit-test problems, default scoring, no qualities: local, pinned (both ends) and banded calls mixed
    jobs = []
    for fname, calls in (("ref_aligner.json", ("align",)), ("ref_pinned_alignment.json", ("align_pinned",)), ("ref_xdrop_aligner.json", ("align_pinned",)),
                         ("ref_banded_global_aligner.json", ("align_global_banded",))):
        for c in util.load_golden(fname):
            if c["call"] in calls and not c["qual_adj"] and c["scores"] == [1, 4, 6, 1, 5] and c["nodes"]:
                if c["call"] == "align":
                    jobs.append((c, "align", False, 1))
                elif c["call"] == "align_pinned":
                    args = c["args"]
                    if len(args) > 2 and args[2] is True:                  # align_pinned(..., xdrop = true, max_gap)
                        jobs.append((c, "align_pinned_xdrop", bool(args[1]), args[3] if len(args) > 3 and isinstance(args[3], int) else 40))
                    elif fname == "ref_pinned_alignment.json":
                        jobs.append((c, "align_pinned", bool(args[1]), 1))
                else:
                    args = c["args"]
                    jobs.append((c, "align_global_banded", args[2] if len(args) > 2 else True, args[1]))
    assert len(jobs) > 60 and sum(1 for j in jobs if j[1] == "align_pinned_xdrop") >= 4
    al = util.HostAligner(engine_lib)
    others = [util.HostAligner(engine_lib) for _ in range(devices - 1)]
    direct = []
    for c, call, flag, arg in jobs:
        try:
            direct.append(al.run(c["nodes"], c["edges"], c["read"], call, pin_left=flag, max_alt_alns=arg))
        except RuntimeError as e:
            direct.append(str(e))
    keep = [k for k, d in enumerate(direct) if not isinstance(d, str)]       # banded cases that throw (no alignment in band) stay direct-only
    if devices > 1 or max_pending:
        arr = (ctypes.c_void_p * devices)(al.ptr, *[o.ptr for o in others])
        b = h.vgh_batch_create_multi(arr, devices, max_pending)
    else:
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
        if max_pending:
            assert h.vgh_batch_flushes(b) >= len(keep) // max_pending
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
    assert checked > 350
    return len(keep)


def test_alignment_batch_equals_direct_calls_on_the_oracle():
    assert batch_vs_direct(util.ORACLE_LIB) > 55


def test_alignment_batch_over_two_contexts_with_size_triggered_flushes():
    """The single-process multi-device path: two engine contexts (two emulated "devices"), a flush every 16 submissions, flushes go
    to the contexts in turn — results and the reference's REQUIREs as for direct calls."""
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    assert batch_vs_direct(util.EMU_LIB, devices=2, max_pending=16) > 55


def test_alignment_batch_takes_submissions_from_many_threads():
    """giraffe's calling pattern: OpenMP threads submit their own reads; a flush by any of them answers everything submitted so far."""
    import threading
    h = util.host()
    h.vgh_batch_create_multi.restype = ctypes.c_void_p
    h.vgh_batch_create_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flushes.argtypes = [ctypes.c_void_p]
    h.vgh_batch_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_batch_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    cases = [c for c in util.load_golden("ref_pinned_alignment.json") if c["call"] == "align_pinned" and not c["qual_adj"] and c["scores"] == [1, 4, 6, 1, 5]
             and c["nodes"] and not (len(c["args"]) > 2 and c["args"][2] is True)]
    als = [util.HostAligner(util.ORACLE_LIB), util.HostAligner(util.ORACLE_LIB)]
    direct = [als[0].run(c["nodes"], c["edges"], c["read"], "align_pinned", pin_left=bool(c["args"][1])) for c in cases]
    arr = (ctypes.c_void_p * 2)(als[0].ptr, als[1].ptr)
    b = h.vgh_batch_create_multi(arr, 2, 5)
    graphs, errors = [None] * len(cases), []

    def worker(t, T):
        try:
            for k in range(t, len(cases), T):
                c = cases[k]
                g = h.vgh_graph_create(); graphs[k] = g
                for nid, seq in c["nodes"]:
                    assert h.vgh_graph_add_node(g, nid, seq.encode()) == 0
                for x, y in c["edges"]:
                    assert h.vgh_graph_add_edge(g, x, y) == 0
        except Exception as e:          # pragma: no cover
            errors.append(e)
    T = 4
    ts = [threading.Thread(target=worker, args=(t, T)) for t in range(T)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors
    # the submissions themselves from four threads AT ONCE: every thread submits into result slots of its own, nothing on this side orders
    # them (ctypes releases the GIL inside the calls) — the shim's own lock has to; small max_pending, so that submissions trigger flushes
    # while other threads are still submitting
    h.vgh_batch_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int]
    h.vgh_batch_add_slot.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    h.vgh_batch_reserve(b, len(cases))
    start = threading.Barrier(T)

    def submit(t):
        try:
            start.wait()
            for k in range(t, len(cases), T):
                assert h.vgh_batch_add_slot(b, k, graphs[k], cases[k]["read"].encode(), 2, int(bool(cases[k]["args"][1])), 1) == 0, h.vgh_last_error().decode()
        except Exception as e:          # pragma: no cover
            errors.append(e)
    ts = [threading.Thread(target=submit, args=(t,)) for t in range(T)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors
    order = list(range(len(cases)))
    buf = ctypes.create_string_buffer(1 << 24)
    assert h.vgh_batch_flush(b, buf, len(buf)) == 0, h.vgh_last_error().decode()
    out = json.loads(buf.value.decode())
    assert len(out) == len(cases)
    for k, got in zip(order, out):
        assert got == direct[k], cases[k]["source"]
    h.vgh_batch_destroy(b)
    for g in graphs:
        h.vgh_graph_destroy(g)


@pytest.mark.gpu
def test_alignment_batch_equals_direct_calls_on_hip():
    assert batch_vs_direct(util.ENGINE_LIB) > 45
