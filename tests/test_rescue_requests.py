"""vgk_rescue_requests — the rescue candidates of a batch of pairs and their requests, one lane per pair over the extension sets the stage left in
HBM (vg_amd/csrc/rescue_requests_device.hpp; MinimizerMapper::map_paired / attempt_rescue, src/minimizer_mapper.cpp:1793-1901, :3264-3348) —
against the table made by host threads over the fetched sets (vg_amd/host/rescue_requests.cpp) and the numpy statement of the same rule
(vg_amd/pipeline.py: paired_stage without a resident graph): every entry, every field, and the lost mates' reads as the rescue takes them."""
import ctypes
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ROOT
from vg_amd import capi, pipeline, workloads


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu", "host"], cwd=ROOT)
    return EMU_LIB


def host_table(wl, out, stdevs, threads=1):
    h = pipeline._host_lib()
    res, ext, nodes = np.ascontiguousarray(out["res"]), np.ascontiguousarray(out["ext"]), np.ascontiguousarray(out["nodes"], dtype=np.uint32)
    n_pairs = wl.n // 2; L = wl.read_len; g = wl.graph
    mapped = np.zeros(n_pairs, np.uint32); lost = np.zeros(n_pairs, np.uint32); req = np.zeros((n_pairs, 6), np.int64); rd = np.zeros(n_pairs * L, np.uint8)
    col = np.ascontiguousarray(g.col, dtype=np.int64)
    h.vgh_rescue_requests.restype = ctypes.c_int64
    h.vgh_rescue_requests.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 4
    m = h.vgh_rescue_requests(n_pairs, res.ctypes.data, ext.ctypes.data, nodes.ctypes.data, g.n_nodes, col.ctypes.data, wl.reads.ctypes.data, L, float(wl.mean), float(wl.sd),
                              float(stdevs), threads, mapped.ctypes.data, lost.ctypes.data, req.ctypes.data, rd.ctypes.data)
    assert m >= 0
    return mapped[:m], lost[:m], req[:m], rd[:m * L].reshape(m, L)


def device_table(eng, rg, wl, stdevs, threads=1):
    h = pipeline._host_lib()
    tab = eng.rescue_requests(rg.dgraph, wl.mean, wl.sd, stdevs)
    m = len(tab); L = wl.read_len
    mapped = np.zeros(m, np.uint32); lost = np.zeros(m, np.uint32); req = np.zeros((m, 6), np.int64); rd = np.zeros(max(m * L, 1), np.uint8)
    h.vgh_rescue_reads.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int] + [ctypes.c_void_p] * 4
    assert h.vgh_rescue_reads(m, tab.ctypes.data, wl.reads.ctypes.data, L, threads, mapped.ctypes.data, lost.ctypes.data, req.ctypes.data, rd.ctypes.data) == 0
    return tab, mapped, lost, req, rd[:m * L].reshape(m, L)


def check(lib, n_pairs, seed, hard, stdevs=4.0, ref_len=300_000):
    wl = workloads.PairedWorkload(n_pairs, ref_len=ref_len, seed=seed, hard=hard)
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    graph = (wl.node_len, wl.seq)
    index = eng.haplo_index(graph, wl.threads); mindex = eng.minimizer_index(graph, wl.threads)
    aligner = pipeline.HostAlignerHandle(lib)
    rg = aligner.rescue_graph(wl)

    class B:
        n = wl.n
    seed_off, _, _ = eng.minimizer_seeds(mindex, index, wl.reads, wl.read_off, keep_on_device=True)
    out = pipeline.align_stage_device(eng, index, B, seeded=int(seed_off[-1]), aligned=False)
    tab, mapped, lost, req, rd = device_table(eng, rg, wl, stdevs)
    h_mapped, h_lost, h_req, h_rd = host_table(wl, out, stdevs)
    assert len(tab) == len(h_mapped) and len(tab) > 0
    assert (mapped == h_mapped).all() and (lost == h_lost).all()
    bad = np.nonzero((req != h_req).any(axis=1))[0]
    assert len(bad) == 0, "requests %s: %s vs %s" % (bad[:5], req[bad[:5]], h_req[bad[:5]])
    assert (rd == h_rd).all()
    # the table's own statement of the strand: a mate is rescued as its reverse complement exactly when its partner's first node is a forward one
    first = out["nodes"][out["ext"]["path_begin"][out["res"]["ext_begin"][mapped]]]
    assert (tab["reverse"] == ((first & 1) == 0)).all()
    # both kinds of request occur: with a seed of the lost mate's own and without one; mates on either strand
    assert (req[:, 4] >= 0).any() and (req[:, 4] < 0).any() and tab["reverse"].any() and not tab["reverse"].all()
    # chunked threads give the same table
    t2 = device_table(eng, rg, wl, stdevs, threads=3)
    assert (t2[3] == req).all() and (t2[4] == rd).all()
    # a caller's array that is too small: the count needed, nothing written beyond it
    small = np.zeros(max(len(tab) // 2, 1), dtype=capi.RESCUE_REQUEST_DT); w = ctypes.c_size_t()
    rc = eng.lib.vgk_rescue_requests(eng.h, rg.dgraph, float(wl.mean), float(wl.sd), float(stdevs), small.ctypes.data, len(small), ctypes.byref(w))
    assert rc == capi.VGK_EOPS and w.value == len(tab) and not small["mapped"].any()
    # a narrower fragment model moves the windows: the table follows (and still equals the host's)
    tab_n, _, _, req_n, _ = device_table(eng, rg, wl, 1.0)
    assert (req_n == host_table(wl, out, 1.0)[2]).all() and (req_n[:, :2] != req[:, :2]).any()
    rg.close(); aligner.close()
    return len(tab)


def test_table_on_the_emulated_kernels(emu_lib):
    assert check(emu_lib, 1500, 11, 0.3) > 100


def test_table_refuses_what_is_not_a_batch_of_pairs(emu_lib):
    wl = workloads.PairedWorkload(20, ref_len=100_000, seed=3, hard=0.3)
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=emu_lib)
    aligner = pipeline.HostAlignerHandle(emu_lib); rg = aligner.rescue_graph(wl)
    out = np.zeros(64, dtype=capi.RESCUE_REQUEST_DT); w = ctypes.c_size_t(7)
    eng.lib.vgk_rescue_requests.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    # no extension call on this context yet: there are no sets
    assert eng.lib.vgk_rescue_requests(eng.h, rg.dgraph, 400.0, 40.0, 4.0, out.ctypes.data, len(out), ctypes.byref(w)) == capi.VGK_EINVAL and w.value == 0
    assert eng.lib.vgk_rescue_requests(eng.h, None, 400.0, 40.0, 4.0, out.ctypes.data, len(out), ctypes.byref(w)) == capi.VGK_EINVAL
    # an odd number of reads is not a batch of pairs
    graph = (wl.node_len, wl.seq)
    index = eng.haplo_index(graph, wl.threads); mindex = eng.minimizer_index(graph, wl.threads)
    L = wl.read_len
    seed_off, _, _ = eng.minimizer_seeds(mindex, index, wl.reads[:3 * L], wl.read_off[:4], keep_on_device=True)
    eng.gapless_extend_seeded(index, 3, int(seed_off[-1]))
    assert eng.lib.vgk_rescue_requests(eng.h, rg.dgraph, 400.0, 40.0, 4.0, out.ctypes.data, len(out), ctypes.byref(w)) == capi.VGK_EINVAL
    assert eng.lib.vgk_rescue_requests(eng.h, rg.dgraph, 400.0, -1.0, 4.0, out.ctypes.data, len(out), ctypes.byref(w)) == capi.VGK_EINVAL
    rg.close(); aligner.close()


@pytest.mark.gpu
def test_table_on_hip():
    assert check(ENGINE_LIB, 60_000, 12, 0.1, ref_len=2_000_000) > 3000
