"""giraffe's alignment stage as one pipeline over the engine (vg_amd/pipeline.py): seeds -> gapless extensions -> for clusters no
full-length extension resolves, tail forests -> the trees as X-drop windows -> total scores (src/minimizer_mapper.cpp:5480-5535).
Every intermediate product of the engine (emulated here, the MI355X in the gpu test) must equal the oracle's: extension sets,
search states, forests, window alignments with their CIGAR ops, per-read totals."""
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, pipeline, workloads
from test_windows import assert_same


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def run_stage(lib, n, seed, graph_bp, inserted):
    wl = workloads.GaplessWorkload(n, seed=seed, graph_bp=graph_bp, inserted_reads=inserted)
    olen = np.repeat(np.array([len(s) for s in wl.nodes]), 2)
    outs = []
    for which in (lib, ORACLE_LIB):
        eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=which)
        outs.append(pipeline.align_stage(eng, eng.haplo_index(wl.nodes, wl.threads), olen, wl.gs))
    a, b = outs
    for f in ("status", "n_ext", "full_length"):
        assert (a["res"][f] == b["res"][f]).all(), f
    ne = int(a["res"]["n_ext"].sum())
    for f in ("offset", "read_begin", "read_end", "score", "left_full", "right_full", "state", "path_len"):
        assert (a["ext"][f][:ne] == b["ext"][f][:ne]).all(), f
    for f in ("status", "first_node", "n_nodes", "n_trees", "root_trim", "bases"):
        assert (a["tail_results"][f] == b["tail_results"][f]).all(), f
    for x, y in zip(a["forest"].fetch(), b["forest"].fetch()):
        assert (x == y).all()
    assert_same(a["tail_alignments"], a["tail_ops"], b["tail_alignments"], b["tail_ops"], "tails against their trees")
    for k in ("tail_score", "ext_total", "read_score"):
        assert (a[k] == b[k]).all(), k
    # the same stage with the glue in the host shim (vg_amd/host/tail_stage.cpp) instead of numpy
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    c = pipeline.align_stage_native(eng, eng.haplo_index(wl.nodes, wl.threads), olen, wl.gs)
    assert (c["ext_total"] == a["ext_total"][:len(c["ext_total"])]).all() and (c["read_score"] == a["read_score"]).all()
    assert c["stats"][0] == len(a["tails"]["problems"]) and c["stats"][1] == len(a["owner"]) and c["stats"][3] == 0
    # ... and with everything behind the extension on the device (vgk_tail_stage)
    d = pipeline.align_stage_device(eng, eng.haplo_index(wl.nodes, wl.threads), wl.gs)
    assert (d["ext_total"] == c["ext_total"]).all() and (d["read_score"] == c["read_score"]).all() and d["stats"] == c["stats"]
    # ... and the tails' alignments chosen on the device (vgk_tail_stage_aligned) = the best tree's alignment of the ORACLE's pipeline
    e = pipeline.align_stage_device(eng, eng.haplo_index(wl.nodes, wl.threads), wl.gs, aligned=True)
    assert (e["ext_total"] == c["ext_total"]).all() and (e["read_score"] == c["read_score"]).all() and e["stats"] == c["stats"]
    want = pipeline.winning_alignments(b)
    tails, tops = e["tails"], e["tail_ops"]
    assert len(tails) == len(want)
    # (the flat-array form of the same comparison, the one bench.py runs over a million reads: it must say what the row-by-row one says)
    arrays = pipeline.winning_alignment_arrays(b)
    verdict = pipeline.compare_tail_alignments(tails, tops, arrays)
    assert verdict["tails"] == verdict["identical"] == len(want) and verdict["first_bad"] is None and verdict["ops"] == sum(len(w[6]) for w in want)
    at = 0
    for i, wrow in enumerate(want):
        k = int(arrays["n_ops"][i])
        row = (int(arrays["ext"][i]), int(arrays["left"][i]), int(arrays["read_begin"][i]), int(arrays["read_end"][i]), int(arrays["score"][i]), int(arrays["first_offset"][i]),
               [(int(arrays["ops_node"][at + j]), int(arrays["ops_op"][at + j]), int(arrays["ops_len"][at + j])) for j in range(k)])
        assert row == wrow, (i, row, wrow)
        at += k
    if len(tops):                                                       # ... and it notices one op changed
        spoiled = tops.copy(); spoiled["len"][len(spoiled) // 2] += 1
        assert pipeline.compare_tail_alignments(tails, spoiled, arrays)["identical"] == len(want) - 1
    assert pipeline.compare_extension_sets(e["res"], e["ext"], e["nodes"], b["res"], b["ext"], b["nodes"], len(b["res"])) == len(b["res"])
    for i, wrow in enumerate(want):
        tl = tails[i]
        ops = tops[tl["ops_begin"]:tl["ops_begin"] + tl["n_ops"]]
        got = (int(tl["ext"]), int(tl["left"]), int(tl["read_begin"]), int(tl["read_end"]), int(tl["score"]), int(tl["first_offset"]),
               [(int(x["node"]), int(x["op"]), int(x["len"])) for x in ops])
        assert got == wrow, (i, got, wrow)
    # size-independent properties of every returned alignment: its read bases fit the tail (pinned at the extension: all of them unless
    # the end is soft-clipped), it starts inside its first node, consecutive nodes are steps some haplotype takes, the graph bases it
    # spends on a node fit the node, and re-scoring its ops against the sequences gives its score
    lens = [len(s) for s in wl.nodes]
    steps = set()
    for t in wl.threads:
        for x, y in zip(t[:-1], t[1:]):
            steps.add((x, y)); steps.add((y ^ 1, x ^ 1))
    comp = str.maketrans("ACGT", "TGCA")
    oseq = lambda o: wl.nodes[o >> 1] if not (o & 1) else wl.nodes[o >> 1].translate(comp)[::-1]
    reads_flat = wl.gs.reads.tobytes().decode(); roff = wl.gs.read_off
    read_of_ext = np.repeat(np.arange(len(e["res"])), e["res"]["n_ext"])
    rescored = 0
    for tl in tails:
        ops = tops[tl["ops_begin"]:tl["ops_begin"] + tl["n_ops"]]
        if not len(ops):
            assert tl["score"] == 0
            continue
        used = int(ops["len"][ops["op"] != capi.OP_D].sum())
        assert used <= tl["read_end"] - tl["read_begin"]
        assert tl["first_offset"] < lens[int(ops["node"][0]) >> 1] or int(ops["len"][0]) == 0
        r = int(read_of_ext[tl["ext"]]); rd = reads_flat[roff[r]:roff[r + 1]]
        tail = rd[tl["read_begin"]:tl["read_end"]]
        if tl["left"]:
            tail = tail.translate(comp)[::-1]
        # the tail's alignment continues its extension: it starts where the extension's matches stop — on the same node right behind the
        # last matched base, or, when the match ran to the node's end, at the start of a node some haplotype steps to (a left tail:
        # the same seen from the other strand, going outwards from the extension's first base)
        x = e["ext"][tl["ext"]]
        path = [int(v) for v in e["nodes"][x["path_begin"]:x["path_begin"] + x["path_len"]]]
        matched = int(x["read_end"]) - int(x["read_begin"])
        if not tl["left"]:
            assert tl["read_begin"] == x["read_end"] and tl["read_end"] == len(rd)
            last = path[-1]; cut = int(x["offset"]) + matched - sum(lens[v >> 1] for v in path[:-1])
        else:
            assert tl["read_begin"] == 0 and tl["read_end"] == x["read_begin"]
            last = path[0] ^ 1; cut = lens[path[0] >> 1] - int(x["offset"])
        first = int(ops["node"][0])
        if cut < lens[last >> 1]:
            assert first == last and tl["first_offset"] == cut, (tl, x, path)
        else:
            assert (last, first) in steps and tl["first_offset"] == 0, (tl, x, path)
        node = int(ops["node"][0]); off = int(tl["first_offset"]); at = 0; score = 0; prev_gap = None
        for x in ops:
            if int(x["node"]) != node:
                assert (node, int(x["node"])) in steps, (node, int(x["node"]))
                node = int(x["node"]); off = 0
            n = int(x["len"]); kind = int(x["op"])
            if kind == capi.OP_M:
                g = oseq(node)[off:off + n]; assert len(g) == n
                score += sum(1 if a == b and a in "ACGT" else -4 for a, b in zip(tail[at:at + n], g)); at += n; off += n; prev_gap = None
            elif kind == capi.OP_I:
                score -= (6 if prev_gap != kind else 1) + (n - 1); at += n; prev_gap = kind
            elif kind == capi.OP_D:
                assert off + n <= lens[node >> 1]
                score -= (6 if prev_gap != kind else 1) + (n - 1); off += n; prev_gap = kind
            else:
                prev_gap = None                                        # a soft clip at the far end
        if at == len(tail) and ops["op"][-1] == capi.OP_M:
            score += 5                                                 # the full-length bonus at the tail's far end
        if "N" not in tail:
            assert score == tl["score"], (tl, ops, score)
            rescored += 1
    assert rescored > len(tails) // 2
    return wl, a


def test_alignment_stage_equals_the_oracles(emu_lib):
    wl, a = run_stage(emu_lib, 1500, 9, 60000, 0.4)
    res, t = a["res"], a["tails"]
    assert 300 < int((res["full_length"] == 0).sum()) < 900 and len(t["problems"]) > 500
    assert (a["tail_results"]["n_trees"] > 1).any()                         # cuts at a node's end: forests
    # a read with one inserted base: everything matches but the gap (150 - 1 matches + 2 bonuses - open - extend ... minus substitutions)
    open_reads = res["full_length"] == 0
    assert np.median(a["read_score"][open_reads]) >= 140
    assert (a["read_score"][~open_reads] >= a["read_score"][open_reads].min()).all()


@pytest.mark.gpu
def test_alignment_stage_on_the_gpu_equals_the_oracles():
    wl, a = run_stage(ENGINE_LIB, 30000, 10, 400000, 0.3)
    assert len(a["tails"]["problems"]) > 10000


def test_device_tail_stage_needs_the_sets_of_an_extension_call(emu_lib):
    wl = workloads.GaplessWorkload(50, seed=3, graph_bp=20000)
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=emu_lib)
    idx = eng.haplo_index(wl.nodes, wl.threads)
    with pytest.raises(capi.VgkError):
        eng.tail_stage(idx, 50, 10)                                      # nothing extended yet on this context
    res, ext, _, _ = eng.gapless_extend(idx, wl.gs)
    et, rs, stats = eng.tail_stage(idx, wl.gs.n, int(res["n_ext"].sum()))
    assert stats[0] == 0 and (et == ext["score"][:len(et)]).all()        # every cluster resolved: no tails, totals = the extensions' scores
    assert (rs == np.maximum.reduceat(et, res["ext_begin"][res["n_ext"] > 0])).all() if (res["n_ext"] > 0).all() else True
    et2, rs2, tails, ops, stats2 = eng.tail_stage_aligned(idx, wl.gs.n, int(res["n_ext"].sum()))
    assert len(tails) == 0 and len(ops) == 0 and (et2 == et).all()


def test_aligned_tail_stage_reports_the_sizes_it_needs(emu_lib):
    import ctypes
    wl = workloads.GaplessWorkload(300, seed=4, graph_bp=30000, inserted_reads=0.5)
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=emu_lib)
    idx = eng.haplo_index(wl.nodes, wl.threads)
    res, ext, _, _ = eng.gapless_extend(idx, wl.gs)
    n_ext = int(res["n_ext"].sum())
    et, rs, tails, ops, stats = eng.tail_stage_aligned(idx, wl.gs.n, n_ext)
    assert len(tails) > 50 and len(ops) > len(tails)
    ext_total = np.zeros(n_ext, dtype=np.int32); read_score = np.zeros(wl.gs.n, dtype=np.int32)
    small_t = np.zeros(10, dtype=capi.TAIL_ALIGNMENT_DT); small_o = np.zeros(10, dtype=capi.OP_DT); big_t = np.zeros(len(tails), dtype=capi.TAIL_ALIGNMENT_DT)
    written = (ctypes.c_size_t * 2)()
    call = lambda t, o: eng.lib.vgk_tail_stage_aligned(eng.h, idx.h, 32, ext_total.ctypes.data, n_ext, read_score.ctypes.data, t.ctypes.data, len(t),
                                                       o.ctypes.data, len(o), ctypes.byref(written), None)
    assert call(small_t, small_o) == -6 and written[0] == len(tails)                 # VGK_EOPS: the tails do not fit
    assert call(big_t, small_o) == -6 and (written[0], written[1]) == (len(tails), len(ops))     # ... the ops do not fit
    big_o = np.zeros(len(ops), dtype=capi.OP_DT)
    assert call(big_t, big_o) == 0 and (big_t == tails).all() and (big_o == ops).all()


def test_a_second_context_runs_over_the_first_ones_indexes(emu_lib):
    second_context_over_shared_indexes(emu_lib, 300)


@pytest.mark.gpu
def test_a_second_context_runs_over_the_first_ones_indexes_on_hip():
    second_context_over_shared_indexes(ENGINE_LIB, 20000)


def second_context_over_shared_indexes(emu_lib, n_reads):
    """the haplotype and minimizer indexes are read-only tables of the device: a second context (its own streams, scratch and stage state) is handed
    the first one's and gives the first one's results — what configs[2]'s two batches in flight do with ONE copy of the indexes (include/vgk.h
    "sharing an index")"""
    wl = workloads.GaplessWorkload(n_reads, seed=21, graph_bp=50000 if n_reads < 1000 else 2_000_000, inserted_reads=0.3)
    sc = capi.Scoring.simple(1, 4, 6, 1, 5)
    a = capi.Engine(sc, lib=emu_lib); b = capi.Engine(sc, lib=emu_lib)
    hi = a.haplo_index(wl.nodes, wl.threads); mi = a.minimizer_index(wl.nodes, wl.threads)

    class B:
        n = wl.n
    outs = []
    for eng in (a, b, a, b):                                               # in turn, as two batches in flight alternate
        so, _, _ = eng.minimizer_seeds(mi, hi, wl.gs.reads, wl.gs.read_off, keep_on_device=True)
        o = pipeline.align_stage_device(eng, hi, B, seeded=int(so[-1]), aligned=True)
        outs.append((so.copy(), o["read_score"].copy(), o["ext_total"].copy(), o["tails"].copy(), o["tail_ops"].copy(), o["stats"]))
    for o in outs[1:]:
        assert (o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all() and (o[2] == outs[0][2]).all() and o[5] == outs[0][5]
        assert o[3].tobytes() == outs[0][3].tobytes() and o[4].tobytes() == outs[0][4].tobytes()
    assert outs[0][5][0] > 20
    b.close(); a.close()


# configs[2]'s graph at a small size (a variant site every ~900 bases, nodes of at most 32), its haplotype index built with unary runs merged (VGAMD_HAPLO_MERGE):
# the whole stage from bare reads on the merged-run index — seeds in, sets, tails and tail alignments out in the graph's own nodes — against the
# oracle's node-by-node stage.
def config2_stage_on_merged_runs(lib, n_reads, ref_len, monkeypatch):
    monkeypatch.setenv("VGAMD_HAPLO_MERGE", "1")
    g = workloads.VariationGraph(ref_len=ref_len)
    wl = workloads.Config2Workload(n_reads, batch=n_reads, seed=3, graph=g)
    graph = (wl.node_len, wl.seq)
    reads, off = wl.batches[0]
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib); ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=ORACLE_LIB)
    hi = eng.haplo_index(graph, wl.threads); mi = eng.minimizer_index(graph, wl.threads)
    assert hi.search_nodes() < 0.4 * len(wl.node_len), (hi.search_nodes(), len(wl.node_len))      # the runs were merged

    class B:
        n = n_reads
    so, _, _ = eng.minimizer_seeds(mi, hi, reads, off, keep_on_device=True)
    got = pipeline.align_stage_device(eng, hi, B, seeded=int(so[-1]), aligned=True)
    ohi = ora.haplo_index(graph, wl.threads); omi = ora.minimizer_index(graph, wl.threads)
    oso, osd, _ = ora.minimizer_seeds(omi, ohi, reads, off)
    assert (so == oso).all()
    sub = capi.GaplessSet(reads, off, osd, oso, node_cap=len(osd) * 16, mism_cap=len(osd) * 12)
    want = pipeline.align_stage(ora, ohi, np.repeat(wl.node_len, 2), sub)
    assert pipeline.compare_extension_sets(got["res"], got["ext"], got["nodes"], want["res"], want["ext"], want["nodes"], n_reads) == n_reads
    assert (got["read_score"] == want["read_score"]).all() and (got["ext_total"] == want["ext_total"][:len(got["ext_total"])]).all()
    verdict = pipeline.compare_tail_alignments(got["tails"], got["tail_ops"], pipeline.winning_alignment_arrays(want))
    assert verdict["tails"] == verdict["identical"] > n_reads // 20, verdict
    return eng.gapless_last_redone()


def test_config2_stage_on_merged_runs_emulated(emu_lib, monkeypatch):
    config2_stage_on_merged_runs(emu_lib, 2500, 150_000, monkeypatch)


@pytest.mark.gpu
def test_config2_stage_on_merged_runs_on_hip(monkeypatch):
    config2_stage_on_merged_runs(ENGINE_LIB, 200_000, 3_000_000, monkeypatch)
