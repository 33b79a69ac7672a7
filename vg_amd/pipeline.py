"""giraffe's alignment stage over the engine's entry points, batch-wise: seeds -> gapless extension (vgk_gapless_extend) -> for the
clusters that no full-length extension resolves, the tails of every extension: tail forests (vgk_tail_forest), every tree a window
of the forest graph (vgk_gssw_pack_windows), left-pinned X-drop -> the extension's total score (src/minimizer_mapper.cpp:5480-5535).

Which route is which (every route takes an engine handle and runs on whatever library that handle binds):

  product routes — what bench.py times on the HIP library and what a maintainer's patch would call
    align_stage_device   the stage with the tails on the device too (vgk_tail_stage); configs[2] and the paired slice run this
    align_stage_native   the stage with the glue behind the extension in the host shim (C++ threads); bench --workload giraffe
    ChainStage           configs[4]: the chain alignment of a batch of long reads behind one host-shim call
    paired_stage         configs[3] slice: both mates through align_stage_device, the lost mate through the native rescue stage
  CHECKER-SIDE ONLY — the stage spelled out in numpy, kept because it is what the OTHER side of every comparison runs (tests/,
  __graft_entry__.smoke() and bench.py's parity / cpu_baseline legs hand it the oracle's engine handle); nothing timed as the
  product goes through these, and they are not part of what a maintainer would take over
    align_stage, tails_of_extensions, tail_sequences, tree_windows, winning_alignments, winning_alignment_arrays, compare_tail_alignments,
    compare_extension_sets, chain_stage"""
import ctypes
import os

import numpy as np

from . import capi, workloads

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def tails_of_extensions(oriented_len, read_off, res, ext, nodes, scoring=(1, 6, 1, 5)):
    """[checker-side: the numpy statement of what run_tail_stage / vgk_tail_stage do natively]
    -> dict of numpy arrays, one entry per TAIL (an extension end that does not reach its read end, of a read whose extension set
    is not full-length): problems (TAIL_DT), ext (index into `ext`), left (bool), read (index), begin / end (the tail's interval of
    the read), gap (longest detectable gap)"""
    n = len(res)
    read_of_ext = np.repeat(np.arange(n), res["n_ext"])
    ext = ext[:len(read_of_ext)]
    L = np.diff(read_off)[read_of_ext]
    open_set = (res["status"] == 0) & (res["full_length"] == 0)
    cand = open_set[read_of_ext]
    nl = oriented_len[nodes]
    csum = np.concatenate([[0], np.cumsum(nl)])
    pb = ext["path_begin"].astype(np.int64); pl = ext["path_len"].astype(np.int64)
    path_bases = csum[pb + pl] - csum[pb]
    last = nodes[np.maximum(pb + pl - 1, 0)]; first = nodes[pb]
    matched = ext["read_end"].astype(np.int64) - ext["read_begin"]
    out = []
    # right tails: behind the last matched base, on the forward state (:5768-5775)
    r = np.nonzero(cand & (ext["right_full"] == 0) & (pl > 0))[0]
    pr = np.zeros(len(r), dtype=capi.TAIL_DT)
    pr["node"] = ext["state"][r, 0]; pr["lo"] = ext["state"][r, 1].astype(np.int32); pr["hi"] = ext["state"][r, 2].astype(np.int32)
    pr["offset"] = ext["offset"][r] + matched[r] - (path_bases[r] - oriented_len[last[r]])
    tail_r = L[r] - ext["read_end"][r]
    # left tails: before the first matched base, looking the other way on the backward state (:5756-5766)
    l = np.nonzero(cand & (ext["left_full"] == 0) & (pl > 0))[0]
    plft = np.zeros(len(l), dtype=capi.TAIL_DT)
    plft["node"] = ext["state"][l, 3]; plft["lo"] = ext["state"][l, 4].astype(np.int32); plft["hi"] = ext["state"][l, 5].astype(np.int32)
    plft["offset"] = oriented_len[first[l] ^ 1] - ext["offset"][l]
    tail_l = ext["read_begin"][l].astype(np.int64)
    problems = np.concatenate([pr, plft])
    tail_len = np.concatenate([tail_r, tail_l]).astype(np.int64)
    e = np.concatenate([r, l]); left = np.concatenate([np.zeros(len(r), bool), np.ones(len(l), bool)])
    gap = workloads.longest_detectable_gap(L[e], tail_len, *scoring)
    problems["walk_distance"] = tail_len + gap                                   # (:5816)
    begin = np.where(left, 0, ext["read_end"][e]); end = np.where(left, ext["read_begin"][e], L[e])
    return dict(problems=problems, ext=e, left=left, read=read_of_ext[e], begin=begin.astype(np.int64), end=end.astype(np.int64), gap=np.asarray(gap, dtype=np.int64))


def tail_sequences(reads, read_off, t):
    """[checker-side] the tails' bases, flat: a right tail as it is, a left tail reverse-complemented (:5660) -> (bases, offsets)"""
    ln = (t["end"] - t["begin"]).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(ln)])
    left = t["left"]
    base = read_off[t["read"]]
    start = np.where(left, base + t["end"] - 1, base + t["begin"])               # first base taken, then forward (right tail) or backward (left tail)
    step = np.where(left, -1, 1)
    # position k of tail i = start[i] + step[i] * k: one running index built from per-tail increments
    inc = np.repeat(step, ln)
    first = off[:-1][ln > 0]
    inc[first] = start[ln > 0] - np.concatenate([[0], (start + step * (ln - 1))[ln > 0][:-1]])
    idx = np.cumsum(inc)
    b = reads[idx]
    lm = np.repeat(left, ln)
    b[lm] = _COMP[b[lm]]
    return b, off


def tree_windows(res, forest, seq, seq_off, gap):
    """[checker-side] one left-pinned X-drop window problem per tree -> (WindowSet, tail index of every tree)"""
    flags = capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK
    if (res["n_trees"] <= 1).all():
        has = np.nonzero(res["n_nodes"] > 0)[0]
        first = res["first_node"][has]; count = res["n_nodes"][has]; cols = res["bases"][has].astype(np.int64); owner = has
    else:
        parent, _, length = forest.fetch()
        first = np.nonzero(parent < 0)[0]
        count = np.diff(np.concatenate([first, [len(parent)]]))
        has = np.nonzero(res["n_nodes"] > 0)[0]
        owner = has[np.searchsorted(res["first_node"][has], first, side="right") - 1]
        csum = np.concatenate([[0], np.cumsum(length, dtype=np.int64)])
        cols = csum[first + count] - csum[first]
    if len(owner) == len(seq_off) - 1 and (owner == np.arange(len(owner))).all():
        return capi.WindowSet(seq, seq_off, first, count, flags, gap, cols=cols), owner       # every tail has one tree: its bases as they lie
    ln = np.diff(seq_off)[owner]
    off = np.concatenate([[0], np.cumsum(ln)])
    inc = np.ones(off[-1], dtype=np.int64)
    nz = ln > 0
    last = seq_off[owner] + ln - 1
    inc[off[:-1][nz]] = seq_off[owner][nz] - np.concatenate([[0], last[nz][:-1]])
    bases = seq[np.cumsum(inc)]
    return capi.WindowSet(bases, off, first, count, flags, gap[owner], cols=cols), owner


def align_stage(eng, index, oriented_len, gs, ops_per_problem=32, timing=None):
    """[checker-side: the route the oracle's handle is driven through in tests, smoke() and bench.py's parity legs; bench.py times it on
    the HIP library only when VGAMD_GIRAFFE_NUMPY_GLUE asks for the comparison of glue costs]
    The whole stage for one batch of clusters (a GaplessSet).  -> dict: the extension outputs, the tails, per tail its best tree's
    score, per extension its total score, per read the best total; `forest`, `batch` results for parity checks.  timing: a dict that
    collects seconds per step."""
    import time
    t0 = [time.perf_counter()]

    def lap(what):
        if timing is not None:
            t = time.perf_counter(); timing[what] = timing.get(what, 0.0) + t - t0[0]; t0[0] = t

    res, ext, nodes, mism = eng.gapless_extend(index, gs); lap("gapless_extend")
    t = tails_of_extensions(oriented_len, gs.read_off, res, ext, nodes); lap("tails_of_extensions (host)")
    out = dict(res=res, ext=ext, nodes=nodes, mism=mism, tails=t)
    total = ext["score"].astype(np.int64).copy()
    if len(t["problems"]):
        tres, forest = eng.tail_forest(index, t["problems"]); lap("tail_forest")
        seq, seq_off = tail_sequences(gs.reads, gs.read_off, t); lap("tail_sequences (host)")
        ws, owner = tree_windows(tres, forest, seq, seq_off, t["gap"]); lap("tree_windows (host)")
        with eng.pack_windows(forest.graph, ws, ops_per_problem) as b:
            lap("pack_windows")
            b.run(); b.sync(); lap("fill + traceback")
            r, ops = b.fetch(); lap("fetch")
        best = np.zeros(len(t["problems"]), dtype=np.int64)                       # a tail nothing aligns to is a soft clip: 0 (:5632-5648)
        np.maximum.at(best, owner, np.where(r["status"] == 0, r["score"], 0))
        np.add.at(total, t["ext"], best)
        out.update(tail_results=tres, forest=forest, windows=ws, owner=owner, tail_alignments=r, tail_ops=ops, tail_score=best)
    read_of_ext = np.repeat(np.arange(len(res)), res["n_ext"])
    best_read = np.zeros(len(res), dtype=np.int64)
    np.maximum.at(best_read, read_of_ext, total[:len(read_of_ext)])
    out.update(ext_total=total, read_score=best_read); lap("totals (host)")
    return out


# ---- the same stage with the host glue in C++ (vg_amd/host/tail_stage.cpp, part of the host shim) ---------------------------------
_HOST = None


def _host_lib():
    global _HOST
    if _HOST is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvgamd_host.so")
        if not os.path.exists(path):
            raise RuntimeError("vg_amd/libvgamd_host.so missing: run `make host`")
        h = ctypes.CDLL(path)
        h.vgh_tail_stage.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        h.vgh_last_error.restype = ctypes.c_char_p
        _HOST = h
    return _HOST


def gaf_lines(seqs, seq_off, results, mappings, runs, node_seq, node_off, names=None, node_ids=None, mapq=None, score=None, threads=4):
    """GAF records of a batch of composed alignments (vgh_gaf_lines, vg_amd/host/gaf_output.cpp: what alignment_to_gaf + the emitter write in vg).
    seqs / seq_off: the reads' bases behind each other; results / mappings / runs: vgk_chain_stitch's output (ChainStage.run(compose=True)["alignments"]);
    node_seq / node_off: the graph's forward strands; names: a list of bytes or None; node_ids: int64 names of the nodes (None: index + 1).
    -> the lines (bytes, one per read, without the newline)"""
    import numpy as np
    h = _host_lib()
    h.vgh_gaf_lines.restype = ctypes.c_int64
    h.vgh_gaf_lines.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    n = len(results)
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8); seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    results = np.ascontiguousarray(results); mappings = np.ascontiguousarray(mappings); runs = np.ascontiguousarray(runs, dtype=np.uint32)
    node_seq = np.ascontiguousarray(node_seq, dtype=np.uint8); node_off = np.ascontiguousarray(node_off, dtype=np.uint64)
    name_arr = (ctypes.c_char_p * n)(*names) if names is not None else None
    ids = np.ascontiguousarray(node_ids, dtype=np.int64) if node_ids is not None else None
    mq = np.ascontiguousarray(mapq, dtype=np.int32) if mapq is not None else None
    sc = np.ascontiguousarray(score, dtype=np.int32) if score is not None else None
    line_off = np.zeros(n + 1, dtype=np.uint64)
    bad = ctypes.c_uint32(0)
    ptr = lambda a: a.ctypes.data if a is not None else None
    def call(buf):
        return h.vgh_gaf_lines(n, name_arr, ptr(seqs), ptr(seq_off), ptr(results), ptr(mappings), ptr(runs), ptr(node_seq), ptr(node_off), len(node_off) - 1,
                               ptr(ids), ptr(mq), ptr(sc), ptr(buf), 0 if buf is None else len(buf), ptr(line_off), int(threads), ctypes.byref(bad))
    need = call(None)
    if need < 0:
        raise ValueError("gaf_lines: the alignment of read %d does not fit the graph or its own length" % bad.value)
    out = np.empty(need, dtype=np.uint8)
    if call(out) != need:
        raise RuntimeError("gaf_lines: size changed between the two calls")
    text = out.tobytes()
    return [text[int(line_off[r]):int(line_off[r + 1]) - 1] for r in range(n)]


def align_stage_native(eng, index, oriented_len, gs, ops_per_problem=32, scoring=(1, 6, 1, 5), timing=None, seeded=None):
    """align_stage with everything behind the gapless extension done by the host shim's run_tail_stage (C++ threads instead of numpy)
    -> dict(res, ext, nodes, ext_total, read_score, stats = (tails, trees, tree nodes, failed), stage_ms)"""
    import time
    t0 = time.perf_counter()
    if seeded is not None:          # the clusters are on the device already (minimizer_seeds(keep_on_device=True)): seeded = number of seeds
        res, ext, nodes, mism = eng.gapless_extend_seeded(index, gs.n, int(seeded))
    else:
        res, ext, nodes, mism = eng.gapless_extend(index, gs)
    t1 = time.perf_counter()
    h = _host_lib()
    n_ext = int(res["n_ext"].sum())
    read_off = np.ascontiguousarray(gs.read_off, dtype=np.uint64); olen = np.ascontiguousarray(oriented_len, dtype=np.uint32)
    ext_total = np.zeros(max(n_ext, 1), dtype=np.int32); read_score = np.zeros(max(gs.n, 1), dtype=np.int32)
    sc = np.array(scoring, dtype=np.int32); stats = np.zeros(4, dtype=np.uint64); ms = np.zeros(6, dtype=np.float64)
    ext = np.ascontiguousarray(ext); nodes = np.ascontiguousarray(nodes)
    rc = h.vgh_tail_stage(eng.lib._name.encode(), eng.h, index.h, gs.reads.ctypes.data, read_off.ctypes.data, gs.n, res.ctypes.data, ext.ctypes.data,
                          nodes.ctypes.data, olen.ctypes.data, sc.ctypes.data, ops_per_problem, ext_total.ctypes.data, len(ext_total), read_score.ctypes.data,
                          stats.ctypes.data, ms.ctypes.data)
    if rc:
        raise RuntimeError("vgh_tail_stage: " + (h.vgh_last_error() or b"?").decode())
    t2 = time.perf_counter()
    if timing is not None:
        for k, v in (("gapless_extend", t1 - t0), ("tail stage (host shim, total)", t2 - t1)):
            timing[k] = timing.get(k, 0.0) + v
        for k, v in zip(("tails derived (host)", "tail_forest", "windows + bases (host)", "pack_windows", "fill + traceback + fetch", "totals (host)"), ms):
            timing[k] = timing.get(k, 0.0) + v * 1e-3
    return dict(res=res, ext=ext, nodes=nodes, ext_total=ext_total[:n_ext], read_score=read_score[:gs.n], stats=tuple(int(x) for x in stats))


# ---- configs[4]: a long read cut at its anchors (workloads.LongReadWorkload) -------------------------------------------------------
def chain_stage(eng, index, wl, match=1, timing=None):
    """[checker-side since ChainStage exists: tests/test_longread_stage.py holds ChainStage to it]
    Every stretch between anchors through WFAExtender (vgk_wfa_extend: connect / prefix / suffix); the connects it gives up on —
    score cap, tables — through BandedGlobalAligner between the two anchors (vgk_banded_align), as giraffe's chain alignment does
    (src/minimizer_mapper.cpp:2955-3100).  -> dict: wfa results, the fallback's results, per-read chain score."""
    import time
    t0 = [time.perf_counter()]

    def lap(what):
        if timing is not None:
            t = time.perf_counter(); timing[what] = timing.get(what, 0.0) + t - t0[0]; t0[0] = t

    res, paths, edits = eng.wfa_extend(index, wl.ws); lap("wfa_extend")
    mode = wl.ws.array["mode"]
    failed = np.nonzero((mode == capi.WFA_CONNECT) & ((res["status"] != 0) | (res["ok"] == 0)))[0]
    score = np.where((res["status"] == 0) & (res["ok"] != 0), res["score"], 0).astype(np.int64)
    out = dict(wfa=res, failed=failed, declined_tails=np.nonzero((mode != capi.WFA_CONNECT) & (res["status"] != 0))[0])      # (a tail the engine declines scores 0 here: vg's own route for it is the pinned X-drop of §17)
    if len(failed):
        if getattr(wl, "connects", None) is not None:                 # the caller keeps the subgraphs between its anchors flat: pick the batch out of them
            bs = wl.connects.select(wl.connect_row[failed])
        else:
            bs = capi.BandedSet.from_lists([wl.between(int(i)) for i in failed])
        lap("fallback problems (host)")
        bres, bops = eng.banded_align(bs); lap("banded_align")
        score[failed] = np.where(bres["status"] == 0, bres["score"], 0)
        out.update(banded=bres, banded_ops=bops)
    chain = np.zeros(wl.n_reads, dtype=np.int64)
    np.add.at(chain, wl.read_of, score)
    chain += wl.anchor_bases * match
    out.update(segment_score=score, chain_score=chain); lap("totals (host)")
    return out


def align_stage_device(eng, index, gs, ops_per_problem=32, timing=None, seeded=None, aligned=False):
    """The stage with everything behind the extension on the device too (vgk_tail_stage): the extension sets come back for the caller, the
    tails are derived, walked, packed and aligned from what the extension call left in HBM.  -> dict(res, ext, nodes, ext_total, read_score, stats); aligned:
    also the tails' winning alignments (vgk_tail_stage_aligned: `tails`, `tail_ops`)"""
    import time
    t0 = time.perf_counter()
    if seeded is not None:
        # (deferred: the sets come down on a side stream while the tail stage runs; `ext` is sized already, filled when that call returns)
        res, ext, nodes, mism = eng.gapless_extend_seeded(index, gs.n, int(seeded), defer=True)
    else:
        res, ext, nodes, mism = eng.gapless_extend(index, gs, defer=True)
    t1 = time.perf_counter()
    tails = tail_ops = None
    if aligned:
        ext_total, read_score, tails, tail_ops, stats = eng.tail_stage_aligned(index, gs.n, len(ext), ops_per_problem)
    else:
        ext_total, read_score, stats = eng.tail_stage(index, gs.n, len(ext), ops_per_problem)
    t2 = time.perf_counter()
    if timing is not None:
        for k, v in (("gapless_extend", t1 - t0), ("tail stage (device, total)", t2 - t1)):
            timing[k] = timing.get(k, 0.0) + v
        for k, v in zip(("tails derived (device)", "tail forest (device)", "windows packed (device)", "fill + traceback + totals"), eng.tail_stage_last_ms()):
            timing[k] = timing.get(k, 0.0) + v * 1e-3
    return dict(res=res, ext=ext, nodes=nodes, ext_total=ext_total, read_score=read_score, stats=stats, tails=tails, tail_ops=tail_ops)


def winning_alignments(out):
    """[checker-side] From align_stage's output (every tree's alignment): per tail what vgk_tail_stage_aligned reports — the best tree's alignment, the
    first among equals, nothing for a soft clip; nodes translated to oriented nodes of the index.  -> list of (ext, left, read_begin,
    read_end, score, first_offset, [(node, op, len)])"""
    t = out["tails"]
    nt = len(t["problems"])
    rows = [None] * nt
    if "tail_alignments" not in out:
        return [(int(t["ext"][i]), int(t["left"][i]), int(t["begin"][i]), int(t["end"][i]), 0, 0, []) for i in range(nt)]
    r, ops, owner, tres = out["tail_alignments"], out["tail_ops"], out["owner"], out["tail_results"]
    parent, fnode, _ = out["forest"].fetch()
    first_node = out["windows"].array["first_node"]
    best = {}
    for w in range(len(owner)):
        if r["status"][w] == 0 and r["score"][w] > 0 and (owner[w] not in best or r["score"][w] > r["score"][best[owner[w]]]):
            best[int(owner[w])] = w
    for i in range(nt):
        row = [int(t["ext"][i]), int(t["left"][i]), int(t["begin"][i]), int(t["end"][i]), 0, 0, []]
        if i in best:
            w = best[i]
            base = int(first_node[w])
            o = ops[r["ops_begin"][w]:r["ops_begin"][w] + r["n_ops"][w]]
            row[4] = int(r["score"][w])
            if len(o):
                v0 = base + int(o["node"][0])
                row[5] = int(r["first_offset"][w]) + (int(tres["root_trim"][i]) if parent[v0] < 0 else 0)
            row[6] = [(int(fnode[base + int(x["node"])]), int(x["op"]), int(x["len"])) for x in o]
        rows[i] = tuple(row)
    return rows


def winning_alignment_arrays(out):
    """[checker-side] winning_alignments as flat arrays, for comparisons at bench scale (a million reads' tails): -> dict(ext, left, read_begin,
    read_end, score, first_offset, n_ops — one entry per tail, in align_stage's tail order — and ops_node / ops_op / ops_len, the winners' ops
    behind each other, nodes translated to oriented nodes of the index)"""
    t = out["tails"]
    nt = len(t["problems"])
    res = dict(ext=t["ext"].astype(np.int64), left=t["left"].astype(np.int64), read_begin=t["begin"].astype(np.int64), read_end=t["end"].astype(np.int64),
               score=np.zeros(nt, dtype=np.int64), first_offset=np.zeros(nt, dtype=np.int64), n_ops=np.zeros(nt, dtype=np.int64),
               ops_node=np.zeros(0, dtype=np.int64), ops_op=np.zeros(0, dtype=np.int64), ops_len=np.zeros(0, dtype=np.int64))
    if "tail_alignments" not in out or nt == 0:
        return res
    r, ops, owner, tres = out["tail_alignments"], out["tail_ops"], np.asarray(out["owner"], dtype=np.int64), out["tail_results"]
    parent, fnode, _ = out["forest"].fetch()
    first_node = out["windows"].array["first_node"].astype(np.int64)
    sc = np.where(r["status"] == 0, r["score"], 0).astype(np.int64)
    order = np.lexsort((np.arange(len(owner)), -sc, owner))              # per tail: the best tree, the first among equals
    lead = order[np.concatenate([[True], owner[order][1:] != owner[order][:-1]])] if len(order) else order
    win = lead[sc[lead] > 0]                                             # (a soft clip otherwise: nothing)
    tail = owner[win]
    res["score"][tail] = sc[win]
    n_ops = r["n_ops"][win].astype(np.int64)
    res["n_ops"][tail] = n_ops
    # the winners' ops, in tail order
    by_tail = np.argsort(tail, kind="stable")
    win, tail, n_ops = win[by_tail], tail[by_tail], n_ops[by_tail]
    begin = r["ops_begin"][win].astype(np.int64)
    seg = np.cumsum(n_ops) - n_ops
    idx = np.repeat(begin, n_ops) + (np.arange(int(n_ops.sum())) - np.repeat(seg, n_ops))
    o = ops[idx]
    base = np.repeat(first_node[win], n_ops)
    res["ops_node"] = fnode[base + o["node"].astype(np.int64)].astype(np.int64); res["ops_op"] = o["op"].astype(np.int64); res["ops_len"] = o["len"].astype(np.int64)
    has = n_ops > 0
    v0 = first_node[win][has] + o["node"][seg[has]].astype(np.int64)
    res["first_offset"][tail[has]] = r["first_offset"][win][has].astype(np.int64) + np.where(parent[v0] < 0, tres["root_trim"][tail[has]].astype(np.int64), 0)
    return res


def compare_tail_alignments(tails, tail_ops, want, n_ext=None):
    """[checker-side] vgk_tail_stage_aligned's output against winning_alignment_arrays(oracle stage): -> dict(tails, identical, ops, first_bad).
    n_ext: only the tails of extensions below it (the oracle ran a prefix of the batch's reads)."""
    if n_ext is not None:
        tails = tails[tails["ext"] < n_ext]
    nt = len(want["ext"])
    if len(tails) != nt:
        return dict(tails=nt, identical=0, ops=int(want["n_ops"].sum()), first_bad=-1, note="%d tails from the engine, %d from the oracle" % (len(tails), nt))
    same = np.ones(nt, dtype=bool)
    for f in ("ext", "left", "read_begin", "read_end", "score", "first_offset", "n_ops"):
        same &= tails[f].astype(np.int64) == want[f]
    if same.all() and nt:
        n_ops = want["n_ops"]
        seg = np.cumsum(n_ops) - n_ops
        idx = np.repeat(tails["ops_begin"].astype(np.int64), n_ops) + (np.arange(int(n_ops.sum())) - np.repeat(seg, n_ops))
        o = tail_ops[idx]
        bad = (o["node"].astype(np.int64) != want["ops_node"]) | (o["op"].astype(np.int64) != want["ops_op"]) | (o["len"].astype(np.int64) != want["ops_len"])
        if bad.any():
            same[np.unique(np.repeat(np.arange(nt), n_ops)[bad])] = False
    bad = np.nonzero(~same)[0]
    return dict(tails=nt, identical=int(same.sum()), ops=int(want["n_ops"].sum()), first_bad=int(bad[0]) if len(bad) else None)


def compare_extension_sets(a_res, a_ext, a_nodes, b_res, b_ext, b_nodes, k):
    """[checker-side] the extension sets of the first k reads of two runs (results, extensions with their search states, path nodes): -> reads whose
    set is identical (every field of every extension, every node of every path)"""
    same = np.ones(k, dtype=bool)
    for f in ("status", "n_ext", "full_length"):
        same &= a_res[f][:k] == b_res[f][:k]
    if not same.all():
        return int(same.sum())
    n_ext = a_res["n_ext"][:k].astype(np.int64)
    tot = int(n_ext.sum())
    read_of = np.repeat(np.arange(k), n_ext)
    ea, eb = a_ext[:tot], b_ext[:tot]
    bad = np.zeros(tot, dtype=bool)
    for f in ("path_len", "offset", "read_begin", "read_end", "n_mismatches", "score", "left_full", "right_full"):
        bad |= ea[f] != eb[f]
    bad |= (ea["state"] != eb["state"]).any(axis=1)
    pl = ea["path_len"].astype(np.int64)
    ok = ~bad
    # the paths of the extensions that agree so far, node by node (each run's own path_begin)
    seg = np.cumsum(pl[ok]) - pl[ok]
    run = np.arange(int(pl[ok].sum())) - np.repeat(seg, pl[ok])
    na = a_nodes[np.repeat(ea["path_begin"][ok].astype(np.int64), pl[ok]) + run]; nb = b_nodes[np.repeat(eb["path_begin"][ok].astype(np.int64), pl[ok]) + run]
    diff = na != nb
    if diff.any():
        bad[np.nonzero(ok)[0][np.unique(np.repeat(np.arange(int(ok.sum())), pl[ok])[diff])]] = True
    same[np.unique(read_of[bad])] = False
    return int(same.sum())


# ---- seeding reads of any length: every minimizer listed on the device, find_seeds' choice in the host shim, the seeds of the chosen on the device ----
# giraffe's defaults where the long-read presets leave them (src/minimizer_mapper.hpp: hit_cap 10, hard_hit_cap 500, minimizer_score_fraction 0.9, max_unique_min 500,
# num_bp_per_min 1000, minimizer_coverage_flank 250; no window downsampling, overlapping minimizers kept)
GIRAFFE_LONG_READ_POLICY = dict(hit_cap=10, hard_hit_cap=500, max_unique_min=500, num_bp_per_min=1000, exclude_overlapping_min=False, coverage_flank=250, window_count=0,
                                max_window_length=(1 << 62), score_fraction=0.9)


def seed_long_reads(eng, mindex, reads, read_off, k, policy=None, threads=0):
    """MinimizerMapper::find_minimizers + find_seeds for reads of any length (src/minimizer_mapper.cpp:3918-4440): vgk_minimizer_list, the host shim's
    select_minimizers over every read's list (every filter, max_unique_min / num_bp_per_min included: :4162, :4312-4320), vgk_minimizer_seeds_of.
    -> dict(minimizer_off, minimizers, take, seed_off (per minimizer), seeds, seeds_per_read)"""
    P = dict(GIRAFFE_LONG_READ_POLICY, **(policy or {}))
    reads = np.ascontiguousarray(reads, dtype=np.uint8); read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    moff, recs = eng.minimizer_list(mindex, reads, read_off)
    h = _host_lib()
    h.vgh_select_minimizers_of_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
    pol = np.array([P["hit_cap"], P["hard_hit_cap"], P["max_unique_min"], P["num_bp_per_min"], int(P["exclude_overlapping_min"]), P["coverage_flank"], P["window_count"],
                    P["max_window_length"]], dtype=np.uint64)
    take = np.zeros(max(len(recs), 1), dtype=np.uint8)
    recs = np.ascontiguousarray(recs)
    if h.vgh_select_minimizers_of_reads(recs.ctypes.data if len(recs) else None, moff.ctypes.data, len(read_off) - 1, reads.ctypes.data, read_off.ctypes.data, k, pol.ctypes.data,
                                        float(P["score_fraction"]), threads, take.ctypes.data) != 0:
        raise RuntimeError("vgh_select_minimizers_of_reads: " + (h.vgh_last_error() or b"?").decode())
    take = take[:len(recs)]
    soff, seeds = eng.minimizer_seeds_of(mindex, recs, take)
    per_read = soff[moff[1:].astype(np.int64)] - soff[moff[:-1].astype(np.int64)]
    return dict(minimizer_off=moff, minimizers=recs, take=take, seed_off=soff, seeds=seeds, seeds_per_read=per_read)


# ---- configs[4] with the whole stage in the host shim (vg_amd/host/chain_stage.cpp) -----------------------------------------------------
class ChainStage:
    """MinimizerMapper's chain alignment for a batch of reads, in C++ behind one call (vgh_chain_stage): every link through WFAExtender;
    what it declines through align_sequence_between_consistently — the local graph between / beyond the anchors cut out of the haplotype
    graph (extract_connecting_graph / extract_extending_graph), strands split, dagified, tips trimmed, then BandedGlobalAligner or pinned
    X-drop — all inside the call.  Owns a host-shim aligner (its own engine context), the haplotype graph and the WFA extender."""

    def __init__(self, wl, lib=None, scores=(1, 4, 6, 1, 5), device=0):
        h = _host_lib()
        h.vgh_graph_create.restype = ctypes.c_void_p
        h.vgh_graph_destroy.argtypes = [ctypes.c_void_p]
        h.vgh_graph_add_node.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p]
        h.vgh_graph_add_edge.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        h.vgh_aligner_create.restype = ctypes.c_void_p
        h.vgh_aligner_create.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 6
        h.vgh_aligner_destroy.argtypes = [ctypes.c_void_p]
        h.vgh_wfa_create.restype = ctypes.c_void_p
        h.vgh_wfa_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        h.vgh_wfa_destroy.argtypes = [ctypes.c_void_p]
        h.vgh_wfa_set_point_budgets.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        h.vgh_wfa_last_kernel_ms.restype = ctypes.c_double; h.vgh_wfa_last_kernel_ms.argtypes = [ctypes.c_void_p]
        h.vgh_wfa_last_wave.restype = ctypes.c_double; h.vgh_wfa_last_wave.argtypes = [ctypes.c_void_p, ctypes.c_int]
        h.vgh_chain_stage.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 6 + [ctypes.c_uint32] + [ctypes.c_void_p] * 4 + \
                                     [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_void_p] * 6
        h.vgh_chain_stage_view.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 4 + [ctypes.POINTER(ctypes.c_double)]
        self.h = h
        self.aligner = h.vgh_aligner_create(lib.encode() if lib else None, device, *scores)
        if not self.aligner:
            raise RuntimeError("vgh_aligner_create: " + (h.vgh_last_error() or b"?").decode())
        g = h.vgh_graph_create()                                        # node v of the workload is node id v + 1: oriented node 2 v + strand in the index
        big = getattr(wl, "graph", None)
        if big is not None:                                               # a VariationGraph's arrays, in two calls
            h.vgh_graph_add_nodes.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            h.vgh_graph_add_edges.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
            ids = np.arange(1, big.n_nodes + 1, dtype=np.int64); so = np.ascontiguousarray(big.col, dtype=np.uint64); sq = np.ascontiguousarray(big.seq)
            if h.vgh_graph_add_nodes(g, big.n_nodes, ids.ctypes.data, sq.ctypes.data, so.ctypes.data):
                raise RuntimeError("vgh_graph_add_nodes: " + (h.vgh_last_error() or b"?").decode())
            to = np.repeat(ids, np.diff(big.pred_off.astype(np.int64))); fr = big.pred_idx.astype(np.int64) + 1
            if h.vgh_graph_add_edges(g, len(to), np.ascontiguousarray(fr).ctypes.data, np.ascontiguousarray(to).ctypes.data):
                raise RuntimeError("vgh_graph_add_edges: " + (h.vgh_last_error() or b"?").decode())
        else:
            for v, s in enumerate(wl.nodes):
                h.vgh_graph_add_node(g, v + 1, s.encode())
            for v, preds in enumerate(wl.preds):
                for p in preds:
                    h.vgh_graph_add_edge(g, p + 1, v + 1)
        flat = np.concatenate([np.asarray(t, dtype=np.int64) for t in wl.threads])
        tn = ((flat >> 1) + 1) * 2 + (flat & 1)                           # handles: (id << 1) | strand
        toff = np.concatenate([[0], np.cumsum([len(t) for t in wl.threads])]).astype(np.int32)
        self.wfa = h.vgh_wfa_create(self.aligner, g, np.ascontiguousarray(tn).ctypes.data, toff.ctypes.data, len(wl.threads), None)
        h.vgh_graph_destroy(g)                                            # (the haplotype graph holds its own copy)
        if not self.wfa:
            raise RuntimeError("vgh_wfa_create: " + (h.vgh_last_error() or b"?").decode())
        a = wl.ws.array
        self.n = wl.n; self.n_reads = wl.n_reads
        self.seqs = np.ascontiguousarray(wl.ws.seqs); self.seq_off = np.ascontiguousarray(wl.ws.seq_off, dtype=np.uint64)
        self.fields = [np.ascontiguousarray(a[k], dtype=np.uint32) for k in ("mode", "from_node", "from_offset", "to_node", "to_offset")]
        self.read_of = np.ascontiguousarray(wl.read_of, dtype=np.uint32)
        self.graph_distance = np.ascontiguousarray(wl.span, dtype=np.uint32)
        self.read_begin = np.ascontiguousarray(wl.link_begin, dtype=np.uint32); self.read_length = np.ascontiguousarray(wl.link_read_length, dtype=np.uint32)
        self.anchor_score = np.ascontiguousarray(wl.anchor_bases, dtype=np.int64)
        h.vgh_wfa_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
        self.registered = h.vgh_wfa_host_register(self.wfa, self.seqs.ctypes.data, self.seqs.nbytes, 1) == 0 and self.seqs.nbytes > 0      # (the links' sequences go up by DMA from here)
        self.anchors = [np.ascontiguousarray(getattr(wl, k)) for k in ("anchor_off", "anchor_length", "anchor_node_offset", "anchor_path_off", "anchor_nodes")] if hasattr(wl, "anchor_off") else None

    def set_point_budgets(self, connect, tail):
        self.h.vgh_wfa_set_point_budgets(self.wfa, connect, tail)

    def run(self, threads=0, dp_for_tails=True, timing=None, compose=False):
        """compose: also ONE alignment per read (find_chain_alignment's composed_path, simplified) — `alignments` = (results CHAIN_RESULT_DT, mappings
        CHAIN_MAPPING_DT, edits uint32) as views of the stage's own arrays, valid until the next run — and `broken` (per read)"""
        link_score = np.zeros(self.n, dtype=np.int32); source = np.zeros(self.n, dtype=np.uint8); status = np.zeros(self.n, dtype=np.int32)
        chain = np.zeros(self.n_reads, dtype=np.int64); stats = np.zeros(6, dtype=np.uint64); ms = np.zeros(6, dtype=np.float64); sizes = np.zeros(2, dtype=np.uint64)
        if compose and self.anchors is None:
            raise ValueError("ChainStage.run(compose=True): the workload states no anchors")
        anchors = [a.ctypes.data for a in self.anchors] if compose else [None] * 5
        rc = self.h.vgh_chain_stage(self.wfa, self.seqs.ctypes.data, self.seq_off.ctypes.data, self.n, *[f.ctypes.data for f in self.fields], self.read_of.ctypes.data,
                                    self.n_reads, self.graph_distance.ctypes.data, self.read_begin.ctypes.data, self.read_length.ctypes.data, self.anchor_score.ctypes.data,
                                    threads, int(dp_for_tails), link_score.ctypes.data, source.ctypes.data, status.ctypes.data, chain.ctypes.data, stats.ctypes.data, ms.ctypes.data,
                                    *anchors, sizes.ctypes.data)
        if rc:
            raise RuntimeError("vgh_chain_stage: " + (self.h.vgh_last_error() or b"?").decode())
        if timing is not None:
            for k, v in zip(("wfa_extend (all links)", "requests for the declined links", "local graphs: extract + split + dagify + trim (host threads)",
                             "banded + X-drop flush", "translation + totals", "one alignment per read: pieces + vgk_chain_stitch"), ms):
                timing[k] = timing.get(k, 0.0) + v * 1e-3
        out = dict(link_score=link_score, link_source=source, wfa_status=status, chain_score=chain,
                   stats=dict(zip(("declined", "between", "no_graph", "too_big", "failed", "broken_reads"), (int(x) for x in stats))), wfa_kernel_ms=self.h.vgh_wfa_last_kernel_ms(self.wfa),
                   wfa_launches=dict(small_tables_ms=self.h.vgh_wfa_last_wave(self.wfa, 0), large_tables_ms=self.h.vgh_wfa_last_wave(self.wfa, 1), problems_in_the_large_launch=int(self.h.vgh_wfa_last_wave(self.wfa, 2))))
        if compose:
            from . import capi
            ptr = [ctypes.c_void_p() for _ in range(4)]; kms = ctypes.c_double()
            if self.h.vgh_chain_stage_view(self.wfa, *[ctypes.byref(p) for p in ptr], ctypes.byref(kms)):
                raise RuntimeError("vgh_chain_stage_view: " + (self.h.vgh_last_error() or b"?").decode())

            def view(p, count, dt):
                if not count:
                    return np.zeros(0, dtype=dt)
                return np.frombuffer((ctypes.c_char * (count * np.dtype(dt).itemsize)).from_address(p.value), dtype=dt)
            out["alignments"] = (view(ptr[0], self.n_reads, capi.CHAIN_RESULT_DT), view(ptr[1], int(sizes[0]), capi.CHAIN_MAPPING_DT), view(ptr[2], int(sizes[1]), np.uint32))
            out["broken"] = view(ptr[3], self.n_reads, np.uint8); out["stitch_kernel_ms"] = kms.value
        return out

    def close(self):
        if getattr(self, "wfa", None):
            if getattr(self, "registered", False):
                self.h.vgh_wfa_host_register(self.wfa, self.seqs.ctypes.data, self.seqs.nbytes, 0); self.registered = False
            self.h.vgh_wfa_destroy(self.wfa); self.wfa = None
        if getattr(self, "aligner", None):
            self.h.vgh_aligner_destroy(self.aligner); self.aligner = None

    def __del__(self):
        self.close()


# ---- a paired-end slice: both mates through the stage, the mate without a full-length extension rescued from the other's position ------------
def paired_stage(eng, index, mindex, wl, host_aligner, oriented_len=None, device=True, rescue_stdevs=4.0, timing=None, host_threads=0, resident=None, want_ops=False,
                 request_table=None):
    """giraffe's paired-end shape on one batch of pairs (PairedWorkload): (1) every read through seeding, gapless extension and the tails
    (align_stage_device, or align_stage over the oracle when device = False); (2) for a pair with exactly one mate whose extension set is
    full-length, the other mate is RESCUED: the nodes at the fragment's distance from the mapped mate — here a run of nodes found by
    column coordinate, a stated stand-in for subgraph_in_distance_range over the absent SnarlDistanceIndex —, the best of its own
    extensions inside them as dozeu's seed, Aligner::align_xdrop + fix_dozeu_score + fix_dozeu_end_deletions for all such mates at
    once (vg_amd/host/rescue_stage.cpp = MinimizerMapper::attempt_rescue, src/minimizer_mapper.cpp:3264-3440); (3) a pair's score = the
    mapped mate's + the better of the rescued alignment and what the stage had for that mate.
    resident: a RescueGraphHandle (host_aligner.rescue_graph(wl)) — the rescue half then runs on the RESIDENT graph (vg_amd/host/rescue_resident.cpp:
    every X-drop pass an extension window whose sub-DAG the device derives, the fix-ups over flat arrays); without it the reference-shaped path
    (rescue_stage.cpp: one HashGraph and one Alignment per mate) — the form the checker runs over the oracle.  want_ops: also the rescued
    alignments as (node, op, length) runs (`rescue_ops`, `rescue_ops_begin`).  request_table (resident route): "device" — vgk_rescue_requests over
    the sets the stage left in HBM (the default when device = True) — or "host": vg_amd/host/rescue_requests.cpp over the fetched sets, the
    statement the tests hold the device's table against.
    -> dict(read_score, rescued (indices of rescued reads), rescue (RESCUE_DT-like int64 [k, 6]), pair_score)"""
    import time
    t0 = time.perf_counter()
    n = wl.n
    if device:
        class _B:
            pass
        b = _B(); b.n = n
        seed_off, _, _ = eng.minimizer_seeds(mindex, index, wl.reads, wl.read_off, keep_on_device=True)
        out = align_stage_device(eng, index, b, seeded=int(seed_off[-1]), aligned=False)
    else:
        so, sd, _ = eng.minimizer_seeds(mindex, index, wl.reads, wl.read_off)
        sub = capi.GaplessSet(wl.reads, wl.read_off, sd, so, node_cap=len(sd) * 16, mism_cap=len(sd) * 12)
        out = align_stage(eng, index, oriented_len, sub)
    t1 = time.perf_counter()
    res, ext, nodes = out["res"], out["ext"], out["nodes"]
    read_score = np.asarray(out["read_score"], dtype=np.int64)
    L = wl.read_len
    g = wl.graph
    if resident is not None:
        # the product route: the request table from the extension sets on chunked host threads (vg_amd/host/rescue_requests.cpp), no numpy in between
        return _paired_stage_resident(eng, wl, host_aligner, resident, out, read_score, rescue_stdevs, timing, host_threads, want_ops, t0, t1,
                                      table_on_device=device if request_table is None else request_table == "device")
    full = (res["status"] == 0) & (res["full_length"] != 0)
    a_full, b_full = full[0::2], full[1::2]
    pairs = np.nonzero(a_full != b_full)[0]
    mapped = np.where(a_full[pairs], 2 * pairs, 2 * pairs + 1); lost = mapped ^ 1
    # where the mapped mate lies: the first node of its first (full-length) extension, on the strand the read reads forward on
    e0 = res["ext_begin"][mapped].astype(np.int64)
    first = nodes[ext["path_begin"][e0].astype(np.int64)].astype(np.int64)
    fwd = (first & 1) == 0
    col = g.col
    # forward-mapped mate starting at column s: its partner lies downstream on the other strand, within [s + mean - k sd - L, s + mean + k sd];
    # reverse-mapped mate ending at column e (the forward end of its last path node): upstream on the forward strand
    lo_d = max(0.0, wl.mean - rescue_stdevs * wl.sd - L); hi_d = (wl.mean + rescue_stdevs * wl.sd) * 1.1 + 40
    s_col = col[first >> 1] + ext["offset"][e0]
    e_col = col[(first >> 1) + 1] - ext["offset"][e0]                  # reverse strand: the read's first base is the node's last minus the offset
    c_lo = np.where(fwd, s_col + lo_d, e_col - hi_d); c_hi = np.where(fwd, s_col + hi_d, e_col - lo_d)
    node_lo = np.clip(np.searchsorted(col, np.maximum(c_lo, 0), side="right") - 1, 0, g.n_nodes - 1)
    node_hi = np.clip(np.searchsorted(col, np.minimum(c_hi, col[-1] - 1), side="right"), 1, g.n_nodes)
    # the mate to rescue as it reads along the FORWARD strand of that subgraph: as sequenced when its partner is on the reverse strand
    rescue_rc = fwd
    rd = wl.reads.reshape(-1, L)[lost]
    rd = np.where(rescue_rc[:, None], _COMP[rd[:, ::-1]], rd)
    # dozeu's seed: the best extension of the lost mate inside the subgraph, on the strand it is rescued on
    req = np.zeros((len(pairs), 6), dtype=np.int64); req[:, 0] = node_lo; req[:, 1] = node_hi; req[:, 4] = -1
    olen = np.repeat(g.node_len.astype(np.int64), 2)
    eb = res["ext_begin"][lost].astype(np.int64); ne = np.where(res["status"][lost] == 0, res["n_ext"][lost].astype(np.int64), 0)
    tot = int(ne.sum())
    if tot:
        # every extension of every lost mate at once: which of them lie inside the mate's subgraph on the strand it is rescued on, the best per mate
        owner = np.repeat(np.arange(len(pairs)), ne)
        row = eb[owner] + (np.arange(tot) - np.repeat(np.cumsum(ne) - ne, ne))
        pb = ext["path_begin"][row].astype(np.int64); pl = ext["path_len"][row].astype(np.int64)
        keep = pl > 0
        owner, row, pb, pl = owner[keep], row[keep], pb[keep], pl[keep]
        if len(row):
            seg = np.cumsum(pl) - pl
            idx = np.repeat(pb, pl) + (np.arange(int(pl.sum())) - np.repeat(seg, pl))
            pv = nodes[idx].astype(np.int64) >> 1
            pmin = np.minimum.reduceat(pv, seg); pmax = np.maximum.reduceat(pv, seg)
            first_o = nodes[pb].astype(np.int64); last_o = nodes[pb + pl - 1].astype(np.int64)
            ok = (((first_o & 1) == 1) == rescue_rc[owner]) & (pmin >= node_lo[owner]) & (pmax < node_hi[owner])
            owner, row, pb, pl, first_o, last_o = owner[ok], row[ok], pb[ok], pl[ok], first_o[ok], last_o[ok]
            if len(row):
                order = np.lexsort((row, -ext["score"][row].astype(np.int64), owner))       # per mate: best score, the earlier extension among equals
                take = order[np.concatenate([[True], owner[order][1:] != owner[order][:-1]])]
                k = owner[take]; x = row[take]
                rb = ext["read_begin"][x].astype(np.int64); re_ = ext["read_end"][x].astype(np.int64); off = ext["offset"][x].astype(np.int64)
                cs = np.concatenate([[0], np.cumsum(olen[nodes.astype(np.int64)])])
                path_bases = cs[pb[take] + pl[take]] - cs[pb[take]]
                lastlen = olen[last_o[take]]
                # seen from the forward strand (a mate rescued as its reverse complement): the path backwards, the read interval mirrored, the
                # offset counted from the last node's other end
                end_in_last = np.where(pl[take] > 1, (re_ - rb) - (path_bases - off - lastlen), off + (re_ - rb))
                rc = rescue_rc[k]
                req[k, 2] = np.where(rc, L - re_, rb); req[k, 3] = np.where(rc, L - rb, re_)
                req[k, 4] = np.where(rc, last_o[take] >> 1, first_o[take] >> 1); req[k, 5] = np.where(rc, lastlen - end_in_last, off)
    t2 = time.perf_counter()
    outv = np.zeros((len(pairs), 6), dtype=np.int64)
    ops_begin = np.zeros(len(pairs) + 1, dtype=np.uint64); ops = np.zeros(0, dtype=capi.OP_DT)
    laps = np.zeros(6, dtype=np.float64); counts = np.zeros(6, dtype=np.uint64)
    if len(pairs):
        h = _host_lib()
        flat = np.ascontiguousarray(rd).ravel(); roff = (np.arange(len(pairs) + 1, dtype=np.uint64) * L)
        # (the resident route returned above: this is the per-graph form — one HashGraph per mate on host threads)
        node_len = np.ascontiguousarray(g.node_len, dtype=np.uint32); seq_off = np.ascontiguousarray(g.col[:-1], dtype=np.uint64); seq = np.ascontiguousarray(g.seq)
        args = [host_aligner.ptr, g.n_nodes, node_len.ctypes.data, seq_off.ctypes.data, seq.ctypes.data, wl.succ_off.ctypes.data, wl.succ.ctypes.data,
                len(pairs), flat.ctypes.data, roff.ctypes.data, req.ctypes.data, 0, host_threads, outv.ctypes.data]
        base_types = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        ops_cap = len(pairs) * 64 if want_ops else 0
        written = ctypes.c_uint64()
        for attempt in range(2):                       # -2: the op runs need more room than the first guess; `written` says how much (as Engine.rescue_requests retries on VGK_EOPS)
            ops = np.zeros(max(ops_cap, 1), dtype=capi.OP_DT)
            if want_ops:
                h.vgh_rescue_stage_ops.argtypes = base_types + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
                rc = h.vgh_rescue_stage_ops(*args, ops_begin.ctypes.data, ops.ctypes.data, ops_cap, ctypes.byref(written))
            else:
                h.vgh_rescue_stage.argtypes = base_types
                rc = h.vgh_rescue_stage(*args)
            if rc != -2 or not want_ops or int(written.value) <= ops_cap:
                break
            ops_cap = int(written.value)
        if rc != 0:
            raise RuntimeError(h.vgh_last_error().decode())
        ops = ops[:int(written.value)] if want_ops else ops[:0]
    t3 = time.perf_counter()
    pair_score = read_score[0::2] + read_score[1::2]
    pair_score[pairs] = read_score[mapped] + np.maximum(outv[:, 0], read_score[lost])
    if timing is not None:
        for k, v in (("stage (seeding, extension, tails)", t1 - t0), ("rescue requests (host)", t2 - t1), ("rescue stage (subgraphs, X-drop passes, fix-ups)", t3 - t2)):
            timing[k] = timing.get(k, 0.0) + v
    return dict(read_score=read_score, rescued=lost, mapped=mapped, requests=req, rescue=outv, pair_score=pair_score, res=res, rescue_ops=ops, rescue_ops_begin=ops_begin,
                rescue_counts=dict(zip(("first_pass", "scans", "second_pass", "fallbacks", "alg_bytes", "cells"), (int(x) for x in counts)), kernel_ms=float(laps[5])))


def _paired_stage_resident(eng, wl, host_aligner, resident, out, read_score, rescue_stdevs, timing, host_threads, want_ops, t0, t1, table_on_device=False):
    import time
    h = _host_lib()
    res, ext, nodes = np.ascontiguousarray(out["res"]), np.ascontiguousarray(out["ext"]), np.ascontiguousarray(out["nodes"], dtype=np.uint32)
    L = wl.read_len; g = wl.graph; n_pairs = wl.n // 2
    bufs = getattr(resident, "_bufs", None)
    if bufs is None or bufs[0] < n_pairs:                   # a streaming caller keeps its request / output arrays
        bufs = resident._bufs = (n_pairs, np.zeros(n_pairs, dtype=np.uint32), np.zeros(n_pairs, dtype=np.uint32), np.zeros((n_pairs, 6), dtype=np.int64),
                                 np.zeros(n_pairs * L, dtype=np.uint8), np.zeros((n_pairs, 6), dtype=np.int64), np.arange(n_pairs + 1, dtype=np.uint64) * L,
                                 np.ascontiguousarray(g.col, dtype=np.int64))
    _, mapped, lost, req, rd, outv, roff, col = bufs
    if table_on_device:
        # the table from the sets where the extension kernels left them (one lane per pair); the host adds the lost mates' own bytes
        tab = eng.rescue_requests(resident.dgraph, wl.mean, wl.sd, rescue_stdevs, out=getattr(resident, "_table", None))
        resident._table = eng._rescue_requests_buf
        m = len(tab)
        h.vgh_rescue_reads.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int] + [ctypes.c_void_p] * 4
        if m and h.vgh_rescue_reads(m, tab.ctypes.data, wl.reads.ctypes.data, L, host_threads, mapped.ctypes.data, lost.ctypes.data, req.ctypes.data, rd.ctypes.data) != 0:
            raise RuntimeError(h.vgh_last_error().decode())
        return _paired_rescue_half(eng, wl, host_aligner, resident, read_score, timing, host_threads, want_ops, t0, t1, res, m, "rescue requests (device table + the mates' reads)")
    h.vgh_rescue_requests.restype = ctypes.c_int64
    h.vgh_rescue_requests.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                      ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 4
    m = h.vgh_rescue_requests(n_pairs, res.ctypes.data, ext.ctypes.data, nodes.ctypes.data, g.n_nodes, col.ctypes.data, wl.reads.ctypes.data, L, float(wl.mean), float(wl.sd),
                              float(rescue_stdevs), host_threads, mapped.ctypes.data, lost.ctypes.data, req.ctypes.data, rd.ctypes.data)
    if m < 0:
        raise RuntimeError(h.vgh_last_error().decode())
    return _paired_rescue_half(eng, wl, host_aligner, resident, read_score, timing, host_threads, want_ops, t0, t1, res, m, "rescue requests (host threads)")


def _paired_rescue_half(eng, wl, host_aligner, resident, read_score, timing, host_threads, want_ops, t0, t1, res, m, table_label):
    """the m requests in the resident graph's buffers through vgh_rescue_stage_resident, and the pairs' scores"""
    import time
    h = _host_lib()
    L = wl.read_len
    _, mapped, lost, req, rd, outv, roff, col = resident._bufs
    t2 = time.perf_counter()
    ops_begin = np.zeros(m + 1, dtype=np.uint64); ops_cap = m * 64 if want_ops else 0
    ops = np.zeros(max(ops_cap, 1), dtype=capi.OP_DT); written = ctypes.c_uint64()
    laps = np.zeros(6, dtype=np.float64); counts = np.zeros(6, dtype=np.uint64)
    if m:
        h.vgh_rescue_stage_resident.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        rc = h.vgh_rescue_stage_resident(host_aligner.ptr, resident.ptr, m, rd.ctypes.data, m * L, roff.ctypes.data, req.ctypes.data, 0, host_threads, outv.ctypes.data,
                                         ops_begin.ctypes.data if want_ops else None, ops.ctypes.data if want_ops else None, ops_cap, ctypes.byref(written), laps.ctypes.data, counts.ctypes.data)
        if rc != 0:
            raise RuntimeError(h.vgh_last_error().decode())
    t3 = time.perf_counter()
    mapped_i = mapped[:m].astype(np.int64); lost_i = lost[:m].astype(np.int64)
    pair_score = read_score[0::2] + read_score[1::2]
    pair_score[mapped_i >> 1] = read_score[mapped_i] + np.maximum(outv[:m, 0], read_score[lost_i])
    if timing is not None:
        for k, v in (("stage (seeding, extension, tails)", t1 - t0), (table_label, t2 - t1), ("rescue stage (resident graph: extension windows, fix-ups)", t3 - t2)):
            timing[k] = timing.get(k, 0.0) + v
        for k, v in zip(("rescue: classify (host)", "rescue: first pass (extension windows + scans)", "rescue: second pass (traced extension windows)", "rescue: alignments + fix-ups (host)", "rescue: full-DP fallback"), laps):
            timing[k] = timing.get(k, 0.0) + v * 1e-3
    return dict(read_score=read_score, rescued=lost_i, mapped=mapped_i, requests=req[:m].copy(), rescue=outv[:m].copy(), pair_score=pair_score, res=res,
                rescue_ops=ops[:int(written.value)] if want_ops else ops[:0], rescue_ops_begin=ops_begin,
                rescue_counts=dict(zip(("first_pass", "scans", "second_pass", "fallbacks", "alg_bytes", "cells"), (int(x) for x in counts)), kernel_ms=float(laps[5])))


class HostAlignerHandle:
    """a vgamd::Aligner of the host shim bound to an engine library (None: the HIP product library) — what paired_stage's rescue half runs on"""

    def __init__(self, lib=None, device=0, scores=(1, 4, 6, 1, 5)):
        h = _host_lib()
        h.vgh_aligner_create.restype = ctypes.c_void_p; h.vgh_aligner_create.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 6
        h.vgh_aligner_destroy.argtypes = [ctypes.c_void_p]
        self.h = h
        self.ptr = h.vgh_aligner_create(lib.encode() if lib else None, device, *scores)
        if not self.ptr:
            raise RuntimeError("vgh_aligner_create: " + (h.vgh_last_error() or b"?").decode())

    def rescue_graph(self, wl):
        """the workload's graph resident in this aligner's engine context, for paired_stage(resident = ...)"""
        return RescueGraphHandle(self, wl)

    def close(self):
        if getattr(self, "ptr", None):
            self.h.vgh_aligner_destroy(self.ptr); self.ptr = None

    __del__ = close


class RescueGraphHandle:
    """vgh_rescue_graph: the graph of a paired workload in the host aligner's engine context (vgk_graph_create) — rescue_resident.hpp"""

    def __init__(self, aligner, wl):
        g = wl.graph
        h = aligner.h
        h.vgh_rescue_graph_create.restype = ctypes.c_void_p
        h.vgh_rescue_graph_create.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        h.vgh_rescue_graph_destroy.argtypes = [ctypes.c_void_p]
        node_len = np.ascontiguousarray(g.node_len, dtype=np.uint32); seq = np.ascontiguousarray(g.seq)
        pred_off = np.ascontiguousarray(g.pred_off, dtype=np.uint32); pred_idx = np.ascontiguousarray(g.pred_idx, dtype=np.uint32)
        # predecessor lists must ascend (the order edges enter a subgraph built node by node, which the reference-shaped path does)
        po = pred_off.astype(np.int64); inner = np.ones(len(pred_idx), dtype=bool); inner[po[:-1][np.diff(po) > 0]] = False
        assert (np.diff(pred_idx.astype(np.int64))[inner[1:]] > 0).all() if len(pred_idx) > 1 else True, "predecessor lists must be ascending"
        self.aligner = aligner
        self.ptr = h.vgh_rescue_graph_create(aligner.ptr, g.n_nodes, node_len.ctypes.data, seq.ctypes.data, pred_off.ctypes.data, pred_idx.ctypes.data)
        if not self.ptr:
            raise RuntimeError("vgh_rescue_graph_create: " + (h.vgh_last_error() or b"?").decode())
        h.vgh_rescue_graph_dgraph.restype = ctypes.c_void_p; h.vgh_rescue_graph_dgraph.argtypes = [ctypes.c_void_p]
        self.dgraph = int(h.vgh_rescue_graph_dgraph(self.ptr) or 0)          # the vgk_dgraph: what Engine.rescue_requests takes

    def close(self):
        if getattr(self, "ptr", None):
            if getattr(self.aligner, "ptr", None):
                self.aligner.h.vgh_rescue_graph_destroy(self.ptr)
            self.ptr = None

    __del__ = close
