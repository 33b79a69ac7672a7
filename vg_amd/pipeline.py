"""giraffe's alignment stage over the engine's entry points, batch-wise: seeds -> gapless extension (vgk_gapless_extend) -> for the
clusters that no full-length extension resolves, the tails of every extension: tail forests (vgk_tail_forest), every tree a window
of the forest graph (vgk_gssw_pack_windows), left-pinned X-drop -> the extension's total score (src/minimizer_mapper.cpp:5480-5535).

Host-side glue only (numpy, vectorised): which tails exist, where they start, what their bases are.  It is what a maintainer's patch
of MinimizerMapper::extension_to_alignment's caller would do per batch, and it is what bench.py --workload giraffe times."""
import numpy as np

from . import capi, workloads

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def tails_of_extensions(oriented_len, read_off, res, ext, nodes, scoring=(1, 6, 1, 5)):
    """-> dict of numpy arrays, one entry per TAIL (an extension end that does not reach its read end, of a read whose extension set
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
    """the tails' bases, flat: a right tail as it is, a left tail reverse-complemented (:5660) -> (bases, offsets)"""
    ln = t["end"] - t["begin"]
    off = np.concatenate([[0], np.cumsum(ln)])
    which = np.repeat(np.arange(len(ln)), ln)
    k = np.arange(off[-1]) - off[which]
    base = read_off[t["read"]][which]
    pos = np.where(t["left"][which], t["end"][which] - 1 - k, t["begin"][which] + k)
    b = reads[base + pos]
    return np.where(t["left"][which], _COMP[b], b), off


def tree_windows(res, forest, seq, seq_off, gap):
    """one left-pinned X-drop window problem per tree -> (WindowSet, tail index of every tree)"""
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
    ln = np.diff(seq_off)[owner]
    off = np.concatenate([[0], np.cumsum(ln)])
    which = np.repeat(np.arange(len(owner)), ln)
    bases = seq[seq_off[owner][which] + (np.arange(off[-1]) - off[which])]
    return capi.WindowSet(bases, off, first, count, flags, gap[owner], cols=cols), owner


def align_stage(eng, index, oriented_len, gs, ops_per_problem=32):
    """The whole stage for one batch of clusters (a GaplessSet).  -> dict: the extension outputs, the tails, per tail its best tree's
    score, per extension its total score, per read the best total; `forest`, `batch` results for parity checks."""
    res, ext, nodes, mism = eng.gapless_extend(index, gs)
    t = tails_of_extensions(oriented_len, gs.read_off, res, ext, nodes)
    out = dict(res=res, ext=ext, nodes=nodes, mism=mism, tails=t)
    total = ext["score"].astype(np.int64).copy()
    if len(t["problems"]):
        tres, forest = eng.tail_forest(index, t["problems"])
        seq, seq_off = tail_sequences(gs.reads, gs.read_off, t)
        ws, owner = tree_windows(tres, forest, seq, seq_off, t["gap"])
        r, ops = eng.align_windows(forest.graph, ws, ops_per_problem)
        best = np.zeros(len(t["problems"]), dtype=np.int64)                       # a tail nothing aligns to is a soft clip: 0 (:5632-5648)
        np.maximum.at(best, owner, np.where(r["status"] == 0, r["score"], 0))
        np.add.at(total, t["ext"], best)
        out.update(tail_results=tres, forest=forest, windows=ws, owner=owner, tail_alignments=r, tail_ops=ops, tail_score=best)
    read_of_ext = np.repeat(np.arange(len(res)), res["n_ext"])
    best_read = np.zeros(len(res), dtype=np.int64)
    np.maximum.at(best_read, read_of_ext, total[:len(read_of_ext)])
    out.update(ext_total=total, read_score=best_read)
    return out
