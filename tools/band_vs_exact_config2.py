#!/usr/bin/env python3
"""dozeu's band against the un-pruned X-drop on the tails of configs[2]: every tail tree of a batch of reads as an
explicit problem through vgk_xdrop_band_align and through vgk_gssw_align (VGK_XDROP_PINNED, every cell kept) — how many tails' answers
(score, end cell, CIGAR) differ.  Builder-run on the GPU box: python tools/band_vs_exact_config2.py --batches 4 [--reads 1000000]
[--ref-len N]; prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vg_amd import capi, pipeline, workloads          # noqa: E402

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def oriented_sequences(node_len, seq):
    """forward strand of node v at 2v, its reverse complement at 2v + 1 -> (bases, offsets per oriented node)"""
    node_len = np.asarray(node_len, dtype=np.int64); n = len(node_len)
    fwd_off = np.concatenate([[0], np.cumsum(node_len)])
    olen = np.repeat(node_len, 2)
    ooff = np.concatenate([[0], np.cumsum(olen)])
    out = np.empty(int(ooff[-1]), dtype=np.uint8)
    # position p of the forward strand of v -> ooff[2v] + p ; of the reverse strand -> ooff[2v + 1] + (len - 1 - p), complemented
    owner = np.repeat(np.arange(n), node_len); within = np.arange(int(fwd_off[-1])) - fwd_off[owner]
    out[ooff[2 * owner] + within] = seq
    out[ooff[2 * owner + 1] + (node_len[owner] - 1 - within)] = _COMP[seq]
    return out, ooff, olen


def explicit_problems(ws, forest, obases, ooff, olen):
    """the windows of a tail forest (one tree each) as explicit graphs: a tree node's bases are the last `length` bases of its oriented node"""
    parent, node, length = forest.fetch()
    arr = ws.array
    first = arr["first_node"].astype(np.int64); count = arr["n_nodes"].astype(np.int64)
    node_off = np.concatenate([[0], np.cumsum(count)])
    total = int(node_off[-1])
    owner = np.repeat(np.arange(ws.n), count)
    idx = first[owner] + (np.arange(total) - node_off[owner])               # forest node of every problem node
    ln = length[idx].astype(np.int64)
    on = node[idx].astype(np.int64)
    start = ooff[on] + (olen[on] - ln)
    seq_node_off = np.concatenate([[0], np.cumsum(ln)])
    inc = np.ones(int(seq_node_off[-1]), dtype=np.int64)
    nz = ln > 0
    last = start + ln - 1
    inc[seq_node_off[:-1][nz]] = start[nz] - np.concatenate([[0], last[nz][:-1]])
    seq = obases[np.cumsum(inc)]
    seq_off = seq_node_off[node_off]
    par = parent[idx].astype(np.int64)
    has = par >= first[owner]                                                # the tree's root has none (or one outside the window)
    pred_idx = (par - first[owner])[has].astype(np.uint32)
    edge_off = np.concatenate([[0], np.cumsum(has)])[node_off]
    # local CSR offsets: n_nodes + 1 entries per problem, laid out at node_off[i] + i
    pred_off = np.zeros(total + ws.n, dtype=np.uint32)
    csum = np.concatenate([[0], np.cumsum(has)])
    local_after = csum[1:] - csum[node_off[owner]]                           # edges of the problem up to and including this node
    pred_off[np.arange(total) + owner + 1] = local_after
    flags = np.full(ws.n, capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK, dtype=np.uint32)
    return capi.ProblemSet(ws.reads, ws.read_off, ln.astype(np.uint32), node_off, seq, seq_off, pred_off, pred_idx, edge_off, flags,
                           max_gap=arr["max_gap_length"])


def same_alignment(ra, oa, rb, ob):
    same = np.ones(len(ra), dtype=bool)
    for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
        same &= ra[f] == rb[f]
    idx = np.flatnonzero(same & (ra["n_ops"] > 0))
    if len(idx):
        cnt = ra["n_ops"][idx].astype(np.int64)
        owner = np.repeat(np.arange(len(idx)), cnt)
        within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        a = oa.view(np.uint64)[ra["ops_begin"][idx].astype(np.int64)[owner] + within]
        b = ob.view(np.uint64)[rb["ops_begin"][idx].astype(np.int64)[owner] + within]
        same[idx[np.unique(owner[a != b])]] = False
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=2); ap.add_argument("--reads", type=int, default=1_000_000); ap.add_argument("--ref-len", type=int, default=0)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=args.lib)
    wl = workloads.Config2Workload(args.reads * args.batches, batch=args.reads, seed=31, graph=workloads.VariationGraph(ref_len=args.ref_len) if args.ref_len else None)
    graph = (wl.node_len, wl.seq)
    index = eng.haplo_index(graph, wl.threads); mindex = eng.minimizer_index(graph, wl.threads)
    obases, ooff, olen = oriented_sequences(wl.node_len, wl.seq)
    tot = dict(tails=0, trees=0, differing=0, differing_same_score=0, band_lower=0, failed_band=0, failed_exact=0, cells_in_band=0, cells_rect=0)
    t_band = t_exact = 0.0
    for b in range(args.batches):
        reads, off = wl.batches[b]
        seed_off, seeds, _ = eng.minimizer_seeds(mindex, index, reads, off)
        gs = capi.GaplessSet(reads, off, seeds, seed_off)
        res, ext, nodes, mism = eng.gapless_extend(index, gs)
        t = pipeline.tails_of_extensions(olen, gs.read_off, res, ext, nodes)
        if not len(t["problems"]):
            continue
        tres, forest = eng.tail_forest(index, t["problems"])
        seq, seq_off = pipeline.tail_sequences(gs.reads, gs.read_off, t)
        ws, owner = pipeline.tree_windows(tres, forest, seq, seq_off, t["gap"])
        ps = explicit_problems(ws, forest, obases, ooff, olen)
        if b == 0:                                                           # the explicit problems ARE the windows: the stage's own answers on them
            with eng.pack_windows(forest.graph, ws, 32) as pb:
                pb.run(); pb.sync(); wr, wops = pb.fetch()
            chk, _ = eng.align(ps, 48)
            assert (wr["score"] == chk["score"]).all() and (wr["status"] == chk["status"]).all(), "explicit problems differ from the windows"
        t0 = time.perf_counter(); bres, bops, st = eng.xdrop_band_align(ps); t_band += time.perf_counter() - t0
        t0 = time.perf_counter(); eres, eops = eng.align(ps, 48); t_exact += time.perf_counter() - t0
        same = same_alignment(bres, bops, eres, eops)
        tot["tails"] += len(t["problems"]); tot["trees"] += ps.n
        tot["differing"] += int((~same).sum()); tot["differing_same_score"] += int((~same & (bres["score"] == eres["score"])).sum())
        tot["band_lower"] += int((bres["score"] < eres["score"]).sum())
        assert (bres["score"] <= eres["score"]).all()
        tot["failed_band"] += int((bres["status"] != 0).sum()); tot["failed_exact"] += int((eres["status"] != 0).sum())
        tot["cells_in_band"] += st[0]; tot["cells_rect"] += st[1]
        forest.close()
        print("batch %d: %d tails, %d trees, differing so far %d" % (b, len(t["problems"]), ps.n, tot["differing"]), file=sys.stderr)
    tot.update(batches=args.batches, reads_per_batch=args.reads, band_s=t_band, exact_s=t_exact,
               workload="configs[2]: %d bp reference, %d nodes; the tail trees of %d x %d reads as explicit problems" % (len(wl.graph.haps[0][0]) if hasattr(wl.graph, "haps") else 0, len(wl.node_len), args.batches, args.reads))
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
