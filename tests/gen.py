"""Seeded random (read, DAG) problem generator for differential tests.
Mirrors the spirit of the reference's src/unittest/support/random_graph.{hpp,cpp}."""
import numpy as np

from vg_amd import capi

BASES = "ACGT"


def random_dag(rng, n_nodes, max_len, p_chain=0.7, with_n=0.0):
    """Nodes come out in topological order; every non-source node has >= 1 predecessor."""
    nodes, preds = [], []
    for v in range(n_nodes):
        ln = int(rng.integers(1, max_len + 1))
        s = "".join(BASES[i] for i in rng.integers(0, 4, ln))
        if with_n and rng.random() < with_n:
            k = int(rng.integers(0, ln)); s = s[:k] + "N" + s[k + 1:]
        nodes.append(s)
        if v == 0:
            preds.append([])
            continue
        if rng.random() < p_chain:
            pr = [v - 1]
        else:
            cnt = int(rng.integers(0, min(v, 3) + 1))
            pr = sorted(set(int(x) for x in rng.integers(0, v, cnt)))
        if pr and rng.random() < 0.3:   # extra edge -> multi-predecessor node
            pr = sorted(set(pr + [int(rng.integers(0, v))]))
        if rng.random() < 0.5:
            pr = pr[::-1]               # predecessor order matters for tie-breaks
        preds.append(pr)
    return nodes, preds


def random_walk_read(rng, nodes, preds, length, sub=0.05, indel=0.02):
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for p in pr:
            succ[p].append(v)
    v = int(rng.integers(0, len(nodes)))
    off = int(rng.integers(0, len(nodes[v])))
    out = []
    while len(out) < length:
        if off >= len(nodes[v]):
            if not succ[v]:
                break
            v = succ[v][int(rng.integers(0, len(succ[v])))]; off = 0
            continue
        c = nodes[v][off]; off += 1
        r = rng.random()
        if r < sub:
            c = BASES[int(rng.integers(0, 4))]
        elif r < sub + indel / 2:
            continue                                  # deletion from the read
        elif r < sub + indel:
            out.append(BASES[int(rng.integers(0, 4))])  # insertion
        out.append(c)
    while len(out) < max(1, length // 3):
        out.append(BASES[int(rng.integers(0, 4))])
    return "".join(out[:length])


def random_problem(rng, max_nodes=10, max_node_len=12, max_read=120, mode=None, traceback=True, with_n=0.0):
    n_nodes = int(rng.integers(1, max_nodes + 1))
    nodes, preds = random_dag(rng, n_nodes, max_node_len, with_n=with_n)
    L = int(rng.integers(1, max_read + 1))
    kind = rng.random()
    if kind < 0.8:
        read = random_walk_read(rng, nodes, preds, L)
    else:
        read = "".join(BASES[i] for i in rng.integers(0, 4, L))
    if with_n and rng.random() < 0.3:
        k = int(rng.integers(0, len(read))); read = read[:k] + "N" + read[k + 1:]
    if mode is None:
        mode = capi.VGK_GSSW_PINNED if rng.random() < 0.4 else capi.VGK_GSSW_LOCAL
    flags = mode | (capi.VGK_GSSW_TRACEBACK if traceback else 0)
    pinning = None
    if mode == capi.VGK_XDROP_PINNED:
        # dozeu pinned extension: reads that start at a source node make the interesting cases
        if rng.random() < 0.7:
            src = [v for v, pr in enumerate(preds) if not pr]
            v = src[int(rng.integers(0, len(src)))]
            walk = random_walk_read(rng, nodes[v:v + 1] + nodes[v + 1:], [[]] + [[q - v for q in pr if q >= v] for pr in preds[v + 1:]], L)
            # restart the walk at offset 0 of the source: simplest is to take the source sequence then continue
            read = (nodes[v] + read)[:max(1, L)] if rng.random() < 0.5 else walk
        return {"read": read, "nodes": nodes, "preds": preds, "flags": flags, "pinning": None,
                "max_gap": int(rng.integers(0, 60))}
    if mode == capi.VGK_GSSW_PINNED:
        has_succ = [False] * n_nodes
        for v, pr in enumerate(preds):
            for p in pr:
                has_succ[p] = True
        pinning = [0 if h else 1 for h in has_succ]
    return {"read": read, "nodes": nodes, "preds": preds, "flags": flags, "pinning": pinning}


def problem_set(problems):
    return capi.ProblemSet.from_lists(problems)


def random_banded_problem(rng, max_nodes=8, max_node_len=8, max_read=40, p_empty=0.15, wide=False):
    """A banded-global problem: DAG that may hold empty nodes, a read sampled from a source-to-sink walk."""
    n_nodes = int(rng.integers(1, max_nodes + 1))
    nodes, preds = random_dag(rng, n_nodes, max_node_len)
    nodes = ["" if rng.random() < p_empty else s for s in nodes]
    succ = [[] for _ in nodes]
    for v, pr in enumerate(preds):
        for p in pr:
            succ[p].append(v)
    sources = [v for v, pr in enumerate(preds) if not pr]
    v = sources[int(rng.integers(0, len(sources)))]
    walk = []
    while True:
        walk.append(nodes[v])
        if not succ[v]:
            break
        v = succ[v][int(rng.integers(0, len(succ[v])))]
    ref = "".join(walk)
    out = []
    for c in ref:
        r = rng.random()
        if r < 0.08:
            out.append(BASES[int(rng.integers(0, 4))])
        elif r < 0.12:
            continue
        elif r < 0.16:
            out.append(BASES[int(rng.integers(0, 4))]); out.append(c)
        else:
            out.append(c)
    if rng.random() < 0.2 or not out:
        out = [BASES[i] for i in rng.integers(0, 4, int(rng.integers(1, max_read + 1)))]
    read = "".join(out[:max_read])
    return dict(read=read, nodes=nodes, preds=preds,
                band_padding=(len(read) + sum(len(s) for s in nodes) + 2) if wide else int(rng.integers(0, 6)),
                permissive=True if wide else bool(rng.random() < 0.7))
