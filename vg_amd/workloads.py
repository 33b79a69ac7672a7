"""Synthetic workloads for bench.py and the full-size tests.

config 2 ("linear-1Mbp"): an i.i.d. uniform 1,000,000 bp reference chopped into
32 bp nodes (a chain DAG), N reads x 150 bp from uniform positions, forward
strand, substitutions 1 %, indels 0.1 % (geometric length, mean 2).  Per read the
host extracts the window [start-117, start+150+117) snapped outward to whole
nodes (R = 384 bp = 12 nodes away from the ends) — the >= 75 bp flank `vg map`
adds via longest_detectable_gap (src/mapper.cpp:2440, src/alignment_scorer.cpp:264-271).
All problems point into ONE shared reference array (no per-read copies): the C ABI
takes plain pointers, so windows may overlap freely.
"""
import numpy as np

from . import capi

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
NODE = 32
FLANK = 117


def longest_detectable_gap(read_length, read_pos, match=1, gap_open=6, gap_extend=1, bonus=5):
    """EditAlignmentScorer::longest_detectable_gap (src/alignment_scorer.cpp:264-271); vectorised over numpy arrays.  C++ integer
    division truncates towards zero."""
    read_length = np.asarray(read_length, dtype=np.int64); read_pos = np.asarray(read_pos, dtype=np.int64)
    overhang = np.minimum(read_pos, read_length - read_pos)
    numer = match * overhang + bonus - gap_open
    gap = np.where(numer >= 0, numer // gap_extend, -((-numer) // gap_extend)) + 1
    out = np.where((gap >= 0) & (overhang > 0), gap, 0)
    return int(out) if out.ndim == 0 else out


def make_reference(length=1_000_000, seed=42):
    rng = np.random.default_rng(seed)
    return ACGT[rng.integers(0, 4, length)]


class LinearWorkload:
    """Holds the numpy arenas and the vgk_gssw_problem array for config 2."""

    STREAM_CHUNK = 65536      # reads of the global stream are generated chunk by chunk, chunk c from the seed (seed, c)

    def __init__(self, n_reads, read_len=150, ref_len=1_000_000, seed=43, ref_seed=42,
                 sub_rate=0.01, indel_rate=0.001, flags=capi.VGK_GSSW_LOCAL | capi.VGK_GSSW_TRACEBACK, stream_begin=0):
        """Reads [stream_begin, stream_begin + n_reads) of ONE global read stream: any cut of the stream into shards (bench.py
        --gpus N through shard.shard_range) sees the same reads as a single process that takes it whole."""
        self.ref = make_reference(ref_len, ref_seed)
        self.n = n_reads
        self.read_len = read_len
        n_ref_nodes = ref_len // NODE
        # sample a little more reference than the read needs so deletions can be absorbed
        span = read_len + 16
        CH = self.STREAM_CHUNK
        starts, read_blocks = [], []
        for c in range(stream_begin // CH, (stream_begin + max(n_reads, 1) - 1) // CH + 1):
            rng = np.random.default_rng([seed, c])
            m = CH
            start_c = rng.integers(0, ref_len - span, m)
            reads_c = np.empty((m, read_len), dtype=np.uint8)
            src = self.ref[start_c[:, None] + np.arange(span)[None, :]]   # (m, span) template bases
            # substitutions (to a uniformly random base, as vg sim does)
            sub = rng.random((m, span)) < sub_rate
            src[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
            reads_c[:] = src[:, :read_len]
            # indels: rare -> per-read fix-up loop over the affected reads only
            ev = rng.random((m, read_len)) < indel_rate
            for r in np.nonzero(ev.any(axis=1))[0]:
                tmpl = src[r]
                out = []
                i = 0
                evr = ev[r]
                while len(out) < read_len and i < span:
                    if i < read_len and evr[i]:
                        ln = int(rng.geometric(0.5))
                        if rng.random() < 0.5:
                            out.extend(ACGT[rng.integers(0, 4, ln)])   # insertion
                        else:
                            i += ln                                    # deletion
                            continue
                    out.append(tmpl[i]); i += 1
                while len(out) < read_len:
                    out.append(ACGT[rng.integers(0, 4)])
                reads_c[r] = np.array(out[:read_len], dtype=np.uint8)
            lo = max(stream_begin - c * CH, 0); hi = min(stream_begin + n_reads - c * CH, CH)
            starts.append(start_c[lo:hi]); read_blocks.append(reads_c[lo:hi])
        start = np.concatenate(starts) if starts else np.zeros(0, dtype=np.int64)
        reads = np.concatenate(read_blocks) if read_blocks else np.zeros((0, read_len), dtype=np.uint8)
        if n_reads == 0:
            start = start[:0]; reads = reads[:0]
        self.reads = reads.reshape(-1)
        self.start = start
        # windows snapped outward to whole nodes, clipped to the reference
        first = np.maximum((start - FLANK) // NODE, 0)
        last = np.minimum((start + read_len + FLANK + NODE - 1) // NODE, n_ref_nodes)   # exclusive
        n_nodes = (last - first).astype(np.int64)
        self.first_node = first
        self.n_nodes = n_nodes
        max_nodes = int(n_nodes.max())
        # shared templates: every node is 32 bp; chain predecessors
        self.node_len_t = np.full(max_nodes, NODE, dtype=np.uint32)
        self.pred_off_t = np.concatenate([[0], np.arange(0, max_nodes)]).astype(np.uint32)  # [0,0,1,2,...]
        self.pred_idx_t = np.arange(max_nodes, dtype=np.uint32)                              # node v -> v-1
        arr = np.zeros(n_reads, dtype=capi.PROBLEM_DT)
        arr["read"] = self.reads.ctypes.data + np.arange(n_reads, dtype=np.int64) * read_len
        arr["read_len"] = read_len
        arr["flags"] = flags
        g = arr["graph"]
        g["n_nodes"] = n_nodes
        g["node_len"] = self.node_len_t.ctypes.data
        g["seq"] = self.ref.ctypes.data + first * NODE
        g["pred_off"] = self.pred_off_t.ctypes.data
        g["pred_idx"] = self.pred_idx_t.ctypes.data
        arr["graph"] = g
        self.array = arr
        self.R = n_nodes * NODE
        self.flags = flags

    @property
    def ptr(self):
        return self.array.ctypes.data

    def graph_arrays(self):
        """The whole reference as ONE graph (node_len, seq, pred_off, pred_idx) for vgk_graph_create: a chain of 32 bp nodes."""
        n = len(self.ref) // NODE
        pred_off = np.concatenate([[0], np.arange(0, n)]).astype(np.uint32)
        return np.full(n, NODE, dtype=np.uint32), self.ref[:n * NODE], pred_off, np.arange(max(n - 1, 1), dtype=np.uint32)

    def windows(self):
        """The same problems as windows of that graph (vgk_gssw_pack_windows): {read, first node, node count}."""
        return capi.WindowSet(self.reads[:self.n * self.read_len], np.arange(self.n + 1, dtype=np.int64) * self.read_len,
                              self.first_node, self.n_nodes, self.flags, cols=self.R)

    # duck-typing the bits of capi.ProblemSet that Engine/Batch use
    @property
    def read_off(self):
        return np.arange(self.n + 1, dtype=np.int64) * self.read_len

    @property
    def seq_off(self):
        return np.concatenate([[0], np.cumsum(self.R)])

    def subset(self, k):
        """First k problems as an independent view (same arenas)."""
        s = object.__new__(LinearWorkload)
        s.__dict__.update(self.__dict__)
        s.n = k; s.array = self.array[:k]; s.R = self.R[:k]; s.n_nodes = self.n_nodes[:k]
        s.start = self.start[:k]; s.first_node = self.first_node[:k]
        return s

    def cells(self):
        return int((self.R * self.read_len).sum())


def build_variation_graph(rng, graph_bp, snp_every, indel_every):
    """24-32 bp chain nodes, SNP bubbles (two 1-bp alternatives) and insertion bubbles; nodes come out in topological order."""
    seqs, preds, kind = [], [], []           # kind: 0 chain, 1 snp alt, 2 insertion
    pos = 0
    last_chain = -1
    while pos < graph_bp:
        ln = int(rng.integers(24, 33))
        seqs.append(ACGT[rng.integers(0, 4, ln)]); kind.append(0)
        v = len(seqs) - 1
        if last_chain < 0:
            preds.append([])
        else:
            preds.append(list(pending))
        pending = [v]
        pos += ln
        r = rng.random()
        if r < ln / snp_every:                       # SNP bubble after this chain node
            a = len(seqs); seqs.append(ACGT[rng.integers(0, 4, 1)]); kind.append(1); preds.append([v])
            b = len(seqs); seqs.append(ACGT[rng.integers(0, 4, 1)]); kind.append(1); preds.append([v])
            pending = [a, b]; pos += 1
        elif r < ln / snp_every + ln / indel_every:  # insertion: v -> ins -> next, and v -> next
            k = int(rng.integers(1, 21))
            a = len(seqs); seqs.append(ACGT[rng.integers(0, 4, k)]); kind.append(2); preds.append([v])
            pending = [v, a]
        last_chain = v
    return seqs, preds, kind


class TailWorkload:
    """config 3 stand-in: giraffe-style tail alignments (pinned X-drop) on a variation graph.

    One long synthetic haplotype graph is built once: 24-32 bp chain nodes, SNP bubbles (two 1-bp
    alternatives) every ~snp_every bp and insertion bubbles (chain -> inserted node -> chain, plus the
    skipping edge) every ~indel_every bp.  Each problem is a window of consecutive nodes starting at
    a chain node (the pin), deep enough for the tail plus its longest detectable gap
    (search_limit = tail + gap, src/minimizer_mapper.cpp:5809-5816); the read tail is a walk from the
    pin through random alternatives with 1 % substitutions.  Mode VGK_XDROP_PINNED, left pin.
    """

    def __init__(self, n_reads, seed=77, graph_bp=2_000_000, snp_every=100, indel_every=1000, max_tail=121,
                 flags=capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK):
        rng = np.random.default_rng(seed)
        seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
        self.g_seq = np.concatenate(seqs)
        g_len = np.array([len(s) for s in seqs], dtype=np.uint32)
        g_off = np.concatenate([[0], np.cumsum(g_len)]).astype(np.int64)
        kind = np.array(kind)
        succ = [[] for _ in seqs]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        chain_idx = np.nonzero(kind == 0)[0]
        chain_idx = chain_idx[chain_idx < len(seqs) - 64]
        # ---- problems -----------------------------------------------------------------------------
        tails = rng.integers(1, max_tail + 1, n_reads)
        starts = chain_idx[rng.integers(0, len(chain_idx), n_reads)]
        # giraffe: max_gap = longest_detectable_gap(read length, tail length) with the tail's end as the read position
        # (src/minimizer_mapper.cpp:5809-5816); for a tail of t bases off a read of >= 2 t that is the overhang t
        gap = np.maximum(longest_detectable_gap(2 * tails, tails), 1)
        depth_bp = tails + np.minimum(gap, tails)                             # = 2t for t <= 75
        reads, read_off, node_len, node_off, pred_off, pred_idx, edge_off, seq_chunks, seq_off = [], [0], [], [0], [], [], [0], [], [0]
        max_gap = []
        for i in range(n_reads):
            a = int(starts[i]); need = int(depth_bp[i]); v = a; bp = 0
            while bp < need and v < len(seqs) - 1:
                if kind[v] != 1 or kind[v - 1] != 1:      # count a bubble's depth once
                    bp += int(g_len[v])
                v += 1
            while kind[v] != 0:                           # never cut a window inside a bubble
                v += 1
            b = v                                         # window = nodes [a, b)
            node_len.append(g_len[a:b]); node_off.append(node_off[-1] + (b - a))
            off = [0]
            for u in range(a, b):
                pr = [p - a for p in preds[u] if p >= a]
                pred_idx.extend(pr); off.append(off[-1] + len(pr))
            pred_off.extend(off); edge_off.append(edge_off[-1] + off[-1])
            seq_chunks.append((int(g_off[a]), int(g_off[b]))); seq_off.append(seq_off[-1] + int(g_off[b] - g_off[a]))
            # read tail: random walk from the pin
            out = []; u = a; o = 0
            while len(out) < tails[i]:
                if o >= g_len[u]:
                    nx = [w for w in succ[u] if w < b]
                    if not nx:
                        break
                    u = nx[int(rng.integers(0, len(nx)))]; o = 0
                    continue
                out.append(self.g_seq[g_off[u] + o]); o += 1
            out = np.array(out, dtype=np.uint8)
            sub = rng.random(len(out)) < 0.01
            out[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
            reads.append(out); read_off.append(read_off[-1] + len(out))
            max_gap.append(int(min(gap[i], 65535)))
        seq = np.concatenate([self.g_seq[s:e] for s, e in seq_chunks])
        self.ps = capi.ProblemSet(np.concatenate(reads), read_off, np.concatenate(node_len), node_off, seq, seq_off,
                                  pred_off, pred_idx, edge_off, np.full(n_reads, flags, dtype=np.uint32), None, max_gap)
        self.n = n_reads
        self.tails = tails

    def cells(self):
        return int(((np.diff(self.ps.read_off) + 1) * np.diff(self.ps.seq_off)).sum())


class BandedWorkload:
    """config 5 stand-in: the "middle" connections of long-read chaining — a global alignment between two anchors with
    BandedGlobalAligner (src/minimizer_mapper_from_chains.cpp:3773).  Each problem is a window of the variation graph from
    one chain node to another, 30..max_len bp long (giraffe's hifi preset connects anchors up to 233 bp apart and allows
    500 bp gaps, src/subcommand/giraffe_main.cpp:996-1000); the read is a source-to-sink walk through the window with HiFi-like
    errors (0.5 % substitutions, 0.3 % indels); band padding = floor(sqrt(L)) + 1 (algorithms::pad_band_random_walk's
    defaults, src/algorithms/pad_band.hpp:21-23), permissive banding as the caller asks."""

    def __init__(self, n_reads, seed=99, graph_bp=1_000_000, snp_every=100, indel_every=1000, max_len=500):
        rng = np.random.default_rng(seed)
        seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
        g_len = np.array([len(s) for s in seqs], dtype=np.int64)
        kind = np.array(kind)
        succ = [[] for _ in seqs]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        chain_idx = np.nonzero(kind == 0)[0]
        chain_idx = chain_idx[chain_idx < len(seqs) - 64]
        target = rng.integers(30, max_len + 1, n_reads)
        starts = chain_idx[rng.integers(0, len(chain_idx), n_reads)]
        problems = []
        for i in range(n_reads):
            a = int(starts[i]); v = a; bp = 0
            while bp < target[i] and v < len(seqs) - 2:
                if kind[v] != 1 or kind[v - 1] != 1:
                    bp += int(g_len[v])
                v += 1
            while kind[v] != 0:
                v += 1
            b = v + 1                                      # window = nodes [a, b), ends on a chain node: one source, one sink
            nodes = [seqs[u].tobytes().decode() for u in range(a, b)]
            pr = [[p - a for p in preds[u] if p >= a] for u in range(a, b)]
            out = []; u = a
            while True:
                out.extend(seqs[u].tolist())
                nx = [w for w in succ[u] if w < b]
                if not nx:
                    break
                u = nx[int(rng.integers(0, len(nx)))]
            read = []
            for c in out:
                r = rng.random()
                if r < 0.005:
                    read.append(int(ACGT[rng.integers(0, 4)]))
                elif r < 0.0065:
                    continue
                elif r < 0.008:
                    read.append(int(ACGT[rng.integers(0, 4)])); read.append(c)
                else:
                    read.append(c)
            read = bytes(read).decode() if read else "A"
            problems.append(dict(read=read, nodes=nodes, preds=pr, band_padding=int(np.sqrt(len(read))) + 1, permissive=True))
        self.problems = problems
        self.bs = capi.BandedSet.from_lists(problems)
        self.n = n_reads


class GaplessWorkload:
    """giraffe's first extension stage (configs[2]/[3]): haplotype-consistent gapless extension of minimizer seeds
    (MinimizerMapper::extend_seed_group -> GaplessExtender::extend, src/minimizer_mapper.cpp:4784).  A variation graph with
    `n_haplotypes` random threads (every SNP / insertion allele chosen independently); 150 bp reads sampled from a thread on either
    strand with 1 % substitutions; `seeds_per_read` seeds at true positions (distinct read offsets), as a minimizer index
    would report for an error-free k-mer."""

    def __init__(self, n_reads, seed=123, graph_bp=1_000_000, n_haplotypes=8, read_len=150, seeds_per_read=6, snp_every=100, indel_every=1000, inserted_reads=0.0):
        """inserted_reads: fraction of the reads that carry one extra base at a random position (a sequencing insertion): no gapless
        extension covers such a read, so its cluster leaves tails (the giraffe workload)"""
        rng = np.random.default_rng(seed)
        seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
        n_nodes = len(seqs)
        kind = np.array(kind)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        succ = [[] for _ in seqs]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        self.nodes = [s.tobytes().decode() for s in seqs]
        # threads: walk from node 0, pick a random successor at every branch
        threads = []
        for _ in range(n_haplotypes):
            t = [0]; v = 0
            while succ[v]:
                v = succ[v][int(rng.integers(0, len(succ[v])))]
                t.append(v)
            threads.append(np.array(t, dtype=np.int64))
        self.threads = [list((2 * t).astype(int)) for t in threads]
        comp = np.zeros(256, dtype=np.uint8); comp[:] = np.arange(256)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        hap_seq = [np.concatenate([seqs[v] for v in t]) for t in threads]
        hap_start = [np.concatenate([[0], np.cumsum(lens[t])]) for t in threads]
        which = rng.integers(0, n_haplotypes, n_reads)
        rev = rng.random(n_reads) < 0.5
        reads = np.zeros((n_reads, read_len), dtype=np.uint8)
        seeds = np.zeros(n_reads * seeds_per_read, dtype=capi.SEED_DT)
        for hidx in range(n_haplotypes):
            sel = np.nonzero(which == hidx)[0]
            if not len(sel):
                continue
            hs = hap_seq[hidx]; st = hap_start[hidx]; t = threads[hidx]
            a = rng.integers(0, len(hs) - read_len, len(sel))
            fw = hs[a[:, None] + np.arange(read_len)[None, :]]                   # forward-strand window of every read
            r = rev[sel]
            rd = np.where(r[:, None], comp[fw[:, ::-1]], fw)
            sub = rng.random(rd.shape) < 0.01
            rd = np.where(sub, ACGT[rng.integers(0, 4, rd.shape)], rd)
            reads[sel] = rd
            ro = np.sort(np.argsort(rng.random((len(sel), read_len)), axis=1)[:, :seeds_per_read], axis=1)   # distinct read offsets
            g = np.where(r[:, None], a[:, None] + read_len - 1 - ro, a[:, None] + ro)                        # forward-strand base hit by the seed
            k = np.searchsorted(st, g, side="right") - 1
            node = t[k]; off = g - st[k]
            off_o = np.where(r[:, None], lens[node] - 1 - off, off)                                          # offset on the read's strand
            flat = (sel[:, None] * seeds_per_read + np.arange(seeds_per_read)[None, :]).ravel()
            seeds["node"][flat] = (2 * node + r[:, None]).ravel()
            diff = ro - off_o
            if inserted_reads > 0:
                ins = rng.random(len(sel)) < inserted_reads
                p = rng.integers(20, read_len - 20, len(sel))
                col = np.arange(read_len)[None, :]
                src = np.where(col > p[:, None], col - 1, col)                    # read' = read[:p] + X + read[p:-1]
                shifted = np.take_along_axis(rd, src, axis=1)
                shifted[np.arange(len(sel)), p] = ACGT[rng.integers(0, 4, len(sel))]
                reads[sel] = np.where(ins[:, None], shifted, rd)
                moved = ins[:, None] & (ro >= p[:, None])                         # the bases behind the insertion sit one read position later
                diff = diff + moved
                gone = moved & (ro + 1 >= read_len)
                diff = np.where(gone, diff[:, :1], diff)                          # (a seed pushed off the read: repeat the first one; the de-duplication drops it)
                seeds["node"][flat] = np.where(gone, (2 * node + r[:, None])[:, :1], 2 * node + r[:, None]).ravel()
            seeds["diff"][flat] = diff.ravel()
        # a cluster is a set: drop seeds that repeat (same node and diagonal) inside a read
        keep = np.ones(len(seeds), dtype=bool)
        s2 = seeds.reshape(n_reads, seeds_per_read)
        for i in range(1, seeds_per_read):
            for j in range(i):
                keep.reshape(n_reads, seeds_per_read)[:, i] &= ~((s2["node"][:, i] == s2["node"][:, j]) & (s2["diff"][:, i] == s2["diff"][:, j]))
        counts = keep.reshape(n_reads, seeds_per_read).sum(axis=1)
        seed_off = np.concatenate([[0], np.cumsum(counts)])
        self.gs = capi.GaplessSet(reads.ravel(), np.arange(n_reads + 1) * read_len, seeds[keep], seed_off,
                                  node_cap=int(counts.sum()) * 16, mism_cap=int(counts.sum()) * 12)
        self.n = n_reads
        self.read_len = read_len


class WfaWorkload:
    """The long-read chaining stage's WFA problems (configs[5]): WFAExtender::connect between consecutive anchors
    (MinimizerMapper connect_consistently, src/minimizer_mapper.cpp:2955, :3925) and ::prefix / ::suffix for the read tails
    (<= max_tail_length, giraffe_main.cpp:997).  The variation graph of the gapless workload with `n_haplotypes` random threads;
    every problem is a window of a thread (either strand) with HiFi-like errors (0.5 %: half substitutions, half 1-bp indels).
    `tail_fraction` of the problems are tails of up to `max_tail` bases, the rest connect two anchors 50..`max_connect` bases apart."""

    def __init__(self, n_problems, seed=321, graph_bp=1_000_000, n_haplotypes=8, max_connect=250, max_tail=100, tail_fraction=0.2,
                 error_rate=0.005, snp_every=100, indel_every=1000):
        rng = np.random.default_rng(seed)
        seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        succ = [[] for _ in seqs]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        self.nodes = [s.tobytes().decode() for s in seqs]
        threads = []
        for _ in range(n_haplotypes):
            t = [0]; v = 0
            while succ[v]:
                v = succ[v][int(rng.integers(0, len(succ[v])))]
                t.append(v)
            threads.append(np.array(t, dtype=np.int64))
        self.threads = [list((2 * t).astype(int)) for t in threads]
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        hap_seq = [np.concatenate([seqs[v] for v in t]) for t in threads]
        hap_start = [np.concatenate([[0], np.cumsum(lens[t])]) for t in threads]
        n = n_problems
        which = rng.integers(0, n_haplotypes, n)
        rev = rng.random(n) < 0.5
        tail = rng.random(n) < tail_fraction
        mode = np.where(tail, np.where(rng.random(n) < 0.5, capi.WFA_SUFFIX, capi.WFA_PREFIX), capi.WFA_CONNECT).astype(np.uint32)
        span = np.where(tail, rng.integers(1, max_tail + 1, n), rng.integers(50, max_connect + 1, n))      # graph bases between from and to
        from_node = np.full(n, capi.WFA_NO_NODE, dtype=np.uint32); from_off = np.zeros(n, dtype=np.uint32)
        to_node = np.full(n, capi.WFA_NO_NODE, dtype=np.uint32); to_off = np.zeros(n, dtype=np.uint32)
        pieces = [None] * n
        for hidx in range(n_haplotypes):
            sel = np.nonzero(which == hidx)[0]
            if not len(sel):
                continue
            hs = hap_seq[hidx]; st = hap_start[hidx]; t = threads[hidx]
            f = rng.integers(1, len(hs) - span[sel] - 2)                    # forward-strand position of the base before the window
            for j, i in enumerate(sel):
                a = int(f[j]); s = int(span[i])
                w = hs[a + 1:a + 1 + s]
                lo, hi = a, a + 1 + s                                       # flanking bases on the forward strand
                if rev[i]:
                    w = comp[w[::-1]]; lo, hi = hi, lo
                # errors: substitutions and 1-bp indels
                e = rng.random(len(w)) < error_rate
                if e.any():
                    out = []
                    for c, bad in zip(w, e):
                        if not bad:
                            out.append(c); continue
                        r = rng.random()
                        if r < 0.5:
                            out.append(ACGT[int(rng.integers(0, 4))])
                        elif r < 0.75:
                            pass
                        else:
                            out.append(c); out.append(ACGT[int(rng.integers(0, 4))])
                    w = np.array(out, dtype=np.uint8)
                pieces[i] = w

                def pos(g):
                    k = int(np.searchsorted(st, g, side="right") - 1)
                    node = int(t[k]); off = int(g - st[k])
                    return (2 * node + 1, int(lens[node]) - 1 - off) if rev[i] else (2 * node, off)
                if mode[i] != capi.WFA_PREFIX:
                    from_node[i], from_off[i] = pos(lo)
                if mode[i] != capi.WFA_SUFFIX:
                    to_node[i], to_off[i] = pos(hi)
        seq_off = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.int64)
        buf = np.concatenate(pieces) if seq_off[-1] else np.zeros(1, np.uint8)
        self.ws = capi.WfaSet(buf, seq_off, mode, from_node, from_off, to_node, to_off,
                              path_cap=int(n) * 16 + int(seq_off[-1]) // 4, edit_cap=int(n) * 8)
        self.n = n
        self.bases = int(seq_off[-1])
        self.graph_bases = int(span.sum())


# ---- configs[2] at its stated size: a chr22-scale SNP + indel graph, giraffe-style tails as windows of the resident graph ----------
def _comp_table():
    t = np.arange(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        t[a] = b
    return t


class VariationGraph:
    """a synthetic reference (uniform ACGT, 41 % GC, seed 22) with SNPs at 1/1000 and indels of 1-20 bp at
    1/10 000 (seed 23; sites closer than 40 bp to the previous one are dropped so that bubbles never nest), as a DAG of <= 32 bp
    nodes in topological order, plus two haplotypes that carry each variant with p = 0.5.  Built with numpy array operations only
    (50.8 Mbp: ~1.7 M nodes in a few seconds).

    node kinds: 0 backbone, 1 SNP alt allele (placed right after its 1-bp reference allele), 2 inserted sequence (placed right
    before the backbone node it precedes).  Edges: backbone chain; both alleles of a SNP share predecessors and successors; an
    insertion hangs between its neighbours, which stay joined; a deletion adds an edge over the deleted backbone nodes."""

    def __init__(self, ref_len=50_818_468, seed=22, var_seed=23, snp_rate=1e-3, indel_rate=1e-4, gc=0.41):
        rng = np.random.default_rng(seed)
        p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
        ref = ACGT[rng.choice(4, size=ref_len, p=p).astype(np.uint8)]
        rng = np.random.default_rng(var_seed)
        n_var = rng.binomial(ref_len, snp_rate + indel_rate)
        pos = np.sort(rng.integers(64, ref_len - 64, n_var))
        keep = np.ones(len(pos), dtype=bool); keep[1:] = np.diff(pos) >= 40
        # (dropping a site can bring its neighbours within 40 bp of each other only if they already were: one pass is enough)
        pos = pos[keep]
        kind = np.where(rng.random(len(pos)) < snp_rate / (snp_rate + indel_rate), 0, np.where(rng.random(len(pos)) < 0.5, 1, 2))   # 0 SNP, 1 insertion, 2 deletion
        vlen = np.where(kind == 0, 1, rng.integers(1, 21, len(pos)))
        # backbone cuts: every 32 bp, and around every variant (SNP: the base itself; deletion: the deleted stretch; insertion: before pos)
        cuts = np.concatenate([np.arange(0, ref_len, NODE), pos, pos[kind == 0] + 1, (pos + vlen)[kind == 2], [ref_len]])
        cuts = np.unique(cuts)
        b_start, b_end = cuts[:-1], cuts[1:]
        nb = len(b_start)
        # extra nodes: SNP alt (after backbone node starting at pos) and insertion (before backbone node starting at pos)
        snp_pos = pos[kind == 0]; ins_pos = pos[kind == 1]; ins_len = vlen[kind == 1]; del_pos = pos[kind == 2]; del_len = vlen[kind == 2]
        snp_b = np.searchsorted(b_start, snp_pos); ins_b = np.searchsorted(b_start, ins_pos)
        alt = ACGT[(np.searchsorted(ACGT, ref[snp_pos]) + rng.integers(1, 4, len(snp_pos))) % 4]
        ins_seq = ACGT[rng.integers(0, 4, int(ins_len.sum()))]
        # order keys: backbone node i -> 3 i + 1, its SNP alt -> 3 i + 2, an insertion before it -> 3 i
        keys = np.concatenate([3 * np.arange(nb) + 1, 3 * snp_b + 2, 3 * ins_b])
        order = np.argsort(keys, kind="stable")
        n = len(keys)
        new_index = np.empty(n, dtype=np.int64); new_index[order] = np.arange(n)
        bb = new_index[:nb]; alt_idx = new_index[nb:nb + len(snp_b)]; ins_idx = new_index[nb + len(snp_b):]
        node_len = np.empty(n, dtype=np.uint32)
        node_len[bb] = (b_end - b_start).astype(np.uint32); node_len[alt_idx] = 1; node_len[ins_idx] = ins_len
        col = np.concatenate([[0], np.cumsum(node_len, dtype=np.int64)])
        seq = np.empty(int(col[-1]), dtype=np.uint8)
        # backbone bases land at their node's columns: one scatter through a per-base offset (column of base x = col[bb[node of x]] + x - start)
        base_node = np.repeat(np.arange(nb), b_end - b_start)
        seq[col[bb][base_node] + (np.arange(ref_len) - b_start[base_node])] = ref
        seq[col[alt_idx]] = alt
        ins_off = np.concatenate([[0], np.cumsum(ins_len)])
        ins_node = np.repeat(np.arange(len(ins_len)), ins_len)
        seq[col[ins_idx][ins_node] + (np.arange(len(ins_seq)) - ins_off[ins_node])] = ins_seq
        # predecessors.  prev(i) = backbone i - 1; a SNP at backbone s: alt shares s's predecessors, and s + 1 also follows the alt;
        # an insertion before backbone j: ins follows j - 1 (and the alt of j - 1 — cannot be: sites are >= 40 bp apart), j follows ins too;
        # a deletion [p, p + k): backbone node starting at p + k also follows the backbone node ending at p
        src = [bb[:-1]]; dst = [bb[1:]]
        src.append(bb[snp_b - 1]); dst.append(alt_idx)                   # into the alt
        src.append(alt_idx); dst.append(bb[snp_b + 1])                   # out of the alt
        src.append(bb[ins_b - 1]); dst.append(ins_idx)
        src.append(ins_idx); dst.append(bb[ins_b])
        del_from = np.searchsorted(b_start, del_pos) - 1; del_to = np.searchsorted(b_start, del_pos + del_len)
        src.append(bb[del_from]); dst.append(bb[del_to])
        src = np.concatenate(src); dst = np.concatenate(dst)
        assert (src < dst).all()
        e = np.lexsort((src, dst))                                        # grouped by destination, predecessors ascending
        dst, src = dst[e], src[e]
        self.node_len = node_len; self.seq = seq; self.col = col
        self.pred_off = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).astype(np.uint32)
        self.pred_idx = src.astype(np.uint32)
        self.n_nodes = n
        self.plain = np.zeros(n, dtype=bool); self.plain[bb] = True      # plain backbone nodes: pins start there ...
        self.plain[bb[snp_b]] = False; self.plain[bb[ins_b]] = False; self.plain[bb[np.minimum(snp_b + 1, nb - 1)]] = False   # ... but not at or right after a bubble
        # haplotypes: which variants each carries; as base arrays + per-base (node, offset)
        self.haps = []
        for h in range(2):
            has_snp = rng.random(len(snp_b)) < 0.5; has_ins = rng.random(len(ins_b)) < 0.5; has_del = rng.random(len(del_from)) < 0.5
            on = np.ones(n, dtype=bool)                                    # nodes on the haplotype's walk
            on[alt_idx] = has_snp; on[bb[snp_b]] = ~has_snp; on[ins_idx] = has_ins
            skip = np.zeros(nb + 1, dtype=np.int64)                        # deleted backbone nodes: (del_from, del_to) exclusive
            np.add.at(skip, del_from[has_del] + 1, 1); np.add.at(skip, del_to[has_del], -1)
            on[bb[np.cumsum(skip[:nb]) > 0]] = False
            nodes_on = np.nonzero(on)[0]
            start = np.concatenate([[0], np.cumsum(node_len[nodes_on], dtype=np.int64)])
            hseq = np.empty(int(start[-1]), dtype=np.uint8)
            rep = np.repeat(np.arange(len(nodes_on)), node_len[nodes_on])
            hseq[:] = seq[col[nodes_on][rep] + (np.arange(len(hseq)) - start[rep])]
            hap_pos = np.full(n, -1, dtype=np.int64); hap_pos[nodes_on] = start[:-1]          # where a node starts on this haplotype (-1: not on it)
            self.haps.append((hseq, hap_pos))

    def arrays(self):
        return self.node_len, self.seq, self.pred_off, self.pred_idx

    def reverse_complement(self):
        """The other strand as a graph of its own (node i -> n - 1 - i, sequences reverse-complemented, edges flipped): right tails
        are left-pinned problems on it (src/minimizer_mapper.cpp:5665, :5720-5725)."""
        n = self.n_nodes
        g = object.__new__(VariationGraph)
        g.n_nodes = n
        g.node_len = self.node_len[::-1].copy()
        g.seq = _comp_table()[self.seq[::-1]]
        g.col = np.concatenate([[0], np.cumsum(g.node_len, dtype=np.int64)])
        dst = np.repeat(np.arange(n), np.diff(self.pred_off.astype(np.int64)))     # forward edges src -> dst
        src = self.pred_idx.astype(np.int64)
        rs, rd = n - 1 - dst, n - 1 - src                                           # flipped: (n-1-dst) -> (n-1-src)
        e = np.lexsort((rs, rd)); rs, rd = rs[e], rd[e]
        g.pred_off = np.concatenate([[0], np.cumsum(np.bincount(rd, minlength=n))]).astype(np.uint32)
        g.pred_idx = rs.astype(np.uint32)
        g.plain = self.plain[::-1].copy()
        g.haps = []
        comp = _comp_table()
        for hseq, hap_pos in self.haps:
            on = hap_pos >= 0
            ends = np.where(on, hap_pos + self.node_len, -1)
            rp = np.where(on[::-1], len(hseq) - ends[::-1], -1)
            g.haps.append((comp[hseq[::-1]], rp))
        # a node right BEFORE a bubble on the forward strand is right after it here
        prev_bad = np.zeros(n, dtype=bool); prev_bad[1:] = ~g.plain[:-1]
        g.plain &= ~prev_bad
        return g


class GraphTails:
    """n tails of 1..max_tail bases against `graph`: each starts at a plain backbone node (the pin) on one of the two haplotypes,
    follows that haplotype with 1 % substitutions, and is aligned left-pinned X-drop against the window of nodes that covers
    tail + longest_detectable_gap(2 tail, tail) more bases (src/minimizer_mapper.cpp:5809-5816) — as vgk_window_problem's."""

    def __init__(self, graph, n, seed=7, max_tail=121, sub_rate=0.01, flags=capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK):
        rng = np.random.default_rng(seed)
        tails = rng.integers(1, max_tail + 1, n)
        h = rng.integers(0, 2, n)
        cand = np.nonzero(graph.plain)[0]
        cand = cand[cand < graph.n_nodes - 64]
        pin = cand[rng.integers(0, len(cand), n)]
        for k in range(2):                                       # the pin must lie on the chosen haplotype: flip the haplotype, else redraw
            off = np.where(h == 0, graph.haps[0][1][pin], graph.haps[1][1][pin])
            bad = off < 0
            h[bad] ^= 1
        off = np.where(h == 0, graph.haps[0][1][pin], graph.haps[1][1][pin])
        bad = off < 0
        while bad.any():                                         # (a plain node deleted on both haplotypes: rare)
            pin[bad] = cand[rng.integers(0, len(cand), int(bad.sum()))]
            off = np.where(h == 0, graph.haps[0][1][pin], graph.haps[1][1][pin]); bad = off < 0
        read_off = np.concatenate([[0], np.cumsum(tails)]).astype(np.int64)
        reads = np.empty(int(read_off[-1]), dtype=np.uint8)
        CH = 262144                                              # fixed-width gathers, chunk by chunk (bounds host memory), then compacted
        ar = np.arange(max_tail, dtype=np.int64)[None, :]
        for c0 in range(0, n, CH):
            c1 = min(n, c0 + CH)
            t = tails[c0:c1]; m = ar < t[:, None]
            block = np.empty((c1 - c0, max_tail), dtype=np.uint8)
            for hh in (0, 1):
                sel = np.nonzero(h[c0:c1] == hh)[0]
                hseq = graph.haps[hh][0]
                block[sel] = hseq[np.minimum(off[c0:c1][sel, None] + ar, len(hseq) - 1)]
            sub = rng.random(block.shape) < sub_rate
            block[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
            reads[read_off[c0]:read_off[c1]] = block[m]
        gap = np.maximum(longest_detectable_gap(2 * tails, tails), 1)
        depth = tails + np.minimum(gap, tails)
        # the window: nodes from the pin up to the first whose end is >= depth columns past the pin's start, plus the 20-base
        # insertion / 1-base alleles a walk of that depth may pass (columns count every node, so this over-covers slightly)
        last = np.searchsorted(graph.col, graph.col[pin] + depth + 24, side="left")
        last = np.minimum(last, graph.n_nodes)
        n_nodes = (last - pin).astype(np.int64)
        self.graph = graph; self.n = n; self.tails = tails
        self.cols = graph.col[pin + n_nodes] - graph.col[pin]
        self.ws = capi.WindowSet(reads, read_off, pin, n_nodes, flags, max_gap=np.minimum(gap, 65535), cols=self.cols)

    def cells(self):
        return int(((self.tails + 1) * self.cols).sum())

    def subset(self, k, begin=0):
        """problems [begin, begin + k) as a WindowSet over the same arenas"""
        ws = self.ws
        s = object.__new__(capi.WindowSet)
        s.reads = ws.reads; s.read_off = ws.read_off[begin:begin + k + 1]; s.n = k; s.array = ws.array[begin:begin + k]; s.cols = ws.cols[begin:begin + k]
        return s


class TailForestWorkload:
    """giraffe's tail path from the extension onward (configs[2]/[3], the part behind GaplessExtender): for every tail a GBWT search
    state + cut (the end of a gapless extension), the walk distance (tail + longest detectable gap, src/minimizer_mapper.cpp:5816)
    and the tail's bases.  A variation graph with `n_haplotypes` random threads (as GaplessWorkload); a tail starts at a random
    position of a random thread on either strand, with the state of ALL visits of that node (get_state(handle): the widest forest
    the walk can produce there), and follows the thread with 1 % substitutions for 1..max_tail bases.  Cuts never fall on a node's
    end, so every problem yields exactly one tree (a skipped root yields a forest; tests/test_tail_forest.py covers those)."""

    def __init__(self, n_tails, seed=99, graph_bp=1_000_000, n_haplotypes=8, read_len=150, max_tail=121, snp_every=100, indel_every=1000):
        rng = np.random.default_rng(seed)
        seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        succ = [[] for _ in seqs]
        for v, pr in enumerate(preds):
            for p in pr:
                succ[p].append(v)
        self.nodes = [s.tobytes().decode() for s in seqs]
        threads = []
        for _ in range(n_haplotypes):
            t = [0]; v = 0
            while succ[v]:
                v = succ[v][int(rng.integers(0, len(succ[v])))]
                t.append(v)
            threads.append(np.array(t, dtype=np.int64))
        self.threads = [list((2 * t).astype(int)) for t in threads]
        comp = np.arange(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        counts = np.zeros(2 * len(seqs), dtype=np.int64)
        for t in threads:
            c = np.bincount(t, minlength=len(seqs))
            counts[0::2] += c; counts[1::2] += c          # the reverse thread visits the other strand of the same nodes
        which = rng.integers(0, n_haplotypes, n_tails); rev = rng.random(n_tails) < 0.5
        tail_len = rng.integers(1, max_tail + 1, n_tails)
        probs = np.zeros(n_tails, dtype=capi.TAIL_DT)
        tails = np.zeros((n_tails, max_tail), dtype=np.uint8)
        for hidx in range(n_haplotypes):
            for strand in (0, 1):
                sel = np.nonzero((which == hidx) & (rev == bool(strand)))[0]
                if not len(sel):
                    continue
                t = threads[hidx]
                if strand:
                    t = t[::-1]
                hs = np.concatenate([seqs[v] if not strand else comp[seqs[v][::-1]] for v in t])
                st = np.concatenate([[0], np.cumsum(lens[t])])
                a = rng.integers(1, len(hs) - max_tail - 1, len(sel))             # the tail's first base on this strand of the thread
                k = np.searchsorted(st, a, side="right") - 1
                off = a - st[k]                                                   # cut: bases [off, len) of the node belong to the tail
                node = 2 * t[k] + strand
                probs["node"][sel] = node; probs["lo"][sel] = 0; probs["hi"][sel] = counts[node] - 1; probs["offset"][sel] = off
                w = hs[a[:, None] + np.arange(max_tail)[None, :]]
                sub = rng.random(w.shape) < 0.01
                tails[sel] = np.where(sub, ACGT[rng.integers(0, 4, w.shape)], w)
        gap = longest_detectable_gap(read_len, tail_len)
        probs["walk_distance"] = tail_len + gap
        self.problems = probs; self.tail_len = tail_len; self.max_gap = gap; self.n = n_tails
        keep = np.arange(max_tail)[None, :] < tail_len[:, None]
        self.reads = tails[keep]                                                  # flat, tail i = reads[read_off[i]:read_off[i+1]]
        self.read_off = np.concatenate([[0], np.cumsum(tail_len)])

    def windows(self, results, lo=0, hi=None):
        """the X-drop problems of tails [lo, hi) once their forests exist: one tree each = nodes [first_node, first_node + n_nodes)"""
        hi = self.n if hi is None else hi
        r = results[lo:hi]
        return capi.WindowSet(self.reads[self.read_off[lo]:self.read_off[hi]], self.read_off[lo:hi + 1] - self.read_off[lo], r["first_node"], r["n_nodes"],
                              capi.VGK_XDROP_PINNED | capi.VGK_GSSW_TRACEBACK, self.max_gap[lo:hi], cols=r["bases"])


class LongReadWorkload:
    """configs[4] as reads, not as loose windows: `n_reads` HiFi-like reads of `read_len` bases, each a walk along a haplotype
    thread (either strand) cut at ANCHORS — error-free 29-mers every 120-400 bases, what the chaining stage hands on — so that a read
    becomes: prefix, anchor, connect, anchor, ..., connect, anchor, suffix.  The stretches between anchors carry 0.5 % errors (half
    substitutions, half 1-bp indels); `sv_fraction` of the connects additionally carry a 25-60 bp insertion, which WFAExtender's
    score cap rejects: those fall back to BandedGlobalAligner between the two anchors, as MinimizerMapper does
    (src/minimizer_mapper.cpp:2955-3100, :3925).  Problems are laid out read by read; `read_of[i]` names problem i's read."""

    def __init__(self, n_reads, seed=515, graph_bp=1_000_000, n_haplotypes=8, read_len=15_000, anchor_len=29, min_gap=120, max_gap=400,
                 error_rate=0.005, sv_fraction=0.01, snp_every=100, indel_every=1000, graph=None):
        """graph: a VariationGraph (the chr22-scale construction of configs[2] / configs[3]: SNPs, insertions AND deletions, two haplotypes that carry
        each variant with p = 0.5) — the reads then follow its haplotypes; without one, a graph of `graph_bp` bases with `n_haplotypes` random threads
        is made here (the tests' small graphs, round 2-5's 1 Mbp leg)."""
        rng = np.random.default_rng(seed)
        if graph is not None:
            lens = graph.node_len.astype(np.int64)
            col = graph.col
            seqs = None
            self.graph = graph                                            # (ChainStage hands its arrays over in bulk)
            self.__dict__["_nodes"] = None; self.__dict__["_preds"] = None
            threads = [np.nonzero(hap_pos >= 0)[0].astype(np.int64) for _, hap_pos in graph.haps]
            n_haplotypes = len(threads)
            self.lens = lens
            self.threads = [(2 * t).astype(np.int64) for t in threads]
            hap_seq = [hseq for hseq, _ in graph.haps]
            hap_start = [np.concatenate([hap_pos[t], [len(hseq)]]).astype(np.int64) for (hseq, hap_pos), t in zip(graph.haps, threads)]
        else:
            seqs, preds, kind = build_variation_graph(rng, graph_bp, snp_every, indel_every)
            lens = np.array([len(s) for s in seqs], dtype=np.int64)
            succ = [[] for _ in seqs]
            for v, pr in enumerate(preds):
                for p in pr:
                    succ[p].append(v)
            self.nodes = [s.tobytes().decode() for s in seqs]
            self.preds = preds; self.lens = lens
            threads = []
            for _ in range(n_haplotypes):
                t = [0]; v = 0
                while succ[v]:
                    v = succ[v][int(rng.integers(0, len(succ[v])))]
                    t.append(v)
                threads.append(np.array(t, dtype=np.int64))
            self.threads = [list((2 * t).astype(int)) for t in threads]
            hap_seq = [np.concatenate([seqs[v] for v in t]) for t in threads]
            hap_start = [np.concatenate([[0], np.cumsum(lens[t])]) for t in threads]
        comp = _comp_table()
        pieces, mode, fn, fo, tn, to, read_of, span, truth = [], [], [], [], [], [], [], [], []
        link_begin, read_total = [], []                                       # where a link starts in its read; its read's length (for longest_detectable_gap_in_range)
        a_off, a_len, a_noff, a_poff, a_nodes = [0], [], [], [0], []          # the anchors, read by read in read order: exact matches along their node paths (chain_stage.hpp)
        self.anchor_bases = np.zeros(n_reads, dtype=np.int64)
        for r in range(n_reads):
            h = int(rng.integers(0, n_haplotypes)); rev = bool(rng.random() < 0.5)
            hs, st, t = hap_seq[h], hap_start[h], threads[h]
            a0 = int(rng.integers(1, len(hs) - read_len - 400))
            # forward-strand segmentation: [tail][anchor][gap][anchor]...[anchor][tail]
            cuts = [a0 + int(rng.integers(20, 100))]
            while cuts[-1] + anchor_len + max_gap + anchor_len + 100 < a0 + read_len:
                cuts.append(cuts[-1] + anchor_len + int(rng.integers(min_gap, max_gap + 1)))
            anchors = [(c, c + anchor_len) for c in cuts]              # [begin, end) on the forward strand
            end = a0 + read_len
            segs = [("tail0", a0, anchors[0][0])] + [("connect", anchors[i][1], anchors[i + 1][0]) for i in range(len(anchors) - 1)] + [("tail1", anchors[-1][1], end)]
            if rev:
                segs = segs[::-1]
            self.anchor_bases[r] = anchor_len * len(anchors)
            for c0, c1 in (anchors[::-1] if rev else anchors):
                k0 = int(np.searchsorted(st, c0, side="right") - 1); k1 = int(np.searchsorted(st, c1 - 1, side="right") - 1)
                walk = [int(t[k]) for k in range(k0, k1 + 1)]
                if rev:
                    a_nodes += [2 * v + 1 for v in walk[::-1]]; a_noff.append(int(st[k1 + 1]) - c1)     # first base on the read's strand = base c1 - 1: its offset from the END of its node
                else:
                    a_nodes += [2 * v for v in walk]; a_noff.append(c0 - int(st[k0]))
                a_len.append(c1 - c0); a_poff.append(len(a_nodes))
            a_off.append(len(a_len))
            pos_in_read = 0; first_link = len(pieces)

            def pos(g):
                k = int(np.searchsorted(st, g, side="right") - 1)
                node = int(t[k]); off = int(g - st[k])
                return (2 * node + 1, int(lens[node]) - 1 - off) if rev else (2 * node, off)
            for what, lo, hi in segs:
                w = hs[lo:hi]
                if rev:
                    w = comp[w[::-1]]
                e = rng.random(len(w)) < error_rate
                if e.any() and graph is not None:                      # (the same error model, drawn array-wise: a chr22-scale run makes 240 Mbp of reads)
                    idx = np.nonzero(e)[0]; x = rng.random(len(idx)); nb = ACGT[rng.integers(0, 4, len(idx))]
                    w = w.copy(); w[idx[x < 0.5]] = nb[x < 0.5]
                    keep_base = np.ones(len(w), dtype=bool); keep_base[idx[(x >= 0.5) & (x < 0.75)]] = False
                    at = idx[x >= 0.75] + 1
                    w = np.insert(w, at, nb[x >= 0.75])[np.insert(keep_base, at, True)]
                elif e.any():
                    out = []
                    for c, bad in zip(w, e):
                        if not bad:
                            out.append(c); continue
                        x = rng.random()
                        if x < 0.5:
                            out.append(ACGT[int(rng.integers(0, 4))])
                        elif x >= 0.75:
                            out.append(c); out.append(ACGT[int(rng.integers(0, 4))])
                    w = np.array(out, dtype=np.uint8) if out else np.zeros(0, np.uint8)
                if what == "connect" and rng.random() < sv_fraction:
                    at = int(rng.integers(0, len(w) + 1))
                    w = np.concatenate([w[:at], ACGT[rng.integers(0, 4, int(rng.integers(25, 61)))], w[at:]])
                # on the read's strand the segment lies between the bases flanking it: from = the base before, to = the base after (exclusive)
                before, after = (lo - 1, hi) if not rev else (hi, lo - 1)
                first_on_read = (what == "tail0") != rev                # the read's own first segment has no anchor before it
                last_on_read = (what == "tail1") != rev
                if first_on_read and what != "connect":
                    mode.append(capi.WFA_PREFIX); fn.append(capi.WFA_NO_NODE); fo.append(0); p = pos(after); tn.append(p[0]); to.append(p[1])
                elif last_on_read and what != "connect":
                    mode.append(capi.WFA_SUFFIX); p = pos(before); fn.append(p[0]); fo.append(p[1]); tn.append(capi.WFA_NO_NODE); to.append(0)
                else:
                    mode.append(capi.WFA_CONNECT); p = pos(before); fn.append(p[0]); fo.append(p[1]); p = pos(after); tn.append(p[0]); to.append(p[1])
                pieces.append(w); read_of.append(r); span.append(hi - lo); truth.append((lo, hi, rev, h))
                link_begin.append(pos_in_read); pos_in_read += len(w) + anchor_len          # an anchor follows every link but the read's last
            read_total += [pos_in_read - anchor_len] * (len(pieces) - first_link)
        seq_off = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.int64)
        buf = np.concatenate(pieces) if seq_off[-1] else np.zeros(1, np.uint8)
        n = len(pieces)
        self.ws = capi.WfaSet(buf, seq_off, np.array(mode, dtype=np.uint32), np.array(fn, dtype=np.uint32), np.array(fo, dtype=np.uint32),
                              np.array(tn, dtype=np.uint32), np.array(to, dtype=np.uint32), path_cap=n * 24 + int(seq_off[-1]) // 4, edit_cap=n * 12)
        self.n = n; self.n_reads = n_reads; self.read_of = np.array(read_of); self.span = np.array(span); self.truth = truth
        self.link_begin = np.array(link_begin, dtype=np.int64); self.link_read_length = np.array(read_total, dtype=np.int64)
        self.read_bases = int(seq_off[-1]) + int(self.anchor_bases.sum())
        self.hap_start = hap_start; self.thread_nodes = threads
        self.anchor_off = np.array(a_off, dtype=np.uint64); self.anchor_length = np.array(a_len, dtype=np.uint32); self.anchor_node_offset = np.array(a_noff, dtype=np.uint32)
        self.anchor_path_off = np.array(a_poff, dtype=np.uint64); self.anchor_nodes = np.array(a_nodes, dtype=np.uint32)

    @classmethod
    def in_parallel(cls, n_reads, graph, seed=515, workers=8, chunk=500, **kw):
        """the same workload generated in chunks of `chunk` reads on `workers` forked processes (read r's chunk draws from seed * 4096 + chunk index:
        the result depends on `chunk`, not on `workers`) and laid behind each other — a chr22-scale leg makes 16 000 reads of 15 kbp, a minute of
        Python on one core"""
        import multiprocessing as mp
        jobs = [(min(chunk, n_reads - lo), seed * 4096 + k) for k, lo in enumerate(range(0, n_reads, chunk))]
        global _LR_GRAPH, _LR_KW
        _LR_GRAPH, _LR_KW = graph, kw
        if workers > 1 and len(jobs) > 1:
            with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
                parts = pool.map(_lr_chunk, jobs)
        else:
            parts = [_lr_chunk(j) for j in jobs]
        _LR_GRAPH = None
        w = object.__new__(cls)
        from . import capi
        first = cls(1, seed=seed, graph=graph, **kw)                   # (the graph-wide members: haplotype tables, threads)
        w.__dict__.update({k: v for k, v in first.__dict__.items() if k in ("graph", "lens", "threads", "hap_start", "thread_nodes")})
        links = np.concatenate([[0], np.cumsum([p["n"] for p in parts])]); reads = np.concatenate([[0], np.cumsum([p["n_reads"] for p in parts])])
        seq_off = np.concatenate([[0]] + [p["seq_off"][1:] + off for p, off in zip(parts, np.concatenate([[0], np.cumsum([p["seq_off"][-1] for p in parts])]))]).astype(np.int64)
        buf = np.concatenate([p["seqs"][:p["seq_off"][-1]] for p in parts])
        cat = lambda k: np.concatenate([p[k] for p in parts])
        n = int(links[-1])
        w.ws = capi.WfaSet(buf, seq_off, cat("mode"), cat("from_node"), cat("from_offset"), cat("to_node"), cat("to_offset"), path_cap=n * 24 + int(seq_off[-1]) // 4, edit_cap=n * 12)
        w.n = n; w.n_reads = int(reads[-1])
        w.read_of = np.concatenate([p["read_of"] + r0 for p, r0 in zip(parts, reads)]); w.span = cat("span"); w.truth = [t for p in parts for t in p["truth"]]
        w.link_begin = cat("link_begin"); w.link_read_length = cat("link_read_length"); w.anchor_bases = cat("anchor_bases")
        w.read_bases = int(seq_off[-1]) + int(w.anchor_bases.sum())
        a0 = np.concatenate([[0], np.cumsum([len(p["anchor_length"]) for p in parts])]); n0 = np.concatenate([[0], np.cumsum([len(p["anchor_nodes"]) for p in parts])])
        w.anchor_off = np.concatenate([[0]] + [p["anchor_off"][1:] + np.uint64(o) for p, o in zip(parts, a0)]).astype(np.uint64)
        w.anchor_path_off = np.concatenate([[0]] + [p["anchor_path_off"][1:] + np.uint64(o) for p, o in zip(parts, n0)]).astype(np.uint64)
        w.anchor_length = cat("anchor_length"); w.anchor_node_offset = cat("anchor_node_offset"); w.anchor_nodes = cat("anchor_nodes")
        return w

    def __getattr__(self, name):
        # a chr22-scale graph: the node strings and predecessor lists (1.7 M Python objects each) are made only when something asks for them
        if name == "nodes" and self.__dict__.get("graph") is not None:
            g = self.graph; buf = g.seq.tobytes(); col = g.col
            self.__dict__["nodes"] = [buf[int(col[v]):int(col[v + 1])].decode() for v in range(g.n_nodes)]
            return self.__dict__["nodes"]
        if name == "preds" and self.__dict__.get("graph") is not None:
            g = self.graph; po = g.pred_off.astype(np.int64); pi = g.pred_idx
            self.__dict__["preds"] = [[int(p) for p in pi[po[v]:po[v + 1]]] for v in range(g.n_nodes)]
            return self.__dict__["preds"]
        raise AttributeError(name)

    def prepare_connects(self):
        """every connect's graph between its anchors (between()) laid out once in one flat BandedSet: `connects`, `connect_row[i]` = problem
        i's row in it (-1: not a connect).  What a caller holds after vg's extract_connecting_graph; chain_stage picks its fallback batch
        out of it with BandedSet.select."""
        from . import capi
        mode = self.ws.array["mode"]
        idx = np.nonzero(mode == capi.WFA_CONNECT)[0]
        self.connects = capi.BandedSet.from_lists([self.between(int(i)) for i in idx])
        self.connect_row = np.full(len(mode), -1, dtype=np.int64); self.connect_row[idx] = np.arange(len(idx))

    def between(self, i):
        """the graph between problem i's anchors as a banded-global problem on the FORWARD strand: every node of the topological order from
        the node of the first flanking base to the node of the last, the outer two cut at the flanks; the sequence reverse-complemented
        when the read runs on the other strand (BandedGlobalAligner aligns end to end, so the strand is free to choose)"""
        cache = self.__dict__.setdefault("_between", {})          # (the extraction between two anchors is the caller's — vg's extract_connecting_graph; once per problem here)
        if i in cache:
            return cache[i]
        lo, hi, rev, h = self.truth[i]
        st, t = self.hap_start[h], self.thread_nodes[h]
        ka = int(np.searchsorted(st, lo - 1, side="right") - 1); kb = int(np.searchsorted(st, hi, side="right") - 1)
        va, vb = int(t[ka]), int(t[kb])
        oa = lo - 1 - int(st[ka]) + 1                                   # first base kept on the first node
        ob = hi - int(st[kb])                                           # bases kept on the last node
        nodes = []
        for v in range(va, vb + 1):
            s = self.nodes[v]
            if v == va and v == vb:
                s = s[oa:ob]
            elif v == va:
                s = s[oa:]
            elif v == vb:
                s = s[:ob]
            nodes.append(s)
        preds = [[p - va for p in self.preds[v] if va <= p] for v in range(va, vb + 1)]
        seq = self.ws.seqs[self.ws.seq_off[i]:self.ws.seq_off[i + 1]]
        if rev:
            seq = _comp_table()[seq[::-1]]
        cache[i] = dict(read=seq.tobytes().decode(), nodes=nodes, preds=preds, band_padding=int(np.sqrt(max(len(seq), 1))) + 1 + 64, permissive=True)
        return cache[i]


_LR_GRAPH = None; _LR_KW = {}


def _lr_chunk(job):
    """one chunk of LongReadWorkload.in_parallel (a forked worker: the graph is the parent's, copy-on-write) -> plain arrays"""
    n, seed = job
    w = LongReadWorkload(n, seed=seed, graph=_LR_GRAPH, **_LR_KW)
    a = w.ws.array
    return dict(n=w.n, n_reads=w.n_reads, seqs=w.ws.seqs, seq_off=np.asarray(w.ws.seq_off), mode=a["mode"].copy(), from_node=a["from_node"].copy(), from_offset=a["from_offset"].copy(),
                to_node=a["to_node"].copy(), to_offset=a["to_offset"].copy(), read_of=np.asarray(w.read_of), span=np.asarray(w.span), truth=w.truth, link_begin=w.link_begin,
                link_read_length=w.link_read_length, anchor_bases=w.anchor_bases, anchor_off=w.anchor_off, anchor_length=w.anchor_length, anchor_node_offset=w.anchor_node_offset,
                anchor_path_off=w.anchor_path_off, anchor_nodes=w.anchor_nodes)


class Config2Workload:
    """BASELINE.json configs[2] as a whole stage at its stated size: the chr22-scale SNP + indel graph of SURVEY §8(d) (VariationGraph: 50.8 Mbp,
    ~1.7 M nodes of <= 32 bp, two haplotypes that carry each variant with p = 0.5) and `n_reads` reads of 150 bp sampled from the two
    haplotypes on either strand, 1 % substitutions, `inserted_reads` of them with one extra base (a sequencing insertion: no gapless
    extension covers such a read, its cluster leaves tails for the X-drop stage).  Nothing but the bare reads is given to the engine:
    minimizer seeding, gapless extension, tail forests and the tails' X-drop alignments all run from the indexes of the graph.
    Reads come in batches of `batch` (flat uint8 arrays + offsets), generated once."""

    def __init__(self, n_reads, batch=1_000_000, seed=31, read_len=150, inserted_reads=0.1, graph=None):
        g = graph if graph is not None else VariationGraph()
        self.graph = g
        self.node_len = g.node_len; self.seq = g.seq
        self.threads = [(2 * np.nonzero(hap_pos >= 0)[0]).astype(np.uint32) for _, hap_pos in g.haps]
        rng = np.random.default_rng(seed)
        comp = _comp_table()
        self.batches = []
        col = np.arange(read_len)[None, :]
        for lo in range(0, n_reads, batch):
            n = min(batch, n_reads - lo)
            which = rng.integers(0, 2, n); rev = rng.random(n) < 0.5
            reads = np.empty((n, read_len), dtype=np.uint8)
            for h in (0, 1):
                sel = np.nonzero(which == h)[0]
                hseq = g.haps[h][0]
                a = rng.integers(0, len(hseq) - read_len, len(sel))
                fw = hseq[a[:, None] + col]
                rd = np.where(rev[sel][:, None], comp[fw[:, ::-1]], fw)
                reads[sel] = rd
            sub = rng.random(reads.shape) < 0.01
            reads[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
            if inserted_reads > 0:
                ins = np.nonzero(rng.random(n) < inserted_reads)[0]
                p = rng.integers(20, read_len - 20, len(ins))
                src = np.where(col > p[:, None], col - 1, col)                # read' = read[:p] + X + read[p:-1]
                shifted = np.take_along_axis(reads[ins], src, axis=1)
                shifted[np.arange(len(ins)), p] = ACGT[rng.integers(0, 4, len(ins))]
                reads[ins] = shifted
            self.batches.append((reads.ravel(), np.arange(n + 1, dtype=np.int64) * read_len))
        self.n = n_reads; self.read_len = read_len


class PairedWorkload:
    """A one-GPU slice of BASELINE.json configs[3]: `n_pairs` read pairs of 2 x 150 bp off the two haplotypes of a VariationGraph (the chr22-scale
    construction at `ref_len` bases), fragment lengths ~ N(mean, sd), either strand, 1 % substitutions; `hard` of the pairs have a second mate
    that no gapless extension will cover (an inserted stretch of 3-6 bases and 3 % substitutions) — the mate giraffe rescues from the mapped
    one's position (MinimizerMapper::attempt_rescue, src/minimizer_mapper.cpp:3264-3440).  Reads lie pair by pair (read 2 i, 2 i + 1): a
    shard of the stream keeps the two reads of a pair together (vg_amd/shard.py, group = 2)."""

    def __init__(self, n_pairs, ref_len=5_000_000, seed=41, read_len=150, mean=400.0, sd=40.0, hard=0.08, graph=None):
        g = graph if graph is not None else VariationGraph(ref_len=ref_len)
        self.graph = g; self.node_len = g.node_len; self.seq = g.seq
        self.threads = [(2 * np.nonzero(hap_pos >= 0)[0]).astype(np.uint32) for _, hap_pos in g.haps]
        rng = np.random.default_rng(seed)
        comp = _comp_table()
        n = n_pairs
        frag = np.clip(np.rint(rng.normal(mean, sd, n)), 2 * read_len // 2 + 20, mean + 4 * sd).astype(np.int64)
        which = rng.integers(0, 2, n); flip = rng.random(n) < 0.5
        col = np.arange(read_len)[None, :]
        left = np.empty((n, read_len), dtype=np.uint8); right = np.empty((n, read_len), dtype=np.uint8)
        start = np.zeros(n, dtype=np.int64)
        for h in (0, 1):
            sel = np.nonzero(which == h)[0]
            hseq = g.haps[h][0]
            a = rng.integers(0, len(hseq) - int(frag.max()) - 1, len(sel))
            start[sel] = a
            left[sel] = hseq[a[:, None] + col]
            right[sel] = hseq[(a + frag[sel] - read_len)[:, None] + col]
        right_rc = comp[right[:, ::-1]]; left_rc = comp[left[:, ::-1]]
        # the fragment as sequenced: mate 1 forward from its left end and mate 2 reverse from its right end, or the other way round
        m1 = np.where(flip[:, None], right_rc, left); m2 = np.where(flip[:, None], left, right_rc)
        reads = np.empty((2 * n, read_len), dtype=np.uint8); reads[0::2] = m1; reads[1::2] = m2
        sub = rng.random(reads.shape) < 0.01
        reads[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
        hard_pairs = np.nonzero(rng.random(n) < hard)[0]
        for i in hard_pairs:                                               # the second mate: an inserted stretch + more substitutions
            r = reads[2 * i + 1].copy()
            k = int(rng.integers(3, 7)); p = int(rng.integers(40, read_len - 40))
            r = np.concatenate([r[:p], ACGT[rng.integers(0, 4, k)], r[p:]])[:read_len]
            s = rng.random(read_len) < 0.03
            r[s] = ACGT[rng.integers(0, 4, int(s.sum()))]
            reads[2 * i + 1] = r
        self.reads = reads.ravel(); self.read_off = np.arange(2 * n + 1, dtype=np.int64) * read_len
        self.n_pairs = n; self.n = 2 * n; self.read_len = read_len; self.mean = mean; self.sd = sd
        self.truth = dict(hap=which, start=start, frag=frag, flip=flip, hard=hard_pairs)
        # successors as CSR over node indices (the rescue stage builds its subgraphs from them)
        dst = np.repeat(np.arange(g.n_nodes), np.diff(g.pred_off.astype(np.int64))); src = g.pred_idx.astype(np.int64)
        e = np.lexsort((dst, src))
        self.succ_off = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=g.n_nodes))]).astype(np.uint32)
        self.succ = dst[e].astype(np.uint32)

    def batch(self, a, b):
        """pairs [a, b) as a workload of their own (same graph; the reads are a view)"""
        w = object.__new__(PairedWorkload)
        w.__dict__.update(self.__dict__)
        L = self.read_len
        w.reads = self.reads[2 * a * L:2 * b * L]; w.read_off = self.read_off[:2 * (b - a) + 1]
        w.n_pairs = b - a; w.n = 2 * (b - a)
        t = self.truth; hard = t["hard"][(t["hard"] >= a) & (t["hard"] < b)] - a
        w.truth = dict(t, hap=t["hap"][a:b], start=t["start"][a:b], frag=t["frag"][a:b], flip=t["flip"][a:b], hard=hard)
        return w

    def subset(self, k):
        """the first k pairs as a workload of their own (same graph, same reads)"""
        w = object.__new__(PairedWorkload)
        w.__dict__.update(self.__dict__)
        w.reads = self.reads[:2 * k * self.read_len]; w.read_off = self.read_off[:2 * k + 1]
        w.n_pairs = k; w.n = 2 * k
        w.truth = dict(self.truth, hap=self.truth["hap"][:k], start=self.truth["start"][:k], frag=self.truth["frag"][:k], flip=self.truth["flip"][:k],
                       hard=self.truth["hard"][self.truth["hard"] < k])
        return w
