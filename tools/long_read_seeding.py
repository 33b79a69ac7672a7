#!/usr/bin/env python3
"""Reads of 15 kbp through vgk_minimizer_list -> find_seeds' choice (host shim) -> vgk_minimizer_seeds_of on the chr22-scale graph: seconds per stage and
what the choice leaves (tools/gpu_r06.sh long_seeds).  Not a bench line: the anchors giraffe chains come from seeds like these, the chaining is out of scope."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vg_amd import capi, pipeline, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = 15000
g = workloads.VariationGraph(ref_len=int(os.environ.get("VGAMD_LONGREAD_REF_LEN", "50818468")))
rng = np.random.default_rng(7)
comp = workloads._comp_table()
reads = np.empty((n, L), dtype=np.uint8)
for i in range(n):
    hseq = g.haps[int(rng.integers(0, 2))][0]; a = int(rng.integers(0, len(hseq) - L)); r = hseq[a:a + L]
    reads[i] = comp[r[::-1]] if rng.random() < 0.5 else r
sub = rng.random(reads.shape) < 0.005
reads[sub] = workloads.ACGT[rng.integers(0, 4, int(sub.sum()))]
off = np.arange(n + 1, dtype=np.uint64) * L
eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5))
threads = [(2 * np.nonzero(hap_pos >= 0)[0]).astype(np.uint32) for _, hap_pos in g.haps]
t = time.perf_counter(); mi = eng.minimizer_index((g.node_len, g.seq), threads); t_index = time.perf_counter() - t
k = 29
flat = reads.ravel()
pipeline.seed_long_reads(eng, mi, flat[:L * 8], off[:9], k)          # warm
t0 = time.perf_counter(); moff, recs = eng.minimizer_list(mi, flat, off); t1 = time.perf_counter()
out = pipeline.seed_long_reads(eng, mi, flat, off, k); t2 = time.perf_counter()
print(json.dumps({"reads": n, "read_length": L, "minimizers_per_read": len(recs) / n, "taken_per_read": float(out["take"].sum()) / n, "seeds_per_read": len(out["seeds"]) / n,
                  "seconds": {"minimizer index": t_index, "vgk_minimizer_list": t1 - t0, "list + choice (host threads) + vgk_minimizer_seeds_of": t2 - t1},
                  "reads_per_s_whole": n / (t2 - t1)}))
