"""Large randomized differential run of every kernel against the oracle (run on a GPU box: python tools/stress.py [seed]).
Not part of the pytest suite: it takes a few minutes and exists to shake out rare divergences."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import gen
import test_banded
import test_gapless
from vg_amd import capi

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ORA = os.path.join(ROOT, "oracle", "libvgoracle.so")
rng = np.random.default_rng(seed)
fails = 0

# gssw local / pinned / xdrop, several scorings
for sc in (capi.Scoring.simple(1, 4, 6, 1, 5), capi.Scoring.simple(2, 3, 5, 2, 7), capi.Scoring.simple(1, 1, 1, 1, 0), capi.Scoring.simple(3, 5, 9, 2, 10)):
    for mode in (None, capi.VGK_XDROP_PINNED):
        problems = [gen.random_problem(rng, max_nodes=14, max_node_len=40, max_read=int(rng.choice([60, 150, 400, 1000])), mode=mode, with_n=0.05)
                    for _ in range(6000)]
        ps = gen.problem_set(problems)
        t = time.time()
        try:
            rg, og = capi.Engine(sc).align(ps)
        except capi.VgkError as e:          # scoring / length outside the engine's documented range: refused, never wrong
            print("gssw mode=%s scoring=%s: refused (%s)" % (mode, list(sc.matrix[:2]), e)); continue
        ro, oo = capi.Engine(sc, lib=ORA).align(ps)
        bad = 0
        for i in range(ps.n):
            if rg["status"][i] != ro["status"][i] or rg["score"][i] != ro["score"][i] or (rg["score"][i] > 0 and capi.cigar_string(rg[i], og) != capi.cigar_string(ro[i], oo)):
                bad += 1
        print("gssw mode=%s scoring=%s: %d problems, %d differ (%.1fs)" % (mode, list(sc.matrix[:2]), ps.n, bad, time.time() - t)); fails += bad

# banded, all rows-per-lane classes, plain and quality-adjusted
import qualadj
for qa in (None, qualadj.qual_adj_tables()):
    problems = test_banded.random_banded_set(seed + 7, 8000) + test_banded.mixed_band_problems(seed + 8, 1500, 20, 240, max_read=700, max_node_len=60)
    if qa is not None:
        for p in problems:
            p["qual"] = rng.integers(0, 41, len(p["read"])).astype(np.uint8)
    bs = capi.BandedSet.from_lists(problems)
    got = capi.Engine(qual_adj=qa).banded_align(bs); ref = capi.Engine(lib=ORA, qual_adj=qa).banded_align(bs)
    bad = [b for b in test_banded._same(problems, ref, got) if b[3]["status"] != -7]
    print("banded qual_adj=%s: %d problems, %d differ, %d refused (band > 2048)" % (qa is not None, len(problems), len(bad), int((got[0]["status"] == -7).sum()))); fails += len(bad)

# gapless
total, full = test_gapless.compare_engines(None, range(seed * 1000, seed * 1000 + 150), n_reads=600)
print("gapless: %d extensions over 90000 reads identical (%d full-length sets)" % (total, full))
# WFA (incl. cyclic threads, custom error models) and the k-best pinned tracebacks
import test_wfa
import test_pinned_multi
ok, statuses = test_wfa.compare_engines(None, range(seed * 1000, seed * 1000 + 120), n_problems=500)
print("wfa: %d alignments identical; statuses %s" % (ok, statuses))
total = test_pinned_multi.compare_engines(None, range(seed * 1000, seed * 1000 + 60), n_problems=150, max_alt=40)
print("pinned k-best: %d alternates identical, none twice" % total)
print("FAILURES", fails)
sys.exit(1 if fails else 0)
