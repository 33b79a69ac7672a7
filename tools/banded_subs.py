#!/usr/bin/env python3
"""vgk_banded_align from host buffers (100 000 problems of the banded bench) with the call cut into 4 / 6 / 8 / 12 sub-batches (VGAMD_BANDED_SUBS):
builder-run on the GPU box, prints alignments/s each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vg_amd import capi, workloads   # noqa: E402
wl = workloads.BandedWorkload(100000, seed=99)
eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 0))
for subs in (4, 6, 8, 12):
    os.environ["VGAMD_BANDED_SUBS"] = str(subs)
    eng.banded_align(wl.bs); eng.banded_align(wl.bs)
    t = time.perf_counter()
    for _ in range(5):
        eng.banded_align(wl.bs)
    t = (time.perf_counter() - t) / 5
    print("sub-batches %2d: %.2f M alignments/s (%.2f ms)" % (subs, 100000 / t / 1e6, t * 1e3))
