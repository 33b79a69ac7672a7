"""vgk_banded_align from host buffers with different numbers of host threads: where the host half (band geometry, arenas, results)
stops scaling.  Prints the VGAMD_BANDED_TIMING laps of the last of three calls per thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vg_amd import capi, workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
wl = workloads.BandedWorkload(n)
eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5))
eng.banded_align(wl.bs)
for threads in (4, 8, 16, 32, 48, 96):
    os.environ["VGAMD_HOST_THREADS"] = str(threads)
    best = 1e9
    for k in range(3):
        if k == 2:
            os.environ["VGAMD_BANDED_TIMING"] = "1"
            print("[threads %d]" % threads, file=sys.stderr, flush=True)
        t = time.perf_counter(); eng.banded_align(wl.bs); best = min(best, time.perf_counter() - t)
    os.environ.pop("VGAMD_BANDED_TIMING", None)
    print("threads %3d: %.2f ms per call, %.2f M alignments/s from host buffers" % (threads, best * 1e3, n / best / 1e6), flush=True)
