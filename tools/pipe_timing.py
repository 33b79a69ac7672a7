"""Per-stage timing of the double-buffered loop of bench.py (run on a GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
from vg_amd import capi, workloads
eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5))
wl = workloads.LinearWorkload(1000000, seed=43)
def timed_pack():
    t = time.perf_counter(); b = eng.pack(wl, 48); return b, 1e3 * (time.perf_counter() - t)
with ThreadPoolExecutor(1) as ex:
    nxt = ex.submit(timed_pack)
    for k in range(10):
        t0 = time.perf_counter(); pb, tpack = nxt.result(); t1 = time.perf_counter()
        if k + 1 < 10: nxt = ex.submit(timed_pack)
        pb.run(); pb.sync(); t2 = time.perf_counter()
        pb.fetch(); t3 = time.perf_counter(); pb.free(); t4 = time.perf_counter()
        print("k=%d wait %.1f (pack took %.1f) run %.1f fetch %.1f free %.1f" % (k, 1e3*(t1-t0), tpack, 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
