import sys, time, os
sys.path.insert(0, os.getcwd())
from vg_amd import capi, workloads
eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5))
wl = workloads.LinearWorkload(1000000, seed=43)
for it in range(4):
    t0 = time.perf_counter(); b = eng.pack(wl, 48); t1 = time.perf_counter()
    b.run(); b.sync(); t2 = time.perf_counter()
    r, o = b.fetch(); t3 = time.perf_counter()
    b.free(); t4 = time.perf_counter()
    print("pack %.1f run %.1f fetch %.1f free %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)))
