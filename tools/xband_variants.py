"""vgk_xdrop_band_align on the xband bench's tails with several builds of the engine library (tools/build_variant.sh): kernel time and
host-inclusive time per build, results compared with the first build's."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vg_amd import capi, workloads

n = int(sys.argv[1]); libs = sys.argv[2:]
ps = workloads.TailWorkload(n, seed=77).ps
first = None
for lib in libs:
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib if lib != "default" else None)
    eng.lib.vgk_xdrop_band_last_ms.restype = ctypes.c_double; eng.lib.vgk_xdrop_band_last_ms.argtypes = [ctypes.c_void_p]
    eng.xdrop_band_align(ps); eng.xdrop_band_align(ps)
    best, k = 1e9, 1e9
    for _ in range(3):
        t = time.perf_counter(); res, ops, st = eng.xdrop_band_align(ps); best = min(best, time.perf_counter() - t)
        k = min(k, eng.lib.vgk_xdrop_band_last_ms(eng.h))
    key = (res["score"].tobytes(), res["n_ops"].tobytes(), res["end_node"].tobytes(), ops.tobytes())
    if first is None: first = key
    print("%-40s kernel %.2f ms, call %.2f ms (%.2f M tails/s), same results as the first build: %s" % (lib, k, best * 1e3, n / best / 1e6, key == first), flush=True)
