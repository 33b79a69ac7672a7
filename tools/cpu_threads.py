"""How many host threads does this box really give us?  Prints the CPU limits the container runs under and times the CPU baseline
(oracle/vgo_gssw_fast.c) at several OMP thread counts, each in its own process (OpenMP reads OMP_NUM_THREADS at start-up)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, %r)
from vg_amd import capi, workloads
wl = workloads.LinearWorkload(int(sys.argv[1]), ref_len=1_000_000)
eng = capi.Engine(lib=os.path.join(%r, "oracle", "libvgoracle.so"))
eng.lib.vgo_gssw_run_fast.argtypes = [ctypes.c_void_p]
with eng.pack(wl, 48) as b:
    eng.lib.vgo_gssw_run_fast(b.h)
    t = time.perf_counter(); eng.lib.vgo_gssw_run_fast(b.h); t = time.perf_counter() - t
print("threads=%%s reads/s=%%.0f GCUPS=%%.1f" %% (os.environ.get("OMP_NUM_THREADS"), wl.n / t, wl.cells() / t / 1e9))
""" % (ROOT, ROOT)

if __name__ == "__main__":
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)|MHz' ; uptime")
    n = sys.argv[1] if len(sys.argv) > 1 else "200000"
    for threads in (8, 16, 32, 64, 128, 256):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
        subprocess.run([sys.executable, "-c", CHILD, n], env=env)
