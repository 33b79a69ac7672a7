cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/stress
timeout -s KILL 900 python tools/stress.py 3 > gpurun_out/stress/stress_seed3.log 2>&1; echo "rc=$?" >> gpurun_out/stress/stress_seed3.log; tail -12 gpurun_out/stress/stress_seed3.log
# the WFA kernels once more with every problem on the wavefront kernel and most of them through its large tables (item filter on)
VGAMD_WFA_KERNEL=wave VGAMD_WFA_SMALL_POINTS=24 timeout -s KILL 600 python - > gpurun_out/stress/wfa_wave_large.log 2>&1 <<'PY'
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_wfa
ok, statuses = test_wfa.compare_engines(None, range(7000, 7080), n_problems=500)
print("wfa (wave form, small tables cut to 24 points): %d alignments identical; statuses %s" % (ok, statuses))
PY
echo "rc=$?" >> gpurun_out/stress/wfa_wave_large.log; tail -3 gpurun_out/stress/wfa_wave_large.log
