cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xband.log
for mode in pk 16 32; do
  unset VGAMD_XBAND_CELLS32 VGAMD_XBAND_ARITH32; [ $mode = 32 ] && export VGAMD_XBAND_CELLS32=1; [ $mode = 16 ] && export VGAMD_XBAND_ARITH32=1
  VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband_$mode.json 2> $O/bench_xband_$mode.err; echo "bench$mode rc=$?"
  python - <<PY
import json
r=json.loads(open("$O/bench_xband_$mode.json").read().strip().splitlines()[-1])
print("$mode", r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r.get("parity"))
PY
  tail -6 $O/bench_xband_$mode.err
done
