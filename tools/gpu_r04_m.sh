cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"))
PY
tail -6 $O/bench_xband.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_xband.csv; head -8 $O/kernel_stats_xband.csv | cut -c1-160
rm -rf $O/prof
