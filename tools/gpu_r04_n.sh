cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
for v in default; do
  unset VGAMD_ENGINE_LIB; [ $v != default ] && export VGAMD_ENGINE_LIB=$PWD/build/variants/libvgamd_$v.so
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/prof_$v.log 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_xband_$v.csv; echo $v; grep xdrop_band $O/kernel_stats_xband_$v.csv | cut -c1-120
  rm -rf $O/prof_$v
done
unset VGAMD_ENGINE_LIB
VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r["band"])
PY
tail -6 $O/bench_xband.err
