cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; mkdir -p $O
for mode in pipe one; do
  unset VGAMD_XBAND_ONE_BATCH; [ $mode = one ] && export VGAMD_XBAND_ONE_BATCH=1
  VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 8 --warmup 2 --no-cpu > $O/bench_xband_$mode.json 2> $O/bench_xband_$mode.err; echo "bench rc=$?"
  python - <<PY
import json
r=json.loads(open("$O/bench_xband_$mode.json").read().strip().splitlines()[-1])
print("$mode", r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r["roofline"]["frac"])
PY
  grep check $O/bench_xband_$mode.err | tail -1
done
unset VGAMD_XBAND_ONE_BATCH
timeout -s KILL 300 python bench.py --workload xband --steps 8 --warmup 2 > $O/bench_xband_cpu.json 2> /dev/null
python - <<PY
import json
r=json.loads(open("$O/bench_xband_cpu.json").read().strip().splitlines()[-1])
print(r["value"], r["parity"], r["cpu_baseline"])
PY
