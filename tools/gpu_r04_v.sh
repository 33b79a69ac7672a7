cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $O
for c in 2 3 4; do
  export VGAMD_CONFIG2_CONTEXTS=$c
  timeout -s KILL 300 python bench.py --workload config2 --reads 12000000 --steps 3 --warmup 1 --no-cpu > $O/bench_config2_ctx$c.json 2> $O/bench_config2_ctx$c.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_config2_ctx$c.json").read().strip().splitlines()[-1])
print("contexts $c", r["value"], r["ms_per_step"], r["config"].get("ms_per_batch"), (r["config"].get("one_context") or {}).get("ms_per_batch"))
PY
done
