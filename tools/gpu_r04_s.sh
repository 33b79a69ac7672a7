cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $O
timeout -s KILL 500 python bench.py --workload paired --steps 3 --warmup 1 --cpu-sample 200000 > $O/bench_paired_100k_pairs.json 2> $O/bench_paired.err < /dev/null; echo "rc=$?"
timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_paired_100k_pairs.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["parity"], r["cpu_baseline"])
PY
