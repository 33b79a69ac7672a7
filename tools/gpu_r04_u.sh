cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_seed_policy.py tests/test_minimizer.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
export VGAMD_CONFIG2_ONE_CONTEXT=1
for pol in 0 1; do
  unset VGAMD_CONFIG2_POLICY; [ $pol = 1 ] && export VGAMD_CONFIG2_POLICY=1
  timeout -s KILL 300 python bench.py --workload config2 --reads 2000000 --steps 2 --warmup 1 --cpu-sample 20000 > $O/bench_config2_policy$pol.json 2> $O/bench_config2_policy$pol.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_config2_policy$pol.json").read().strip().splitlines()[-1])
c=r["config"]; print("policy $pol", r["value"], r["ms_per_step"], c.get("kernel_ms_per_batch"), r["parity"], {k: c["totals"][k] for k in ("seeds","ext","tails")} if "totals" in c else "")
PY
done
