cd $GRAFT_REPO_ROOT; O=gpurun_out/r04q; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_banded.py -x -q -m gpu > $O/pytest_banded.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_banded.log
VGAMD_BANDED_TIMING=1 timeout -s KILL 300 python bench.py --workload banded --steps 5 --warmup 2 --no-cpu > $O/bench_banded.json 2> $O/bench_banded.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_banded.json").read().strip().splitlines()[-1])
print(r["value"], r["config"]["end_to_end_from_host_buffers_alignments_per_s"], r["config"]["end_to_end_one_batch_alignments_per_s"], r["roofline"]["frac"])
PY
grep -n "prepare" $O/bench_banded.err | tail -8 | head -3; tail -42 $O/bench_banded.err | head -30
