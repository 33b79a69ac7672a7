cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_banded.py tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_banded.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest_banded.log
VGAMD_BANDED_TIMING=1 timeout -s KILL 300 python bench.py --workload banded --steps 5 --warmup 2 --no-cpu > $O/bench_banded.json 2> $O/bench_banded.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_banded.json").read().strip().splitlines()[-1])
print(r["value"], r["config"]["end_to_end_from_host_buffers_alignments_per_s"], r["config"]["end_to_end_one_batch_alignments_per_s"], r["roofline"]["frac"])
PY
grep -n "prepare" $O/bench_banded.err | tail -8 | head -3; tail -42 $O/bench_banded.err | head -30
VGAMD_XBAND_TIMING=1 timeout -s KILL 200 python bench.py --workload xband --steps 8 --warmup 2 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err < /dev/null
timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print("xband", r["value"], r["ms_per_step"])
PY
grep -E "check|pack" $O/bench_xband.err | tail -5
