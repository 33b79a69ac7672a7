cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gssw_gpu_parity.py tests/test_reference_tap.py tests/test_chain_alignment.py tests/test_giraffe_stage.py tests/test_alignment_batch.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in spec two one; do
  unset VGAMD_WALK_ONE_PASS VGAMD_NO_SPEC_FILL; [ $v = one ] && export VGAMD_WALK_ONE_PASS=1; [ $v = two ] && export VGAMD_NO_SPEC_FILL=1
  timeout -s KILL 200 python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary > $O/bench_$v.json 2> $O/bench_$v.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1]); o=d["config"]["one_stream"]
print("$v headline %.2f M reads/s fill %.2f walk %.2f step %.2f ms parity %s e2e %s" % (d["value"]/1e6, o["fill_ms"], o["traceback_ms"], o["ms_per_step"], d["parity"], d.get("end_to_end_double_buffered_per_s")))
PY
done
unset VGAMD_WALK_ONE_PASS
B="python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1"
( cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1 ) < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gssw_walk\|gssw_fill" "$f" < /dev/null | cut -c1-120
