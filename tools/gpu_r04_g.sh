# round 4: the banded call with its geometry on the device — parity tests, the bench line, the call's laps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_banded.py -m gpu -q -x > $O/pytest_banded.log 2>&1 < /dev/null; tail -3 $O/pytest_banded.log
timeout -s KILL 200 python3 bench.py --workload banded --no-cpu --no-secondary --steps 5 --warmup 2 > $O/bench_banded.json 2> $O/bench_banded.err < /dev/null; echo "bench rc=$?"; tail -3 $O/bench_banded.err
VGAMD_BANDED_TIMING=1 timeout -s KILL 200 python3 bench.py --workload banded --no-cpu --no-secondary --steps 1 --warmup 0 > /dev/null 2> $O/bench_banded_laps.err < /dev/null
grep "device geometry" $O/bench_banded_laps.err | tail -22
timeout -s KILL 120 python3 tools/banded_subs.py 2> /dev/null < /dev/null | tee $O/banded_subs.txt
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04g'
d = json.loads(open(O + '/bench_banded.json').read().strip().split('\n')[-1]); c = d['config']
print('banded resident %.2f M/s; from host buffers: device geometry %.2f M/s, host geometry %.2f M/s, one batch %.2f M/s' % (d['value']/1e6, c['end_to_end_from_host_buffers_alignments_per_s']/1e6, c['end_to_end_host_geometry_alignments_per_s']/1e6, c['end_to_end_one_batch_alignments_per_s']/1e6))
PY
