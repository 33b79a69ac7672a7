# round 4: the X-drop band call after its host-side passes were chunked — parity tests, the bench line, the call's laps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04x; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_xdrop_band.py -m gpu -q -x > $O/pytest_xband.log 2>&1 < /dev/null; tail -2 $O/pytest_xband.log
VGAMD_XBAND_TIMING=1 timeout -s KILL 200 python3 bench.py --workload xband --no-cpu --no-secondary --steps 5 --warmup 2 > $O/bench_xband.json 2> $O/bench_xband.err < /dev/null; echo "bench rc=$?"
grep "vgk_xdrop_band_align" $O/bench_xband.err | tail -18
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04x'
d = json.loads(open(O + '/bench_xband.json').read().strip().split('\n')[-1])
print('xband %.2f M tails/s, %.2f ms/step, parity %s' % (d['value']/1e6, d['ms_per_step'], d.get('parity')))
PY
