# round 4, last GPU call: the first-pass traceback across nodes with several predecessors — gssw and window parity on the MI355X, the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04last; mkdir -p $O
timeout -s KILL 55 python -m pytest tests/test_gssw_gpu_parity.py tests/test_windows.py -m gpu -q -x > $O/pytest.log 2>&1 < /dev/null; tail -2 $O/pytest.log
timeout -s KILL 40 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-secondary --no-cpu > $O/bench_headline.json 2> $O/bench_headline.err < /dev/null; echo "bench rc=$?"
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04last'
d = json.loads(open(O + '/bench_headline.json').read().strip().split('\n')[-1]); o = d['config']['one_stream']
print('headline %.2f M reads/s fill %.2f tail %.2f step %.2f ms parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity']))
PY
