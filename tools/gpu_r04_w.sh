# round 4: the headline after the walks' profile words moved into registers and the second fill's column cut — gssw parity tests, bench, kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gssw_gpu_parity.py -m gpu -q -x > $O/pytest_gssw.log 2>&1 < /dev/null; tail -2 $O/pytest_gssw.log
timeout -s KILL 200 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary > $O/bench_headline.json 2> $O/bench_headline.err < /dev/null; echo "bench rc=$?"
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1 > $O/stats.log 2>&1 ) < /dev/null
python3 - <<'PY'
import json, glob, csv, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04w'
d = json.loads(open(O + '/bench_headline.json').read().strip().split('\n')[-1])
o = d['config']['one_stream']; print('headline %.2f M reads/s fill %.2f tail %.2f second fill %.2f step %.2f ms frac %.3f parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o.get('second_fill_ms', 0), o['ms_per_step'], d['roofline']['frac'], d['parity']))
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
