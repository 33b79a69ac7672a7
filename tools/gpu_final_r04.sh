# round 4 closing run: the whole -m gpu suite, smoke(), and the driver's bench command with its secondary records
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04_final; mkdir -p $O
timeout -s KILL 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
t0=$(date +%s)
timeout -s KILL 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_run.json 2> $O/bench_default_run.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_final/bench_default_run.json').read().strip().split('\n')[-1])
o=d['config']['one_stream']; print('headline %.2f M reads/s fill %.2f walk %.2f step %.2f ms frac %.3f parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['roofline']['frac'], d['parity']))
for r in d.get('secondary', []):
    print(r['workload'], r.get('error') or ('%.3g %s, %.2f ms/step, frac %s, parity %s, wall %s s' % (r['value'], r['unit'], r['ms_per_step'], (r.get('roofline') or {}).get('frac'), {k: v for k, v in (r.get('parity') or {}).items() if k in ('checked', 'identical')}, r['wall_s'])))
PY
