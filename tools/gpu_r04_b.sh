# round 4, GPU call B: parity of what changed (wide route, minimizer lookups, gapless prefetch, gbwt checks), then the default bench run
# with its secondary records (the driver's command), timed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gssw_wide.py tests/test_chain_alignment.py tests/test_gbwt_file.py tests/test_minimizer.py tests/test_gapless.py tests/test_giraffe_stage.py tests/test_gssw_gpu_parity.py -m gpu -q -x > $O/pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_b.log; tail -5 $O/pytest_b.log
t0=$(date +%s)
timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b/bench_default.json').read().strip().split('\n')[-1])
o=d['config']['one_stream']; print('headline %.2f M reads/s fill %.2f walk %.2f step %.2f ms parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity']))
for r in d.get('secondary', []):
    print(r['workload'], r.get('error') or ('%.3g %s, %.2f ms/step, frac %s, parity %s, wall %s s' % (r['value'], r['unit'], r['ms_per_step'], (r.get('roofline') or {}).get('frac'), {k: v for k, v in (r.get('parity') or {}).items() if k in ('checked', 'identical')}, r['wall_s'])))
    if r['workload'] == 'config2': print('   ', json.dumps(r['config'].get('kernel_ms_per_batch')), r['config'].get('ms_per_batch'))
PY
