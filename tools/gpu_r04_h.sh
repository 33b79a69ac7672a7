# round 4, GPU call H: TB_REWALK with checkpoints every 16 columns and the miss list — parity, kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_tb_rewalk.py tests/test_windows.py tests/test_giraffe_stage.py -m gpu -q -x > $O/pytest_h.log 2>&1; echo "pytest rc=$?" >> $O/pytest_h.log; tail -3 $O/pytest_h.log
export VGAMD_TB_REWALK=1
timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 --cpu-sample 100000 > $O/bench_rewalk.json 2> $O/bench_rewalk.err
python3 -c "
import json
d=json.loads(open('$O/bench_rewalk.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('rewalk', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv,glob
for r in csv.DictReader(open(glob.glob('gpurun_out/r04h/prof/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('band','fill_kernel','rewalk')): print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3),'ms')
PY
