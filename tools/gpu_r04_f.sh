# round 4, GPU call F: TB_REWALK after the boundary rows went lane-major (LDS-staged in the fill) — parity + per-kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04f; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_tb_rewalk.py tests/test_windows.py -m gpu -q -x > $O/pytest_f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f.log; tail -3 $O/pytest_f.log
export VGAMD_TB_REWALK=1
timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 --cpu-sample 100000 > $O/bench_rewalk.json 2> $O/bench_rewalk.err
python3 -c "
import json
d=json.loads(open('$O/bench_rewalk.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('rewalk', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-150
