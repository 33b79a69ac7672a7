# round 4, GPU call E: TB_REWALK with the band — parity on the MI355X, then the headline batch in both modes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_tb_rewalk.py tests/test_gssw_gpu_parity.py tests/test_windows.py tests/test_reference_tap.py tests/test_giraffe_stage.py -m gpu -q -x > $O/pytest_e.log 2>&1; echo "pytest rc=$?" >> $O/pytest_e.log; tail -5 $O/pytest_e.log
for mode in rewalk codes; do
  if [ $mode = codes ]; then export VGAMD_TB_CODES=1; else unset VGAMD_TB_CODES; fi
  timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 > $O/bench_$mode.json 2> $O/bench_$mode.err
  python3 -c "
import json
d=json.loads(open('$O/bench_$mode.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('$mode', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
done
unset VGAMD_TB_CODES
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
