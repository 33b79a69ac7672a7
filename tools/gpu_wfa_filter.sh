cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfafilter
timeout -s KILL 600 python -m pytest tests/test_wfa.py tests/test_longread_stage.py -m gpu -x -q 2>&1 | tail -3
for mode in filter nofilter; do
  if [ $mode = nofilter ]; then export VGAMD_WFA_NO_FILTER=1; else unset VGAMD_WFA_NO_FILTER; fi
  VGAMD_WFA_STATS=1 timeout -s KILL 300 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/wfafilter/longread_$mode.json 2> gpurun_out/wfafilter/longread_$mode.err
  grep "wfa wave" gpurun_out/wfafilter/longread_$mode.err | tail -1 | cut -c1-600
  python -c "
import json
d=json.loads(open('gpurun_out/wfafilter/longread_$mode.json').read().strip().splitlines()[-1]); c=d['config']; print('$mode', 'reads/s', round(d['value']), 'step ms', round(d['ms_per_step'],1), 'wfa kernel ms', round(c.get('wfa_kernel_ms',0),1), c.get('stage_ms'))"
  timeout -s KILL 300 python bench.py --workload wfa --steps 5 --warmup 2 --no-cpu > gpurun_out/wfafilter/wfa_$mode.json 2> gpurun_out/wfafilter/wfa_$mode.err
  python -c "
import json
d=json.loads(open('gpurun_out/wfafilter/wfa_$mode.json').read().strip().splitlines()[-1]); print('$mode wfa 500k ms', round(d['ms_per_step'],2), d['roofline'].get('avg_launch_ms'))"
done
