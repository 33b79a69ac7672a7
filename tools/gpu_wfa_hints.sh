cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfahints
timeout -s KILL 400 python -m pytest tests/test_longread_stage.py tests/test_wfa.py -m gpu -x -q 2>&1 | tail -2
for k in 1 2; do
timeout -s KILL 400 python bench.py --workload longread --steps 5 --warmup 2 > gpurun_out/wfahints/longread_$k.json 2> gpurun_out/wfahints/longread_$k.err
python -c "
import json
d=json.loads(open('gpurun_out/wfahints/longread_$k.json').read().strip().splitlines()[-1]); c=d['config']; print('reads/s', round(d['value']), 'step ms', round(d['ms_per_step'],1), 'wfa kernel ms', round(c.get('wfa_kernel_ms',0),1), c.get('stage_ms'), d['parity']['identical'], d['parity'].get('identical_with_point_budgets'))"
done
