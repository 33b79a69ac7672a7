# the WFA launch of the long-read stage against the number of resident wavefronts per CU (is the heavy problems' critical path bound by
# latency under load or by the dependent chain itself?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfawaves
for w in ${WAVES:-1 2 4 8 12}; do
  VGAMD_WFA_WAVES_PER_CU=$w timeout -s KILL 240 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/wfawaves/w$w.json 2> gpurun_out/wfawaves/w$w.err
  python -c "
import json
d=json.loads(open('gpurun_out/wfawaves/w$w.json').read().strip().splitlines()[-1]); c=d['config']; print('waves/CU', $w, 'reads/s', round(d['value']), 'step ms', round(d['ms_per_step'],1), 'wfa kernel ms', round(c.get('wfa_kernel_ms',0),1), c.get('wfa_launches'))"
done
