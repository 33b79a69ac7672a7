# WFA point budget on the GPU box: the long-read stage and the loose-problem WFA bench under a few budgets
cd $GRAFT_REPO_ROOT
for b in 0 512 256 128 64; do
  VGAMD_WFA_POINT_BUDGET=$b timeout 300 python bench.py --workload longread --reads 2000 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('longread budget $b: reads/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'wfa kernel', round(c['wfa_kernel_ms'],1), 'fallbacks', c['fallbacks'], 'parity', d['parity']['identical'], '/', d['parity']['checked'], 'declined tails', d['parity']['reads_with_a_declined_tail'], 'stage', {k: round(v,1) for k,v in c['stage_ms'].items()})"
done
for b in 0 256 128; do
  VGAMD_WFA_POINT_BUDGET=$b timeout 200 python bench.py --workload wfa --steps 5 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wfa 500k budget $b: ms/launch', round(d['roofline']['avg_launch_ms'],2), 'failed', d.get('problems_failed'))"
done
