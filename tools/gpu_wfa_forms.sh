# the long-read stage's WFA call in the three kernel forms (and hand-over thresholds of the hybrid)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfaforms
run() {
  tag=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/wfaforms/$tag.json 2> gpurun_out/wfaforms/$tag.err
  python -c "
import json
d=json.loads(open('gpurun_out/wfaforms/$tag.json').read().strip().splitlines()[-1]); c=d['config']; print('$tag', 'reads/s', round(d['value']), 'step ms', round(d['ms_per_step'],1), 'wfa kernel ms', round(c.get('wfa_kernel_ms',0),1), c.get('wfa_launches'), c.get('links'), d.get('parity'))"
}
run wave VGAMD_WFA_KERNEL=wave
run hybrid128 VGAMD_WFA_KERNEL=hybrid
run hybrid64 VGAMD_WFA_KERNEL=hybrid VGAMD_WFA_HAND_OVER_POINTS=64
run hybrid256 VGAMD_WFA_KERNEL=hybrid VGAMD_WFA_HAND_OVER_POINTS=256
run hybrid512 VGAMD_WFA_KERNEL=hybrid VGAMD_WFA_HAND_OVER_POINTS=512
