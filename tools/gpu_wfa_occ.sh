cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfaocc
run() { tag=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/wfaocc/$tag.json 2> gpurun_out/wfaocc/$tag.err
  python -c "
import json
d=json.loads(open('gpurun_out/wfaocc/$tag.json').read().strip().splitlines()[-1]); c=d['config']; print('$tag', 'reads/s', round(d['value']), 'step ms', round(d['ms_per_step'],1), 'wfa kernel ms', round(c.get('wfa_kernel_ms',0),1), c.get('wfa_launches'), c.get('links'))"
}
run base12 A=1
run occ4_12 VGAMD_ENGINE_LIB=$PWD/build/variants/libvgamd_occ4.so VGAMD_WFA_WAVES_PER_CU=12
run occ4_16 VGAMD_ENGINE_LIB=$PWD/build/variants/libvgamd_occ4.so VGAMD_WFA_WAVES_PER_CU=16
