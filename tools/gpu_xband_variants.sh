# the X-drop band path: parity on the GPU, kernel and host-inclusive time (tools/xband_variants.py), the host laps of one call, the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/xband
timeout -s KILL 300 python -m pytest tests/test_xdrop_band.py -m gpu -x -q 2>&1 | tail -2
VGAMD_XBAND_TIMING=1 timeout -s KILL 500 python tools/xband_variants.py 200000 default > gpurun_out/xband/variants.txt 2> gpurun_out/xband/variants.err
cat gpurun_out/xband/variants.txt; grep vgk_xdrop_band_align gpurun_out/xband/variants.err | tail -7
timeout -s KILL 300 python bench.py --workload xband > gpurun_out/xband/bench_xband_200k.json 2> gpurun_out/xband/bench_xband.err
python -c "
import json
d=json.loads(open('gpurun_out/xband/bench_xband_200k.json').read().strip().splitlines()[-1]); print(round(d['value']), d['band']['fill_kernel_ms'], d['band']['tails_whose_band_mode_alignment_differs_from_exact_mode'], d['parity'], d['roofline']['frac'])"
