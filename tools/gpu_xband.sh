cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/xband
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py tests/test_gssw_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout -s KILL 400 python bench.py --workload xband > gpurun_out/xband/xband.json 2> gpurun_out/xband/xband.err
python -c "
import json
d=json.loads(open('gpurun_out/xband/xband.json').read().strip().splitlines()[-1]); print(round(d['value']), d['unit'], d['band'], d['parity'])"
