cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tiled
timeout -s KILL 900 python -m pytest tests/test_gssw_gpu_parity.py tests/test_windows.py tests/test_giraffe_stage.py tests/test_tail_forest.py -m gpu -x -q 2>&1 | tail -3
VARIANTS="base untiled" bash tools/gpu_walk_experiment.sh
timeout -s KILL 400 python bench.py --steps 10 --warmup 3 > gpurun_out/tiled/bench.json 2> gpurun_out/tiled/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/tiled/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['one_stream'], d['config'].get('two_lanes'), d['parity'], d['roofline']['frac'])"
