cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03j
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03j/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03j/pytest_gpu.log
tail -5 gpurun_out/r03j/pytest_gpu.log
timeout -s KILL 300 python bench.py --workload xband > gpurun_out/r03j/xband.json 2> gpurun_out/r03j/xband.err
python -c "
import json
d=json.loads(open('gpurun_out/r03j/xband.json').read().strip().splitlines()[-1]); print(d['value'], d['band'], d['parity'])"
