# where the host-inclusive calls spend their time (banded: VGAMD_BANDED_TIMING laps; gapless: VGAMD_TIMING)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/e2e
VGAMD_BANDED_TIMING=1 timeout -s KILL 300 python bench.py --workload banded --steps 3 --warmup 1 --no-cpu > gpurun_out/e2e/banded.json 2> gpurun_out/e2e/banded.err
grep "vgk_banded_align" gpurun_out/e2e/banded.err | tail -24
python -c "
import json
d=json.loads(open('gpurun_out/e2e/banded.json').read().strip().splitlines()[-1]); print('banded', d['value'], d['config'].get('end_to_end_from_host_buffers_alignments_per_s'))"
VGAMD_TIMING=1 timeout -s KILL 300 python bench.py --workload gapless --steps 3 --warmup 1 --no-cpu > gpurun_out/e2e/gapless.json 2> gpurun_out/e2e/gapless.err
tail -30 gpurun_out/e2e/gapless.err | cut -c1-200
python -c "
import json
d=json.loads(open('gpurun_out/e2e/gapless.json').read().strip().splitlines()[-1]); print('gapless', d['value'], d['config'].get('end_to_end_from_host_buffers_reads_per_s'))"
