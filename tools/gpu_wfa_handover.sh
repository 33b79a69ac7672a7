cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfahand
for h in ${HS:-16 32 64 128}; do
  VGAMD_WFA_HAND_OVER_POINTS=$h timeout -s KILL 300 python bench.py --workload wfa --steps 5 --warmup 2 --no-cpu > gpurun_out/wfahand/h$h.json 2> gpurun_out/wfahand/h$h.err
  python -c "
import json
d=json.loads(open('gpurun_out/wfahand/h$h.json').read().strip().splitlines()[-1]); print('hand over at', $h, 'ms', round(d['ms_per_step'],2), 'e2e/s', round(d['config']['end_to_end_from_host_buffers_alignments_per_s']))"
done
