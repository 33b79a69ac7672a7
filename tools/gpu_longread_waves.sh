#!/bin/bash
# the long-read stage with two batches in flight against the resident wavefronts per CU each lane's WFA launch takes (12 = a launch fills the CUs' LDS
# and the other lane's launch waits for it; fewer = the two launches run side by side)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05/longread
for l in ${LANES:-}; do
  VGAMD_LONGREAD_LANES=$l timeout -s KILL 300 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/r05/longread/lanes$l.json 2> gpurun_out/r05/longread/lanes$l.err
  python - gpurun_out/r05/longread/lanes$l.json $l <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("lanes", sys.argv[2], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1))
PY
done
for w in ${WAVES:-6 4 8}; do
  VGAMD_WFA_WAVES_PER_CU=$w timeout -s KILL 300 python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > gpurun_out/r05/longread/w$w.json 2> gpurun_out/r05/longread/w$w.err
  python - gpurun_out/r05/longread/w$w.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("waves/CU", sys.argv[2], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1))
PY
done
