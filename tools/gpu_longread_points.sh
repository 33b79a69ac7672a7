#!/bin/bash
# the long-read stage against what the WFA kernel's large size can store per link (links that outgrow it are declined and take the DP route)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05/longread
for p in ${POINTS:-65536 262144}; do for l in ${LANES:-2 4}; do
  VGAMD_WFA_LARGE_POINTS=$p VGAMD_LONGREAD_LANES=$l timeout -s KILL 400 python bench.py --workload longread --steps 3 --warmup 1 > gpurun_out/r05/longread/points${p}_lanes$l.json 2> gpurun_out/r05/longread/points${p}_lanes$l.err
  python - gpurun_out/r05/longread/points${p}_lanes$l.json $p $l <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("points", sys.argv[2], "lanes", sys.argv[3], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1), c["links"], {k: v for k, v in d["parity"].items() if k != "what"})
PY
done; done
