#!/bin/bash
# the long-read stage against its batch size: a launch of the WFA kernel is as long as its heaviest link's dependent chain, whatever else it holds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05/longread
for b in ${BATCHES:-8000 16000}; do
  VGAMD_LONGREAD_BATCH=$b timeout -s KILL 600 python bench.py --workload longread --steps 3 --warmup 1 $([ $b -gt 8000 ] && echo --no-cpu) > gpurun_out/r05/longread/batch$b.json 2> gpurun_out/r05/longread/batch$b.err
  python - gpurun_out/r05/longread/batch$b.json $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("batch", sys.argv[2], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1), c["links"], d["parity"] and {k: v for k, v in d["parity"].items() if k != "what"}, "gen s", round(c["generation_seconds"], 1))
PY
done
