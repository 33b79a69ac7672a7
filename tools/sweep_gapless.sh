# gapless kernel experiments on the GPU box: per-kernel average durations for a few settings (env passed through "name|ENV=.. ENV=..")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/gsweep
i=0
for cfg in "$@"; do
  i=$((i+1)); name=${cfg%%|*}; envs=${cfg#*|}; [ "$envs" = "$cfg" ] && envs=""
  ( cd /tmp && env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gsweep/$i -o g -- python $GRAFT_REPO_ROOT/bench.py --workload gapless --steps 4 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/gsweep/$i.log 2>&1 )
  echo "== $name"; python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/gsweep/$i/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "gapless" in r["Name"]: print("   %-40s calls %s avg %.3f ms" % (r["Name"].split("(")[0][5:], r["Calls"], float(r["AverageNs"])/1e6))
PY
done
