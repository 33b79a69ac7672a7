# WFA kernel on the GPU box under a few settings: average kernel duration per setting ("name|ENV=.. ENV=..")
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  name=${cfg%%|*}; envs=${cfg#*|}; [ "$envs" = "$cfg" ] && envs=""
  env $envs timeout 200 python bench.py --workload wfa --reads ${READS:-500000} --steps 5 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms/launch', round(d['roofline']['avg_launch_ms'],3), 'failed', d.get('problems_failed'))"
done
