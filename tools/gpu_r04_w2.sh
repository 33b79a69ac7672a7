# round 4: gssw_walk_first_kernel over the reads in fill order (the default) against problem order (VGAMD_WALK_PROBLEM_ORDER=1) — kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04w2; mkdir -p $O
for v in fill problem; do
  if [ $v = problem ]; then export VGAMD_WALK_PROBLEM_ORDER=1; fi
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1 > $O/stats_$v.log 2>&1 ) < /dev/null
  tail -1 $O/stats_$v.log | cut -c1-200
done
python3 - <<'PY'
import glob, csv, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04w2'
for v in ('fill', 'problem'):
    for f in glob.glob(O + '/stats_%s/**/*kernel_stats.csv' % v, recursive=True):
        for r in list(csv.DictReader(open(f)))[:5]: print(v, r['Name'][:70], r['Calls'], r['AverageNs'])
PY
