# round 4, third measurement pass: linear again — the fill is two launches now (every read without traceback codes, the missed reads with them:
# DESIGN.md §27.12); the constant sums both per batch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; mkdir -p $OUT; rm -rf $OUT/*_linear*
B="python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_linear -o s -- $B > $OUT/stats_linear.log 2>&1 ) < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_linear -o p -- $B > $OUT/${c}_linear.log 2>&1 ) < /dev/null
done
ls $OUT | grep linear
