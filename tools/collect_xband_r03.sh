# kernel statistics and the FETCH_SIZE / WRITE_SIZE passes of the X-drop band kernels (as tools/collect_r03.sh does for the other workloads)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_pmc; mkdir -p $OUT; rm -rf $OUT/*xband*
B="python $GRAFT_REPO_ROOT/bench.py --workload xband --reads 200000 --no-cpu"
( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_xband -o s -- $B > $OUT/stats_xband.log 2>&1 )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_xband -o p -- $B > $OUT/${c}_xband.log 2>&1 )
done
ls $OUT | grep xband
