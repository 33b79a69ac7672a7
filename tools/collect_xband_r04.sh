# kernel statistics and the FETCH_SIZE / WRITE_SIZE passes of the X-drop band kernels of round 4 (the packed fill), one dispatch per call
# (VGAMD_XBAND_ONE_BATCH: the call's sub-batches would each be a dispatch); tools/pmc_constants.py r04 xband reads the CSVs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp VGAMD_XBAND_ONE_BATCH=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; mkdir -p $OUT; rm -rf $OUT/*xband*
B="python $GRAFT_REPO_ROOT/bench.py --workload xband --reads 200000 --no-cpu --steps 2 --warmup 2"
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_xband -o s -- $B > $OUT/stats_xband.log 2>&1 )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_xband -o p -- $B > $OUT/${c}_xband.log 2>&1 )
done
ls $OUT | grep xband
