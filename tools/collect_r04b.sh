# round 4, second measurement pass: the workloads whose kernels changed after tools/collect_r04.sh ran — gapless and configs[2] (the set rules read
# the per-node table), linear (the tracebacks as two kernels; the constant is the fill's) — kernel statistics and FETCH_SIZE / WRITE_SIZE passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; mkdir -p $OUT; rm -rf $OUT/*_gapless* $OUT/*_config2* $OUT/*_linear*
run() {   # name, workload, reads
  local B="python $GRAFT_REPO_ROOT/bench.py --workload $2 --reads $3 --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o s -- $B > $OUT/stats_$1.log 2>&1 ) < /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_$1 -o p -- $B > $OUT/${c}_$1.log 2>&1 ) < /dev/null
  done
}
run linear linear 400000
export VGAMD_CONFIG2_ONE_CONTEXT=1
run config2 config2 1000000
unset VGAMD_CONFIG2_ONE_CONTEXT
run gapless gapless 1000000
ls $OUT | head -40
