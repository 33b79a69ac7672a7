# Round 4's measurement pass on the GPU box (every command under `timeout`): kernel statistics and the FETCH_SIZE / WRITE_SIZE counter
# passes (separate runs, kernel trace only) of the workloads whose kernels changed this round — linear (the fill: v_pk_maximum3_f16) and
# configs[2] as one context over one batch (the seeding kernel) — plus gapless / banded / wfa / xband again so that every stored constant
# is of this round's library.  The CSVs land in gpurun_out/r04_pmc/; tools/pmc_constants.py r04 (run in the dev container, where git is)
# turns them into profiles/pmc_constants.json with the commit they were measured at.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_pmc; rm -rf $OUT; mkdir -p $OUT
run() {   # name, workload, reads, extra env
  local B="python $GRAFT_REPO_ROOT/bench.py --workload $2 --reads $3 --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o s -- $B > $OUT/stats_$1.log 2>&1 )
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_$1 -o p -- $B > $OUT/${c}_$1.log 2>&1 )
  done
}
run linear linear 400000
export VGAMD_CONFIG2_ONE_CONTEXT=1
run config2 config2 1000000
unset VGAMD_CONFIG2_ONE_CONTEXT
run gapless gapless 1000000
run banded banded 100000
run wfa wfa 500000
ls $OUT
