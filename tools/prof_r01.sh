# Round-1 profiling recipe (run on the GPU box through gpurun; outputs under gpurun_out/).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > $OUT/prof_stats.log 2>&1
# hardware counters: separate passes, kernel-trace only (HBM bytes per MI355X_MICROARCH.md §HBM)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o r01 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --reads 400000 > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o r01 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --reads 400000 > $OUT/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_sq -o r01 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --reads 400000 > $OUT/prof_sq.log 2>&1
find $OUT -name "*.csv" | head -40
