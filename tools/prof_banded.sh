# Banded-global profiling recipe (run on the GPU box through gpurun; outputs under gpurun_out/).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bprof_stats -o banded -- python bench.py --workload banded --reads 100000 --steps 3 --warmup 1 --no-cpu > $OUT/bprof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/bprof_sq -o banded -- python bench.py --workload banded --reads 100000 --steps 1 --warmup 0 --no-cpu > $OUT/bprof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/bprof_sq2 -o banded -- python bench.py --workload banded --reads 100000 --steps 1 --warmup 0 --no-cpu > $OUT/bprof_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bprof_fetch -o banded -- python bench.py --workload banded --reads 100000 --steps 1 --warmup 0 --no-cpu > $OUT/bprof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bprof_write -o banded -- python bench.py --workload banded --reads 100000 --steps 1 --warmup 0 --no-cpu > $OUT/bprof_write.log 2>&1
find $OUT -name "banded*.csv" | head -40
