# WFA profiling recipe (run on the GPU box through gpurun; outputs under gpurun_out/).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wprof_stats -o wfa -- python bench.py --workload wfa --reads 500000 --steps 3 --warmup 1 --no-cpu > $OUT/wprof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/wprof_sq -o wfa -- python bench.py --workload wfa --reads 500000 --steps 1 --warmup 0 --no-cpu > $OUT/wprof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/wprof_sq2 -o wfa -- python bench.py --workload wfa --reads 500000 --steps 1 --warmup 0 --no-cpu > $OUT/wprof_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/wprof_fetch -o wfa -- python bench.py --workload wfa --reads 500000 --steps 1 --warmup 0 --no-cpu > $OUT/wprof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wprof_write -o wfa -- python bench.py --workload wfa --reads 500000 --steps 1 --warmup 0 --no-cpu > $OUT/wprof_write.log 2>&1
