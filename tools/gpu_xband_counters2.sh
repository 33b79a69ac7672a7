# SQ counters and HBM bytes of the X-drop band kernels after the round-3 rework (one pass per counter set, kernel trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
WORKLOAD=xband READS=200000 KERNELS=xdrop_band bash tools/pmc_gapless.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" | cut -c1-400 | tee gpurun_out/xband/counters_r03b.txt
