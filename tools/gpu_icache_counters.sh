cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/icache
SETS=("SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES")
for wl in linear:gssw_walk:1000000 wfa:wfa:500000 gapless:gapless:1000000; do
  IFS=: read w k n <<< "$wl"
  WORKLOAD=$w READS=$n KERNELS=$k bash tools/pmc_gapless.sh "${SETS[@]}" > gpurun_out/icache/$w.txt 2>&1
  cat gpurun_out/icache/$w.txt | cut -c1-400
done
