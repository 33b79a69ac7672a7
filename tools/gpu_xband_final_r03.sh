# round 3, X-drop band: SQ counters, kernel statistics and the FETCH_SIZE / WRITE_SIZE passes of the final kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/xband
bash tools/gpu_xband_counters2.sh > gpurun_out/xband/xband_counters.txt 2>&1
cat gpurun_out/xband/xband_counters.txt | tail -8
bash tools/collect_xband_r03.sh
