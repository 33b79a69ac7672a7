cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wfaphases
VGAMD_WFA_STATS=1 timeout -s KILL 300 python bench.py --workload longread --steps 1 --warmup 1 --no-cpu > gpurun_out/wfaphases/longread.json 2> gpurun_out/wfaphases/longread.err
grep "wfa wave" gpurun_out/wfaphases/longread.err | head -1 | cut -c1-900
