cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; mkdir -p $O
VGAMD_TIMING=1 timeout -s KILL 400 python bench.py --workload paired --steps 2 --warmup 1 --no-cpu > $O/bench_paired.json 2> $O/bench_paired.err; echo "rc=$?"
grep -E "rescue_stage|align_xdrop_many" $O/bench_paired.err | tail -14
