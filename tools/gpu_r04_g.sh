# round 4, GPU call G: where the band kernel's time goes (VGAMD_TB_DBG: 1 = one column only, 2 = no stores)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04g; mkdir -p $O
export VGAMD_TB_REWALK=1
for dbg in 0 1 2; do
  export VGAMD_TB_DBG=$dbg
  cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$dbg -o p -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof$dbg.log 2>&1
  cd $GRAFT_REPO_ROOT; f=$(find $O/prof$dbg -name '*kernel_stats.csv' | head -1); echo "dbg $dbg"; [ -n "$f" ] && grep -E "band_kernel|bandwalk|fill_kernel" "$f" | cut -d, -f1,4 | cut -c1-120
done
