cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04x; mkdir -p $O; export TMPDIR=/tmp
for v in default gdouble; do
  unset VGAMD_ENGINE_LIB; [ $v != default ] && export VGAMD_ENGINE_LIB=$GRAFT_REPO_ROOT/build/variants/libvgamd_$v.so
  B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --no-cpu --steps 3 --warmup 1"
  ( cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- $B > $O/stats_$v.log 2>&1 ) < /dev/null
  echo $v; f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gapless_search\|gapless_rules" "$f" < /dev/null | cut -c1-110
done
