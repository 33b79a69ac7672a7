cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04y; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
export VGAMD_XBAND_ONE_BATCH=1
B="python $GRAFT_REPO_ROOT/bench.py --workload xband --no-cpu --steps 3 --warmup 2"
( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_xband -o s -- $B > $O/stats_xband.log 2>&1 ) < /dev/null
f=$(find $O/stats_xband -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "xdrop_band" "$f" < /dev/null | cut -c1-120
