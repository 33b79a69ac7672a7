cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_gapless.py tests/test_wfa.py tests/test_tail_forest.py tests/test_giraffe_stage.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest.log
B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --no-cpu --steps 3 --warmup 1"
( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gapless -o s -- $B > $O/stats_gapless.log 2>&1 ) < /dev/null
f=$(find $O/stats_gapless -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gapless" "$f" < /dev/null | cut -c1-120
export VGAMD_CONFIG2_ONE_CONTEXT=1
B="python $GRAFT_REPO_ROOT/bench.py --workload config2 --reads 1000000 --no-cpu --steps 2 --warmup 1"
( cd /tmp && timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -o s -- $B > $O/stats_config2.log 2>&1 ) < /dev/null
f=$(find $O/stats_config2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gapless\|tail\|minimizer_kernel" "$f" < /dev/null | cut -c1-120 | head -8
