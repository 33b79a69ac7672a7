# the whole -m gpu suite and smoke() on the final library (no bench lines)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_final; mkdir -p $O
timeout -s KILL 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
