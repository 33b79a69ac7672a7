# Everything profiles/r02 holds, in one go (run on the GPU box through gpurun): bench lines of every workload, kernel stats, counters.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_final}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_1M.json 2> $OUT/bench_1M.err
python bench.py --workload gapless --steps 5 --warmup 2 > $OUT/bench_gapless_1M.json 2>> $OUT/bench.err
python bench.py --workload wfa --steps 5 --warmup 1 > $OUT/bench_wfa_500k.json 2>> $OUT/bench.err
python bench.py --workload banded --steps 5 --warmup 2 > $OUT/bench_banded_100k.json 2>> $OUT/bench.err
python bench.py --workload tails --tails-per-problem-graphs --steps 5 --warmup 2 > $OUT/bench_tails_200k.json 2>> $OUT/bench.err
timeout 300 python bench.py --workload forest --steps 3 --warmup 1 > $OUT/bench_forest_1M.json 2>> $OUT/bench.err
timeout 300 python bench.py --workload giraffe --steps 3 --warmup 1 > $OUT/bench_giraffe_1M.json 2>> $OUT/bench.err
timeout 400 python bench.py --workload longread --steps 3 --warmup 1 > $OUT/bench_longread_4k.json 2>> $OUT/bench.err
timeout 300 python bench.py --workload xband > $OUT/bench_xband_200k.json 2>> $OUT/bench.err
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/forest -o forest -- python $GRAFT_REPO_ROOT/bench.py --workload forest --steps 3 --warmup 1 --no-cpu > $OUT/forest.log 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/giraffe -o giraffe -- python $GRAFT_REPO_ROOT/bench.py --workload giraffe --steps 3 --warmup 1 --no-cpu > $OUT/giraffe.log 2>&1)
timeout 300 bash tools/prof_gapless_r02.sh ${1:-r02_final}/gapless > $OUT/prof_gapless.log 2>&1
bash tools/prof_r02.sh ${1:-r02_final}/headline > $OUT/prof_headline.log 2>&1
cd /tmp
for w in wfa banded; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --no-cpu > $OUT/$w.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${w}_fetch -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-cpu > $OUT/${w}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${w}_write -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 1 --warmup 0 --no-cpu > $OUT/${w}_write.log 2>&1
done
find $OUT -name "*_kernel_stats.csv" | xargs -I{} sh -c 'echo {}; head -4 {} | cut -c1-150'
