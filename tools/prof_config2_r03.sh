# configs[2] as the whole stage at chr22 scale: the bench line, the kernel statistics and the FETCH / WRITE counter passes (run through gpurun;
# everything lands under gpurun_out/r03_config2/, the summaries are copied into profiles/r03/ afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_config2
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 420 python bench.py --workload config2 --steps 2 --warmup 1 > $OUT/bench_config2_10M.json 2> $OUT/bench_config2_10M.err
tail -c 400 $OUT/bench_config2_10M.err
( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload config2 --reads 2000000 --steps 2 --warmup 1 --no-cpu > $OUT/stats.log 2>&1 )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload config2 --reads 1000000 --steps 1 --warmup 0 --no-cpu > $OUT/pmc_$c.log 2>&1 )
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/stats/*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"].split("(")[0][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:10]:
        print(k, "%.4g" % v)
PY
