# FETCH_SIZE / WRITE_SIZE / duration of the gapless kernels in one go (two counter passes + one stats pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/gfetch; rm -rf $OUT; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --reads 1000000 --no-cpu"
( cd /tmp; timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o g -- $B --steps 3 --warmup 1 > $OUT/stats.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o g -- $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o g -- $B --steps 1 --warmup 0 > $OUT/write.log 2>&1 )
python3 - <<PY
import csv, collections
for what in ("fetch", "write"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open("$OUT/%s/g_counter_collection.csv" % what)):
        k = r["Kernel_Name"].split("(")[0]; agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    print(what, {k.replace("vgk::", ""): "%.0f KiB" % (v / cnt[k]) for k, v in agg.items() if "gapless" in k})
for r in csv.DictReader(open("$OUT/stats/g_kernel_stats.csv")):
    if "gapless" in r["Name"]: print("   %-34s avg %.3f ms" % (r["Name"].split("(")[0][5:], float(r["AverageNs"]) / 1e6))
PY
