# Gapless profiling, round 2 (run on the GPU box through gpurun): kernel stats + HBM traffic counters of the two-pass extension
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-gprof}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --reads 1000000 --no-cpu"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o gapless -- $B --steps 3 --warmup 1 > $OUT/stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o gapless -- $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o gapless -- $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -o gapless -- $B --steps 1 --warmup 0 > $OUT/sq.log 2>&1
python3 - <<PY
import csv, glob, collections
for what in ("fetch", "write", "sq"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % what):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            if "gapless" in k: print(what, k, dict(v))
PY
head -6 $OUT/stats/*kernel_stats.csv | cut -c1-160
