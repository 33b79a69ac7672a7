# Round-2 profiling recipe for the banded kernels (run on the GPU box through gpurun; outputs under gpurun_out/$1).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_banded_r02}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload banded --no-cpu"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r02 -- $B --steps 3 --warmup 1 > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r02 -- $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r02 -- $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o r02 -- $B --steps 1 --warmup 0 > $OUT/sq.log 2>&1
python3 - <<PY
import csv, glob, collections
for what in ("fetch", "write", "sq"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % what):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "banded" in k: print(what, k, {c: (x, n[(k, c)]) for c, x in v.items()})
PY
grep -E "banded" $OUT/stats/*kernel_stats.csv | cut -c1-160
