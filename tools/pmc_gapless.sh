# counter passes over the gapless kernels (one pass per counter set given as arguments; kernel trace only, as gpurun requires;
# every pass under its own timeout: a TA/TCP counter set hung a box for the full gpurun limit once)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gpmc
rm -rf $OUT; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload ${WORKLOAD:-gapless} --reads ${READS:-1000000} --no-cpu --steps 1 --warmup 0"
i=0
for set in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o g -- $B > $OUT/p$i.log 2>&1 )
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "${KERNELS:-gapless}" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k, {c: "%.4g" % (x / n[(k, c)]) for c, x in v.items()})
PY
