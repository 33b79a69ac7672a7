cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04z2; mkdir -p $O; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1"
( cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1 ) < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gssw_walk\|gssw_fill" "$f" < /dev/null | cut -c1-120
( cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1 ) < /dev/null
timeout 60 python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/fetch/**/*counter_collection.csv", recursive=True)
if fs:
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]=="FETCH_SIZE": k=r["Kernel_Name"].split("(")[0]; tot[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot:
        if "gssw" in k: print(k, "FETCH KiB per dispatch", tot[k]/len(n[k]))
PY
