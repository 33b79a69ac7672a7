cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04r; mkdir -p $O; export TMPDIR=/tmp
for v in default walknt; do
  unset VGAMD_ENGINE_LIB; [ $v != default ] && export VGAMD_ENGINE_LIB=$GRAFT_REPO_ROOT/build/variants/libvgamd_$v.so
  B="python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1"
  ( cd /tmp && timeout -s KILL 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- $B > $O/stats_$v.log 2>&1 ) < /dev/null
  ( cd /tmp && timeout -s KILL 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o p -- $B > $O/fetch_$v.log 2>&1 ) < /dev/null
  echo $v
  f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gssw_walk_kernel\|gssw_fill_kernel" "$f" < /dev/null | cut -c1-110
  timeout 60 python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/fetch_$v/**/*counter_collection.csv", recursive=True)
if fs:
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]=="FETCH_SIZE": k=r["Kernel_Name"].split("(")[0]; tot[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in tot:
        if "gssw" in k: print(k, "FETCH KiB per dispatch", tot[k]/len(n[k]))
PY
done
