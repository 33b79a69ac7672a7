# counter passes over the gapless kernels (one pass per counter set; kernel trace only, as gpurun requires)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-gpmc}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --reads 1000000 --no-cpu --steps 1 --warmup 0"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o g -- $B > $OUT/p$i.log 2>&1 )
done <<SETS
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM
TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU SQ_INSTS
SETS
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "gapless" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k, {c: "%.4g" % (x / n[(k, c)]) for c, x in v.items()})
PY
