#!/bin/bash
# Round-5 GPU runs, one parameterised script (no more one-off gpu_r04_*.sh files): `gpurun -- bash tools/gpu_r05.sh <stage>`.
# Everything is written under gpurun_out/r05/<stage>/; what is cited goes to profiles/r05/ by hand.
set -u
stage=${1:-tests}
out=gpurun_out/r05/$stage
mkdir -p "$out"
export TMPDIR=/tmp
case "$stage" in
  new_tests)      # the tests added this round, before the whole suite is paid for
    timeout 900 python -m pytest tests/test_gssw_gpu_parity.py tests/test_windows.py tests/test_banded.py -m gpu -x -q \
        -k "speculat or far_pred or variation or geometry" > "$out/pytest_new.log" 2>&1; echo "rc=$?" >> "$out/pytest_new.log"; tail -5 "$out/pytest_new.log" ;;
  tests)          # the whole -m gpu suite + smoke
    timeout 2400 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -5 "$out/pytest_gpu.log"
    timeout 600 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "rc=$?" >> "$out/smoke.log"; tail -2 "$out/smoke.log" ;;
  headline)       # the headline alone, and the speculation's feedback on a stream of reads that all miss (5 % indels)
    timeout 600 python bench.py --no-secondary > "$out/bench_headline.json" 2> "$out/bench_headline.err"; tail -c 600 "$out/bench_headline.json"
    for pol in 0 2 1; do
      VGAMD_SPEC_POLICY=$pol timeout 600 python bench.py --reads 200000 --indel-rate 0.05 --steps 12 --warmup 0 --no-cpu --no-e2e --no-secondary \
          > "$out/bench_indel5_policy$pol.json" 2> "$out/bench_indel5_policy$pol.err"
      python - "$out/bench_indel5_policy$pol.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); o = d["config"]["one_stream"]
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "speculated", o["speculated_steps"], o["speculation"], o["step_ms_fill_plus_tail"])
PY
    done ;;
  config2)        # configs[2] with every tail alignment of a million reads compared
    timeout 900 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --cpu-sample 1000000 > "$out/bench_config2.json" 2> "$out/bench_config2.err"
    python - "$out/bench_config2.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["parity"], d["cpu_baseline"]["value"])
PY
    ;;
  rescue_tests)   # extension windows, the rescue stage on the resident graph, the paired stage — on HIP
    timeout 1200 python -m pytest tests/test_extension_windows.py tests/test_rescue_resident.py tests/test_paired_stage.py -m gpu -x -q > "$out/pytest_rescue.log" 2>&1
    echo "rc=$?" >> "$out/pytest_rescue.log"; tail -5 "$out/pytest_rescue.log" ;;
  paired)         # the configs[3] slice: the rescue half on the resident graph, and round 4's per-graph form beside it; host threads 16 and 2
    for th in 0 2; do
      VGAMD_HOST_THREADS=$th timeout 900 python bench.py --workload paired --steps 3 --warmup 1 --cpu-sample $([ $th = 0 ] && echo 200000 || echo 2000) \
          > "$out/bench_paired_threads$th.json" 2> "$out/bench_paired_threads$th.err"
      python - "$out/bench_paired_threads$th.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), d["parity"] and {k: v for k, v in d["parity"].items() if k != "what"}, {k: round(v, 2) for k, v in d["config"]["stage_ms_per_batch"].items()}, d["config"].get("one_context"), d["roofline_rescue"])
PY
    done
    VGAMD_PAIRED_PER_GRAPH=1 timeout 600 python bench.py --workload paired --steps 2 --warmup 1 --no-cpu > "$out/bench_paired_per_graph.json" 2> "$out/bench_paired_per_graph.err"
    python -c "import json,sys; d=json.loads(open('$out/bench_paired_per_graph.json').read().strip().splitlines()[-1]); print('per-graph form', round(d['value']))" ;;
  pmc)            # kernel statistics + FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel trace only) of the named workloads -> gpurun_out/r05_pmc (tools/pmc_constants.py r05)
    P=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc; mkdir -p $P
    shift
    for w in "$@"; do
      case $w in linear) R=400000;; config2) R=1000000; export VGAMD_CONFIG2_ONE_CONTEXT=1;; gapless) R=1000000;; banded) R=100000;; wfa) R=500000;; paired) R=500000;; longread) R=4000;; xband) R=200000;; *) R=0;; esac
      B="python $GRAFT_REPO_ROOT/bench.py --workload $w --reads $R --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
      ( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_$w -o s -- $B > $P/stats_$w.log 2>&1 )
      for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do      # (PMC_COUNTERS="": the kernel statistics only)
        ( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/${c}_$w -o p -- $B > $P/${c}_$w.log 2>&1 )
      done
      unset VGAMD_CONFIG2_ONE_CONTEXT
    done
    ls $P ;;
  merged_runs)    # the gapless search on the merged-run index against the node-by-node index: the gapless leg, configs[2], and the new gpu tests
    timeout 900 python -m pytest tests/test_gapless.py tests/test_giraffe_stage.py tests/test_tail_forest.py tests/test_minimizer.py tests/test_wfa.py -m gpu -x -q > "$out/pytest_merged.log" 2>&1
    echo "rc=$?" >> "$out/pytest_merged.log"; tail -3 "$out/pytest_merged.log"
    for m in merged plain; do
      if [ $m = plain ]; then export VGAMD_HAPLO_NO_MERGE=1; else unset VGAMD_HAPLO_NO_MERGE; fi
      timeout 600 python bench.py --workload gapless --steps 5 --warmup 2 > "$out/bench_gapless_$m.json" 2> "$out/bench_gapless_$m.err"
      timeout 900 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --cpu-sample 1000000 > "$out/bench_config2_$m.json" 2> "$out/bench_config2_$m.err"
      python - "$out" $m <<'PY'
import json, sys
o, m = sys.argv[1], sys.argv[2]
g = json.loads(open("%s/bench_gapless_%s.json" % (o, m)).read().strip().splitlines()[-1])
c = json.loads(open("%s/bench_config2_%s.json" % (o, m)).read().strip().splitlines()[-1])
print(m, "gapless", round(g["value"]), g["parity"], "launch ms", g["roofline"].get("avg_launch_ms"), "| config2", round(c["value"]), {k: v for k, v in c["parity"].items() if k != "what"},
      "kernel ms/batch", {k: round(v, 2) for k, v in c["config"]["kernel_ms_per_batch"].items()}, "one context", c["config"]["one_context"])
PY
    done
    unset VGAMD_HAPLO_NO_MERGE ;;
  misc)           # the wide route's first number; configs[2] with one copy of the indexes per context (round 4's form) beside the shared copy
    timeout 600 python bench.py --workload wide --steps 3 --warmup 1 > "$out/bench_wide.json" 2> "$out/bench_wide.err"
    python -c "import json; d=json.loads(open('$out/bench_wide.json').read().strip().splitlines()[-1]); print('wide', round(d['value']), d['parity'], d['config']['kernel_ms_per_step'], d['config']['gcups_fill'], d['roofline']['frac'], d['cpu_baseline'] and d['cpu_baseline']['value'])"
    for own in 0 1; do
      if [ $own = 1 ]; then export VGAMD_CONFIG2_OWN_INDEXES=1; else unset VGAMD_CONFIG2_OWN_INDEXES; fi
      timeout 900 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --no-cpu > "$out/bench_config2_own$own.json" 2> "$out/bench_config2_own$own.err"
      python -c "import json; d=json.loads(open('$out/bench_config2_own$own.json').read().strip().splitlines()[-1]); print('config2 own=$own', round(d['value']), d['config']['ms_per_batch'], d['config']['one_context'])"
    done
    unset VGAMD_CONFIG2_OWN_INDEXES ;;
  ab_r04)         # configs[2] on round 4's library (build/variants/libvgamd_r04.so: git worktree of 6434d4a, `make lib`) beside this round's, same bench, same session
    for rep in 1 2; do
      VGAMD_ENGINE_LIB=$GRAFT_REPO_ROOT/build/variants/libvgamd_r04.so VGAMD_CONFIG2_OWN_INDEXES=1 timeout 600 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --no-cpu > "$out/config2_r04lib_$rep.json" 2> "$out/config2_r04lib_$rep.err"
      VGAMD_CONFIG2_OWN_INDEXES=1 timeout 600 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --no-cpu > "$out/config2_r05lib_own_$rep.json" 2> "$out/config2_r05lib_own_$rep.err"
      timeout 600 python bench.py --workload config2 --reads 8000000 --steps 3 --warmup 1 --no-cpu > "$out/config2_r05lib_shared_$rep.json" 2> "$out/config2_r05lib_shared_$rep.err"
    done
    python - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/config2_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f.split("/")[-1], round(d["value"] / 1e6, 1), "M reads/s", round(c["ms_per_batch"], 2), "ms/batch", {k: round(v, 2) for k, v in c["kernel_ms_per_batch"].items()}, "one context", round(c["one_context"]["ms_per_batch"], 2))
    except Exception as e:
        print(f, "failed", e)
PY
    ;;
  banded_blocks)  # bands of more than 512 diagonals as blocks of 8 rows per lane: the banded gpu tests, and the fill's time beside round 4's tall-lane kernels
    timeout 900 python -m pytest tests/test_banded.py -m gpu -x -q > "$out/pytest_banded.log" 2>&1; echo "rc=$?" >> "$out/pytest_banded.log"; tail -3 "$out/pytest_banded.log"
    for lib in vg_amd/libvgamd.so build/variants/libvgamd_r04.so; do
      PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests timeout 600 python - $lib <<'PY' | tee -a "$out/wide_band_fill_ms.txt"
import ctypes, sys, time, numpy as np
import util
from vg_amd import capi
from test_banded import wide_band_problems
problems = wide_band_problems(95, 2000)
bs = capi.BandedSet.from_lists(problems)
eng = capi.Engine(lib=sys.argv[1])
eng.lib.vgk_banded_last.restype = ctypes.c_double; eng.lib.vgk_banded_last.argtypes = [ctypes.c_void_p, ctypes.c_int]
eng.banded_align(bs)
t = time.perf_counter(); res, ops = eng.banded_align(bs); t = time.perf_counter() - t
cells = eng.lib.vgk_banded_last(eng.h, 2); fill = eng.lib.vgk_banded_last(eng.h, 0)
print(sys.argv[1], "2000 wide-band problems: fill %.2f ms, %.0f GCUPS, call %.1f ms, aligned %d" % (fill, cells / (fill * 1e-3) / 1e9 if fill else 0, 1e3 * t, int((res["status"] == 0).sum())))
PY
    done ;;
  longread)       # configs[4] as reads: two batches in flight (two ChainStages) against one lane; 2 host threads as well
    for tag in two_lanes one_lane two_lanes_2threads; do
      case $tag in one_lane) export VGAMD_LONGREAD_ONE_LANE=1;; *) unset VGAMD_LONGREAD_ONE_LANE;; esac
      case $tag in two_lanes_2threads) T=2;; *) T=0;; esac
      if [ $T = 0 ]; then timeout 600 python bench.py --workload longread --steps 3 --warmup 1 > "$out/bench_longread_$tag.json" 2> "$out/bench_longread_$tag.err"
      else timeout 600 taskset -c 0-$((T-1)) python bench.py --workload longread --steps 3 --warmup 1 --no-cpu > "$out/bench_longread_$tag.json" 2> "$out/bench_longread_$tag.err"; fi
      python - "$out/bench_longread_$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[1], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["batches"], c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1), d["parity"] and {k: v for k, v in d["parity"].items() if k != "what"}, "budgets:", round(c["with_point_budgets"]["reads_per_s"]))
PY
    done; unset VGAMD_LONGREAD_ONE_LANE ;;
  longread_sweep) # the long-read stage against one knob at a time: SWEEP="lanes 3 4" | "waves 6 8" (resident wavefronts per CU of a WFA launch) |
                  # "points 65536 262144" (what the WFA kernel's large size stores per link) | "batch 8000 16000" (reads per batch)
    set -- ${SWEEP:-batch 8000 16000}; knob=$1; shift
    for v in "$@"; do
      case $knob in lanes) export VGAMD_LONGREAD_LANES=$v;; waves) export VGAMD_WFA_WAVES_PER_CU=$v;; points) export VGAMD_WFA_LARGE_POINTS=$v;; batch) export VGAMD_LONGREAD_BATCH=$v;; *) echo "unknown knob $knob"; exit 2;; esac
      timeout -s KILL 600 python bench.py --workload longread --steps 3 --warmup 1 ${SWEEP_ARGS:-} > "$out/${knob}_$v.json" 2> "$out/${knob}_$v.err"
      python - "$out/${knob}_$v.json" $knob $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[2], sys.argv[3], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1), c["links"],
      d["parity"] and {k: v for k, v in d["parity"].items() if k != "what"})
PY
    done; unset VGAMD_LONGREAD_LANES VGAMD_WFA_WAVES_PER_CU VGAMD_WFA_LARGE_POINTS VGAMD_LONGREAD_BATCH ;;
  xband_eights)   # the X-drop band's 8-lane class (tails of at most 63 bases eight to a wavefront) against four to a wavefront: gpu tests, the xband leg
    timeout 900 python -m pytest tests/test_xdrop_band.py tests/test_rescue_fixups.py tests/test_golden_gssw_oracle.py -m gpu -x -q > "$out/pytest_xband.log" 2>&1; echo "rc=$?" >> "$out/pytest_xband.log"; tail -3 "$out/pytest_xband.log"
    for m in eights fours; do
      case $m in eights) export VGAMD_XBAND_EIGHTS=1;; *) unset VGAMD_XBAND_EIGHTS;; esac
      for rep in 1 2; do
      timeout 600 python bench.py --workload xband --steps 5 --warmup 2 $([ $m = fours ] && echo --no-cpu) > "$out/bench_xband_${m}_$rep.json" 2> "$out/bench_xband_${m}_$rep.err"
      python - "$out/bench_xband_${m}_$rep.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[1], round(d["value"]), "ms/step", round(d["ms_per_step"], 2), {k: v for k, v in c.items() if "ms" in k or "kernel" in k}, d["roofline"].get("frac"), d.get("parity") and {k: v for k, v in d["parity"].items() if k != "what"})
PY
      done
    done; unset VGAMD_XBAND_EIGHTS ;;
  wfa_runs)       # the WFA wavefront kernel on the merged-run index against the node-by-node walk: gpu tests, the WFA leg, the long-read stage
    timeout 900 python -m pytest tests/test_wfa.py tests/test_longread_stage.py tests/test_chain_alignment.py -m gpu -x -q > "$out/pytest_wfa.log" 2>&1; echo "rc=$?" >> "$out/pytest_wfa.log"; tail -3 "$out/pytest_wfa.log"
    for m in runs nodes; do
      case $m in nodes) export VGAMD_WFA_NO_MERGE=1;; *) unset VGAMD_WFA_NO_MERGE;; esac
      timeout 600 python bench.py --workload wfa --steps 5 --warmup 2 $([ $m = nodes ] && echo --no-cpu) > "$out/bench_wfa_$m.json" 2> "$out/bench_wfa_$m.err"
      python - "$out/bench_wfa_$m.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "ms/step", round(d["ms_per_step"], 2), d.get("parity") and {k: v for k, v in d["parity"].items() if k != "what"}, {k: v for k, v in d["config"].items() if "ms" in k})
PY
      timeout 600 python bench.py --workload longread --steps 3 --warmup 1 $([ $m = nodes ] && echo --no-cpu) > "$out/bench_longread_$m.json" 2> "$out/bench_longread_$m.err"
      python - "$out/bench_longread_$m.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[1], round(d["value"]), "ms/batch", round(c["ms_per_batch"], 1), c["one_lane"], {k: round(v, 1) for k, v in c["stage_ms_per_batch"].items()}, "wfa ms", round(c["wfa_kernel_ms"], 1), d["parity"] and {k: v for k, v in d["parity"].items() if k != "what"}, "budgets:", round(c["with_point_budgets"]["reads_per_s"]))
PY
    done; unset VGAMD_WFA_NO_MERGE ;;
  gapless_pmc)    # where the gapless search's bytes come from: L2 hits / misses, vector-memory and scratch instruction counts of its kernels (counter passes, kernel trace only)
    P=$GRAFT_REPO_ROOT/gpurun_out/r05_pmc; mkdir -p $P
    ( cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -ioE "\b(TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|TCP_TCC_READ_REQ_sum|TCP_TCC_WRITE_REQ_sum|SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR|SQ_INSTS_FLAT|SQ_INSTS_LDS|SQ_INSTS_SALU|SQ_INSTS_VALU|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_INSTS_SMEM|SQ_INST_LEVEL_VMEM|SQ_INSTS_VMEM|SQ_INSTS_SCRATCH[A-Z_]*|SPI_[A-Z_]*SCRATCH[A-Z_]*)\b" | sort -u > $P/counters_available.txt ); cat $P/counters_available.txt | tr '\n' ' '; echo
    B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --reads 1000000 --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
    for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
      tag=$(echo $set | tr ' ' '+')
      ( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/gapless_$tag -o p -- $B > $P/gapless_$tag.log 2>&1 )
      python - $P/gapless_$tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counters in", sys.argv[1]); sys.exit(0)
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]); tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    if "gapless" in k[0]: print(k[0], k[1], "%.4g per dispatch over %d dispatches" % (tot[k] / len(n[k]), len(n[k])))
PY
    done ;;
  default)        # what the driver runs: the headline + every secondary record
    timeout 1700 python bench.py > "$out/bench_default_run.json" 2> "$out/bench_default_run.err"; tail -c 400 "$out/bench_default_run.json" ;;
  *) echo "unknown stage $stage"; exit 2 ;;
esac
