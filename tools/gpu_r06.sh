#!/bin/bash
# Round-6 GPU runs, one parameterised script: `gpurun -- bash tools/gpu_r06.sh <stage> [args]`.
# Everything is written under gpurun_out/r06/<stage>/; what is cited goes to profiles/r06/ by hand.
set -u
stage=${1:-tests}
out=gpurun_out/r06/$stage
mkdir -p "$out"
export TMPDIR=/tmp
last_json() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith('{')][-1]); $2" "$1"; }
case "$stage" in
  tests)          # the whole -m gpu suite + smoke
    timeout 2700 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -5 "$out/pytest_gpu.log"
    timeout 600 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "rc=$?" >> "$out/smoke.log"; tail -2 "$out/smoke.log" ;;
  some_tests)     # named test files / -k expression: SOME="tests/test_x.py -k foo"
    timeout 1500 python -m pytest ${SOME:-tests} -m gpu -x -q > "$out/pytest_some.log" 2>&1; echo "rc=$?" >> "$out/pytest_some.log"; tail -8 "$out/pytest_some.log" ;;
  default)        # what the driver runs: the headline line + every secondary record
    t0=$(date +%s); timeout 1700 python bench.py > "$out/bench_default_run.out" 2> "$out/bench_default_run.err"; echo "wall $(( $(date +%s) - t0 )) s"
    cp bench_secondary.json "$out/" 2>/dev/null
    python - "$out/bench_default_run.out" <<'PY'
import json, sys
lines = open(sys.argv[1]).read().strip().splitlines()
print("last line bytes", len(lines[-1]), "json lines", sum(l.startswith("{") for l in lines))
d = json.loads(lines[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["parity"])
for l in lines:
    if l.startswith("[secondary] "):
        r = json.loads(l[12:]); print(r["workload"], r.get("value"), r.get("error"), r.get("parity") and {k: v for k, v in r["parity"].items() if k != "what"}, r["wall_s"])
PY
    ;;
  leg)            # one secondary leg: LEG="longread --steps 3 --warmup 1"
    set -- ${LEG:-longread --steps 3 --warmup 1}; w=$1; shift
    timeout 1200 python bench.py --workload $w "$@" > "$out/bench_$w${TAG:-}.json" 2> "$out/bench_$w${TAG:-}.err"; tail -3 "$out/bench_$w${TAG:-}.err"
    last_json "$out/bench_$w${TAG:-}.json" "print(round(d['value']), d['ms_per_step'], d.get('parity') and {k: v for k, v in d['parity'].items() if k != 'what'}, d['roofline'].get('frac'), d['config'].get('stage_ms_per_batch'), d['config'].get('stitch_device_ms'))" ;;
  ab)             # the headline on another build of the library beside the shipped one, twice each: AB=build/variants/libvgamd_NAME.so [ARGS="--workload linear"]
    for rep in 1 2; do
      for lib in shipped ${AB:?}; do
        if [ $lib = shipped ]; then unset VGAMD_ENGINE_LIB; tag=shipped; else export VGAMD_ENGINE_LIB=$GRAFT_REPO_ROOT/$lib; tag=$(basename $lib .so); fi
        timeout 600 python bench.py ${ARGS:-} --no-secondary --no-cpu --no-e2e > "$out/${tag}_$rep.json" 2> "$out/${tag}_$rep.err"
        last_json "$out/${tag}_$rep.json" "print('$tag', $rep, round(d['value']), round(d['ms_per_step'], 3), d['roofline'].get('avg_launch_ms'), d['roofline'].get('second_fill_ms'), d['roofline'].get('traceback_tail_ms'), d.get('parity'))"
      done
    done; unset VGAMD_ENGINE_LIB ;;
  sweep)          # one leg against one environment knob: SWEEP="VGAMD_LONGREAD_LANES 2 3 4" LEGARGS="longread --steps 3 --warmup 1 --no-cpu"
    set -- ${SWEEP:?}; knob=$1; shift
    for v in "$@"; do
      export $knob=$v
      timeout 900 python bench.py --workload ${LEGARGS:?} > "$out/${knob}_$v.json" 2> "$out/${knob}_$v.err"
      last_json "$out/${knob}_$v.json" "print('$knob', '$v', round(d['value']), round(d['ms_per_step'], 2), d['config'].get('ms_per_batch'), d['config'].get('kernel_ms_per_batch'), d['config'].get('one_lane') or d['config'].get('one_context'))"
    done; unset $knob ;;
  long_seeds)     # 15 kbp reads through the uncapped seeding calls on the chr22-scale graph
    timeout 900 python tools/long_read_seeding.py ${N:-4000} > "$out/long_read_seeding.json" 2> "$out/long_read_seeding.err"; tail -2 "$out/long_read_seeding.err"; cat "$out/long_read_seeding.json" ;;
  two_cpus)       # legs pinned to TWO host CPUs (what a rank of an 8-rank run on a 16-CPU box gets) beside the whole box: TWO="longread banded"
    for w in ${TWO:-longread banded}; do
      for cpus in all 2; do
        if [ $cpus = 2 ]; then pre="taskset -c 0-1"; export VGAMD_HOST_THREADS=2; else pre=""; unset VGAMD_HOST_THREADS; fi
        timeout 900 $pre python bench.py --workload $w --steps 3 --warmup 1 --no-cpu > "$out/bench_${w}_cpus_$cpus.json" 2> "$out/bench_${w}_cpus_$cpus.err"
        last_json "$out/bench_${w}_cpus_$cpus.json" "print('$w', '$cpus', round(d['value']), d['ms_per_step'], d['config'].get('stage_ms_per_batch'), d['config'].get('host_ms'), d['config'].get('one_lane'))"
      done
    done; unset VGAMD_HOST_THREADS ;;
  pmc)            # kernel statistics + FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel trace only) of the named workloads -> gpurun_out/r06_pmc (tools/pmc_constants.py r06)
    P=$GRAFT_REPO_ROOT/gpurun_out/r06_pmc; mkdir -p $P
    shift
    for w in "$@"; do
      case $w in linear) R=400000;; config2) R=1000000; export VGAMD_CONFIG2_ONE_CONTEXT=1;; gapless) R=1000000;; banded) R=100000;; wfa) R=500000;; paired) R=500000;; longread) R=4000; export VGAMD_LONGREAD_BATCH=4000 VGAMD_LONGREAD_ONE_LANE=1;; xband) R=200000;; *) R=0;; esac
      B="python $GRAFT_REPO_ROOT/bench.py --workload $w --reads $R --no-cpu --no-e2e --no-secondary --steps 2 --warmup 1"
      ( cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_$w -o s -- $B > $P/stats_$w.log 2>&1 )
      for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do      # (PMC_COUNTERS="": the kernel statistics only)
        ( cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/${c}_$w -o p -- $B > $P/${c}_$w.log 2>&1 )
      done
      unset VGAMD_CONFIG2_ONE_CONTEXT VGAMD_LONGREAD_BATCH VGAMD_LONGREAD_ONE_LANE
    done
    ls $P ;;
  stats_default)  # rocprofv3 kernel statistics of the headline at the bench's own size (1 M reads per launch, the default steps / warmup; no secondary legs, no CPU leg, one stream)
    P=$GRAFT_REPO_ROOT/gpurun_out/r06_pmc; mkdir -p $P
    ( cd /tmp && timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_default -o s -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu --no-e2e > $P/stats_default.log 2>&1 )
    grep -E "^\{" $P/stats_default.log | tail -1 | cut -c1-300; head -6 $P/stats_default/s_kernel_stats.csv | cut -c1-160 ;;
  registers)      # VGPRs, spills, scratch and LDS of every kernel of the built library
    python tools/kernel_registers.py vg_amd/libvgamd.so > "$out/kernel_registers.txt" 2>&1; head -50 "$out/kernel_registers.txt" ;;
  *) echo "unknown stage $stage"; exit 2 ;;
esac
