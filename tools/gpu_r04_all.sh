#!/bin/bash
# Round 4's one-off GPU scripts folded into one (round-5 advisor item): `gpurun -- bash tools/gpu_r04_all.sh <letter>` runs what tools/gpu_r04_<letter>.sh ran;
# profiles/r04/README.md names the stages.  Provenance for profiles/r04 — not part of the product.  (Bodies unindented: they hold here-documents.)
case "${1:-}" in
a)
# round 4, GPU call A: the wide route's parity on the MI355X, v_pk_maximum3_f16 as an unsigned max3, the fill's build switches on the
# headline batch (fill / walk ms per 1 M reads), and how long each secondary bench leg takes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; mkdir -p $O
timeout -s KILL 60 tools/pkmax3_check > $O/pkmax3_check.txt 2>&1; cat $O/pkmax3_check.txt
timeout -s KILL 600 python -m pytest tests/test_gssw_wide.py tests/test_chain_alignment.py tests/test_gbwt_file.py tests/test_rescue_fixups.py -m gpu -q > $O/pytest_wide.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wide.log; tail -4 $O/pytest_wide.log
for v in default shlor max3 notb notbmax3; do
  lib=build/variants/libvgamd_$v.so; [ $v = default ] && lib=vg_amd/libvgamd.so
  VGAMD_ENGINE_LIB=$PWD/$lib timeout -s KILL 200 python bench.py --no-cpu --no-e2e --no-secondary --steps 5 --warmup 2 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_$v.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('$v', 'fill %.2f walk %.2f step %.2f ms' % (o['fill_ms'], o['traceback_ms'], o['ms_per_step']))" 2>&1 | tail -1
done
for w in "config2 --reads 1000000 --steps 3 --warmup 1 --cpu-sample 50000" "gapless --steps 5 --warmup 2" "xband --steps 3 --warmup 1" "banded --reads 100000 --steps 5 --warmup 2" "wfa --reads 500000 --steps 5 --warmup 2" "longread --steps 3 --warmup 1"; do
  set -- $w; n=$1; t0=$(date +%s.%N)
  timeout -s KILL 300 python bench.py --workload $w > $O/leg_$n.json 2> $O/leg_$n.err; rc=$?
  t1=$(date +%s.%N); echo "leg $n rc=$rc wall $(echo "$t1 - $t0" | bc) s"
done
;;
b)
# round 4, GPU call B: parity of what changed (wide route, minimizer lookups, gapless prefetch, gbwt checks), then the default bench run
# with its secondary records (the driver's command), timed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gssw_wide.py tests/test_chain_alignment.py tests/test_gbwt_file.py tests/test_minimizer.py tests/test_gapless.py tests/test_giraffe_stage.py tests/test_gssw_gpu_parity.py -m gpu -q -x > $O/pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/pytest_b.log; tail -5 $O/pytest_b.log
t0=$(date +%s)
timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b/bench_default.json').read().strip().split('\n')[-1])
o=d['config']['one_stream']; print('headline %.2f M reads/s fill %.2f walk %.2f step %.2f ms parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity']))
for r in d.get('secondary', []):
    print(r['workload'], r.get('error') or ('%.3g %s, %.2f ms/step, frac %s, parity %s, wall %s s' % (r['value'], r['unit'], r['ms_per_step'], (r.get('roofline') or {}).get('frac'), {k: v for k, v in (r.get('parity') or {}).items() if k in ('checked', 'identical')}, r['wall_s'])))
    if r['workload'] == 'config2': print('   ', json.dumps(r['config'].get('kernel_ms_per_batch')), r['config'].get('ms_per_batch'))
PY
;;
c)
# round 4, GPU call C: configs[2] — one context vs two contexts in flight, by batch count and warm-up
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04c; mkdir -p $O
for cfg in "2000000 6 3" "4000000 4 2"; do
  set -- $cfg
  for one in 1 0; do
    tag=r$1_one$one
    if [ $one = 1 ]; then export VGAMD_CONFIG2_ONE_CONTEXT=1; else unset VGAMD_CONFIG2_ONE_CONTEXT; fi
    timeout -s KILL 400 python bench.py --workload config2 --reads $1 --steps $2 --warmup $3 --no-cpu > $O/c2_$tag.json 2> $O/c2_$tag.err
    python3 -c "
import json
d=json.loads(open('$O/c2_$tag.json').read().strip().split('\n')[-1]); c=d['config']
print('$tag', '%.1f M reads/s' % (d['value']/1e6), 'ms/batch %.1f' % c['ms_per_batch'], 'kernels', {k: round(v,2) for k,v in c['kernel_ms_per_batch'].items()}, 'stage', {k: round(v,2) for k,v in c['stage_ms_per_batch'].items()}, 'one_context', c.get('one_context'))"
  done
done
;;
d)
# round 4, GPU call D: the recomputing traceback (TB_REWALK) — parity, then the headline batch in both modes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_gssw_gpu_parity.py tests/test_windows.py tests/test_reference_tap.py tests/test_tail_forest.py tests/test_giraffe_stage.py tests/test_alignment_batch.py -m gpu -q -x > $O/pytest_d.log 2>&1; echo "pytest rc=$?" >> $O/pytest_d.log; tail -5 $O/pytest_d.log
for mode in rewalk codes; do
  if [ $mode = codes ]; then export VGAMD_TB_CODES=1; else unset VGAMD_TB_CODES; fi
  timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 --cpu-sample 200000 > $O/bench_$mode.json 2> $O/bench_$mode.err
  python3 -c "
import json
d=json.loads(open('$O/bench_$mode.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('$mode', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
done
;;
e)
# round 4, GPU call E: TB_REWALK with the band — parity on the MI355X, then the headline batch in both modes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_tb_rewalk.py tests/test_gssw_gpu_parity.py tests/test_windows.py tests/test_reference_tap.py tests/test_giraffe_stage.py -m gpu -q -x > $O/pytest_e.log 2>&1; echo "pytest rc=$?" >> $O/pytest_e.log; tail -5 $O/pytest_e.log
for mode in rewalk codes; do
  if [ $mode = codes ]; then export VGAMD_TB_CODES=1; else unset VGAMD_TB_CODES; fi
  timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 > $O/bench_$mode.json 2> $O/bench_$mode.err
  python3 -c "
import json
d=json.loads(open('$O/bench_$mode.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('$mode', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
done
unset VGAMD_TB_CODES
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
;;
f)
# round 4, GPU call F: TB_REWALK after the boundary rows went lane-major (LDS-staged in the fill) — parity + per-kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04f; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_tb_rewalk.py tests/test_windows.py -m gpu -q -x > $O/pytest_f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f.log; tail -3 $O/pytest_f.log
export VGAMD_TB_REWALK=1
timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 --cpu-sample 100000 > $O/bench_rewalk.json 2> $O/bench_rewalk.err
python3 -c "
import json
d=json.loads(open('$O/bench_rewalk.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('rewalk', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-150
;;
g)
# round 4: the banded call with its geometry on the device — parity tests, the bench line, the call's laps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $O
timeout -s KILL 400 python -m pytest tests/test_banded.py -m gpu -q -x > $O/pytest_banded.log 2>&1 < /dev/null; tail -3 $O/pytest_banded.log
timeout -s KILL 200 python3 bench.py --workload banded --no-cpu --no-secondary --steps 5 --warmup 2 > $O/bench_banded.json 2> $O/bench_banded.err < /dev/null; echo "bench rc=$?"; tail -3 $O/bench_banded.err
VGAMD_BANDED_TIMING=1 timeout -s KILL 200 python3 bench.py --workload banded --no-cpu --no-secondary --steps 1 --warmup 0 > /dev/null 2> $O/bench_banded_laps.err < /dev/null
grep "device geometry" $O/bench_banded_laps.err | tail -22
timeout -s KILL 120 python3 tools/banded_subs.py 2> /dev/null < /dev/null | tee $O/banded_subs.txt
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04g'
d = json.loads(open(O + '/bench_banded.json').read().strip().split('\n')[-1]); c = d['config']
print('banded resident %.2f M/s; from host buffers: device geometry %.2f M/s, host geometry %.2f M/s, one batch %.2f M/s' % (d['value']/1e6, c['end_to_end_from_host_buffers_alignments_per_s']/1e6, c['end_to_end_host_geometry_alignments_per_s']/1e6, c['end_to_end_one_batch_alignments_per_s']/1e6))
PY
;;
h)
# round 4, GPU call H: TB_REWALK with checkpoints every 16 columns and the miss list — parity, kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_tb_rewalk.py tests/test_windows.py tests/test_giraffe_stage.py -m gpu -q -x > $O/pytest_h.log 2>&1; echo "pytest rc=$?" >> $O/pytest_h.log; tail -3 $O/pytest_h.log
export VGAMD_TB_REWALK=1
timeout -s KILL 300 python bench.py --no-e2e --no-secondary --steps 10 --warmup 3 --cpu-sample 100000 > $O/bench_rewalk.json 2> $O/bench_rewalk.err
python3 -c "
import json
d=json.loads(open('$O/bench_rewalk.json').read().strip().split('\n')[-1]); o=d['config']['one_stream']
print('rewalk', '%.2f M reads/s  fill %.2f walk %.2f step %.2f ms  parity %s failed %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity'], d['problems_failed']))"
cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rewalk -- python $GRAFT_REPO_ROOT/bench.py --no-e2e --no-secondary --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<'PY'
import csv,glob
for r in csv.DictReader(open(glob.glob('gpurun_out/r04h/prof/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('band','fill_kernel','rewalk')): print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3),'ms')
PY
;;
i)
# round 4, GPU call I: the WFA hybrid with both kernels at once — parity, then the wfa and longread bench legs against the sequential form
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04i; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_wfa.py tests/test_longread_stage.py tests/test_chain_alignment.py -m gpu -q -x > $O/pytest_i.log 2>&1; echo "pytest rc=$?" >> $O/pytest_i.log; tail -3 $O/pytest_i.log
for mode in at_once sequential; do
  if [ $mode = at_once ]; then export VGAMD_WFA_AT_ONCE=1; else unset VGAMD_WFA_AT_ONCE; fi
  for w in "wfa --reads 500000 --steps 5 --warmup 2" "longread --steps 3 --warmup 1"; do
    set -- $w
    timeout -s KILL 300 python bench.py --workload $w > $O/$1_$mode.json 2> $O/$1_$mode.err
    python3 -c "
import json
d=json.loads(open('$O/$1_$mode.json').read().strip().split('\n')[-1])
print('$1 $mode', '%.4g %s  %.2f ms/step  parity %s' % (d['value'], d['unit'], d['ms_per_step'], {k: v for k, v in (d.get('parity') or {}).items() if k in ('checked','identical')}))"
  done
done
;;
j)
# round 4, GPU call J: the paired-end slice — parity test and its bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_paired_stage.py tests/test_aligner_client.py tests/test_rescue_fixups.py -m gpu -q -x > $O/pytest_j.log 2>&1; echo "pytest rc=$?" >> $O/pytest_j.log; tail -3 $O/pytest_j.log
timeout -s KILL 400 python bench.py --workload paired --steps 3 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err; echo "rc=$?"; tail -c 300 $O/bench_paired.err
python3 -c "
import json
d=json.loads(open('$O/bench_paired.json').read().strip().split('\n')[-1]); c=d['config']
print('%.3g %s  %.1f ms/step  rescued %d (positive %d)  parity %s  cpu %s' % (d['value'], d['unit'], d['ms_per_step'], c['pairs_rescued'], c['rescued_with_positive_score'], d['parity'], d['cpu_baseline']['value']))
print(c['stage_ms'])"
;;
k)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04k; mkdir -p $O
VGAMD_TIMING=1 timeout -s KILL 400 python bench.py --workload paired --steps 2 --warmup 1 --no-cpu > $O/bench_paired.json 2> $O/bench_paired.err; echo "rc=$?"
grep -E "rescue_stage|align_xdrop_many" $O/bench_paired.err | tail -14
;;
l)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xband.log
for mode in pk 16 32; do
  unset VGAMD_XBAND_CELLS32 VGAMD_XBAND_ARITH32; [ $mode = 32 ] && export VGAMD_XBAND_CELLS32=1; [ $mode = 16 ] && export VGAMD_XBAND_ARITH32=1
  VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband_$mode.json 2> $O/bench_xband_$mode.err; echo "bench$mode rc=$?"
  python - <<PY
import json
r=json.loads(open("$O/bench_xband_$mode.json").read().strip().splitlines()[-1])
print("$mode", r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r.get("parity"))
PY
  tail -6 $O/bench_xband_$mode.err
done
;;
last)
# round 4, last GPU call: the first-pass traceback across nodes with several predecessors — gssw and window parity on the MI355X, the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04last; mkdir -p $O
timeout -s KILL 55 python -m pytest tests/test_gssw_gpu_parity.py tests/test_windows.py -m gpu -q -x > $O/pytest.log 2>&1 < /dev/null; tail -2 $O/pytest.log
timeout -s KILL 40 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-secondary --no-cpu > $O/bench_headline.json 2> $O/bench_headline.err < /dev/null; echo "bench rc=$?"
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04last'
d = json.loads(open(O + '/bench_headline.json').read().strip().split('\n')[-1]); o = d['config']['one_stream']
print('headline %.2f M reads/s fill %.2f tail %.2f step %.2f ms parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o['ms_per_step'], d['parity']))
PY
;;
m)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"))
PY
tail -6 $O/bench_xband.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_xband.csv; head -8 $O/kernel_stats_xband.csv | cut -c1-160
rm -rf $O/prof
;;
n)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
for v in default; do
  unset VGAMD_ENGINE_LIB; [ $v != default ] && export VGAMD_ENGINE_LIB=$PWD/build/variants/libvgamd_$v.so
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/prof_$v.log 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_xband_$v.csv; echo $v; grep xdrop_band $O/kernel_stats_xband_$v.csv | cut -c1-120
  rm -rf $O/prof_$v
done
unset VGAMD_ENGINE_LIB
VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 3 --warmup 1 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r["band"])
PY
tail -6 $O/bench_xband.err
;;
o)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; mkdir -p $O
for mode in pipe one; do
  unset VGAMD_XBAND_ONE_BATCH; [ $mode = one ] && export VGAMD_XBAND_ONE_BATCH=1
  VGAMD_XBAND_TIMING=1 timeout -s KILL 300 python bench.py --workload xband --steps 8 --warmup 2 --no-cpu > $O/bench_xband_$mode.json 2> $O/bench_xband_$mode.err; echo "bench rc=$?"
  python - <<PY
import json
r=json.loads(open("$O/bench_xband_$mode.json").read().strip().splitlines()[-1])
print("$mode", r["value"], r["ms_per_step"], r["roofline"].get("avg_launch_ms"), r["roofline"]["frac"])
PY
  grep check $O/bench_xband_$mode.err | tail -1
done
unset VGAMD_XBAND_ONE_BATCH
timeout -s KILL 300 python bench.py --workload xband --steps 8 --warmup 2 > $O/bench_xband_cpu.json 2> /dev/null
python - <<PY
import json
r=json.loads(open("$O/bench_xband_cpu.json").read().strip().splitlines()[-1])
print(r["value"], r["parity"], r["cpu_baseline"])
PY
;;
p)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
timeout -s KILL 1500 python tools/band_vs_exact_config2.py --batches 6 --reads 1000000 > $O/band_vs_exact_config2.json 2> $O/band_vs_exact_config2.err; echo "rc=$?"
tail -8 $O/band_vs_exact_config2.err; cat $O/band_vs_exact_config2.json
;;
q)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
timeout -s KILL 900 python -m pytest tests/test_banded.py tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_banded.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest_banded.log
VGAMD_BANDED_TIMING=1 timeout -s KILL 300 python bench.py --workload banded --steps 5 --warmup 2 --no-cpu > $O/bench_banded.json 2> $O/bench_banded.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads(open("$O/bench_banded.json").read().strip().splitlines()[-1])
print(r["value"], r["config"]["end_to_end_from_host_buffers_alignments_per_s"], r["config"]["end_to_end_one_batch_alignments_per_s"], r["roofline"]["frac"])
PY
grep -n "prepare" $O/bench_banded.err | tail -8 | head -3; tail -42 $O/bench_banded.err | head -30
VGAMD_XBAND_TIMING=1 timeout -s KILL 200 python bench.py --workload xband --steps 8 --warmup 2 --no-cpu > $O/bench_xband.json 2> $O/bench_xband.err < /dev/null
timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_xband.json").read().strip().splitlines()[-1])
print("xband", r["value"], r["ms_per_step"])
PY
grep -E "check|pack" $O/bench_xband.err | tail -5
;;
r)
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
;;
s)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $O
timeout -s KILL 500 python bench.py --workload paired --steps 3 --warmup 1 --cpu-sample 200000 > $O/bench_paired_100k_pairs.json 2> $O/bench_paired.err < /dev/null; echo "rc=$?"
timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_paired_100k_pairs.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["parity"], r["cpu_baseline"])
PY
;;
t)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_gapless.py tests/test_wfa.py tests/test_tail_forest.py tests/test_giraffe_stage.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest.log
B="python $GRAFT_REPO_ROOT/bench.py --workload gapless --no-cpu --steps 3 --warmup 1"
( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gapless -o s -- $B > $O/stats_gapless.log 2>&1 ) < /dev/null
f=$(find $O/stats_gapless -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gapless" "$f" < /dev/null | cut -c1-120
export VGAMD_CONFIG2_ONE_CONTEXT=1
B="python $GRAFT_REPO_ROOT/bench.py --workload config2 --reads 1000000 --no-cpu --steps 2 --warmup 1"
( cd /tmp && timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -o s -- $B > $O/stats_config2.log 2>&1 ) < /dev/null
f=$(find $O/stats_config2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gapless\|tail\|minimizer_kernel" "$f" < /dev/null | cut -c1-120 | head -8
;;
u)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_seed_policy.py tests/test_minimizer.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
export VGAMD_CONFIG2_ONE_CONTEXT=1
for pol in 0 1; do
  unset VGAMD_CONFIG2_POLICY; [ $pol = 1 ] && export VGAMD_CONFIG2_POLICY=1
  timeout -s KILL 300 python bench.py --workload config2 --reads 2000000 --steps 2 --warmup 1 --cpu-sample 20000 > $O/bench_config2_policy$pol.json 2> $O/bench_config2_policy$pol.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_config2_policy$pol.json").read().strip().splitlines()[-1])
c=r["config"]; print("policy $pol", r["value"], r["ms_per_step"], c.get("kernel_ms_per_batch"), r["parity"], {k: c["totals"][k] for k in ("seeds","ext","tails")} if "totals" in c else "")
PY
done
;;
v)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04v; mkdir -p $O
for c in 2 3 4; do
  export VGAMD_CONFIG2_CONTEXTS=$c
  timeout -s KILL 300 python bench.py --workload config2 --reads 12000000 --steps 3 --warmup 1 --no-cpu > $O/bench_config2_ctx$c.json 2> $O/bench_config2_ctx$c.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
r=json.loads(open("$O/bench_config2_ctx$c.json").read().strip().splitlines()[-1])
print("contexts $c", r["value"], r["ms_per_step"], r["config"].get("ms_per_batch"), (r["config"].get("one_context") or {}).get("ms_per_batch"))
PY
done
;;
w)
# round 4: the headline after the walks' profile words moved into registers and the second fill's column cut — gssw parity tests, bench, kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gssw_gpu_parity.py -m gpu -q -x > $O/pytest_gssw.log 2>&1 < /dev/null; tail -2 $O/pytest_gssw.log
timeout -s KILL 200 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary > $O/bench_headline.json 2> $O/bench_headline.err < /dev/null; echo "bench rc=$?"
( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1 > $O/stats.log 2>&1 ) < /dev/null
python3 - <<'PY'
import json, glob, csv, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04w'
d = json.loads(open(O + '/bench_headline.json').read().strip().split('\n')[-1])
o = d['config']['one_stream']; print('headline %.2f M reads/s fill %.2f tail %.2f second fill %.2f step %.2f ms frac %.3f parity %s' % (d['value']/1e6, o['fill_ms'], o['traceback_ms'], o.get('second_fill_ms', 0), o['ms_per_step'], d['roofline']['frac'], d['parity']))
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
;;
w2)
# round 4: gssw_walk_first_kernel over the reads in fill order (the default) against problem order (VGAMD_WALK_PROBLEM_ORDER=1) — kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04w2; mkdir -p $O
for v in fill problem; do
  if [ $v = problem ]; then export VGAMD_WALK_PROBLEM_ORDER=1; fi
  ( cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -o s -- python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1 > $O/stats_$v.log 2>&1 ) < /dev/null
  tail -1 $O/stats_$v.log | cut -c1-200
done
python3 - <<'PY'
import glob, csv, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04w2'
for v in ('fill', 'problem'):
    for f in glob.glob(O + '/stats_%s/**/*kernel_stats.csv' % v, recursive=True):
        for r in list(csv.DictReader(open(f)))[:5]: print(v, r['Name'][:70], r['Calls'], r['AverageNs'])
PY
;;
x)
# round 4: the X-drop band call after its host-side passes were chunked — parity tests, the bench line, the call's laps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r04x; mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_xdrop_band.py -m gpu -q -x > $O/pytest_xband.log 2>&1 < /dev/null; tail -2 $O/pytest_xband.log
VGAMD_XBAND_TIMING=1 timeout -s KILL 200 python3 bench.py --workload xband --no-cpu --no-secondary --steps 5 --warmup 2 > $O/bench_xband.json 2> $O/bench_xband.err < /dev/null; echo "bench rc=$?"
grep "vgk_xdrop_band_align" $O/bench_xband.err | tail -18
python3 - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r04x'
d = json.loads(open(O + '/bench_xband.json').read().strip().split('\n')[-1])
print('xband %.2f M tails/s, %.2f ms/step, parity %s' % (d['value']/1e6, d['ms_per_step'], d.get('parity')))
PY
;;
y)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04y; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 400 python -m pytest tests/test_xdrop_band.py -x -q -m gpu > $O/pytest_xband.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest_xband.log
export VGAMD_XBAND_ONE_BATCH=1
B="python $GRAFT_REPO_ROOT/bench.py --workload xband --no-cpu --steps 3 --warmup 2"
( cd /tmp && timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_xband -o s -- $B > $O/stats_xband.log 2>&1 ) < /dev/null
f=$(find $O/stats_xband -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "xdrop_band" "$f" < /dev/null | cut -c1-120
;;
z)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04z; mkdir -p $O; export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_gssw_gpu_parity.py tests/test_reference_tap.py tests/test_chain_alignment.py tests/test_giraffe_stage.py tests/test_alignment_batch.py -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest.log
for v in spec two one; do
  unset VGAMD_WALK_ONE_PASS VGAMD_NO_SPEC_FILL; [ $v = one ] && export VGAMD_WALK_ONE_PASS=1; [ $v = two ] && export VGAMD_NO_SPEC_FILL=1
  timeout -s KILL 200 python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary > $O/bench_$v.json 2> $O/bench_$v.err < /dev/null; echo "bench rc=$?"
  timeout 30 python3 - <<PY
import json
d=json.loads(open("$O/bench_$v.json").read().strip().split("\n")[-1]); o=d["config"]["one_stream"]
print("$v headline %.2f M reads/s fill %.2f walk %.2f step %.2f ms parity %s e2e %s" % (d["value"]/1e6, o["fill_ms"], o["traceback_ms"], o["ms_per_step"], d["parity"], d.get("end_to_end_double_buffered_per_s")))
PY
done
unset VGAMD_WALK_ONE_PASS
B="python $GRAFT_REPO_ROOT/bench.py --workload linear --reads 400000 --no-cpu --no-e2e --no-secondary --steps 3 --warmup 1"
( cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/stats.log 2>&1 ) < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -h "gssw_walk\|gssw_fill" "$f" < /dev/null | cut -c1-120
;;
z2)
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
;;
*) echo "stages: a b c d e f g h i j k l last m n o p q r s t u v w w2 x y z z2"; exit 2 ;;
esac
