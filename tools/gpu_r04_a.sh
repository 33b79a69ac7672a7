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
