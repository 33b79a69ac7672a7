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
