# round 4, GPU call J: the paired-end slice — parity test and its bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; mkdir -p $O
timeout -s KILL 600 python -m pytest tests/test_paired_stage.py tests/test_aligner_client.py tests/test_rescue_fixups.py -m gpu -q -x > $O/pytest_j.log 2>&1; echo "pytest rc=$?" >> $O/pytest_j.log; tail -3 $O/pytest_j.log
timeout -s KILL 400 python bench.py --workload paired --steps 3 --warmup 1 > $O/bench_paired.json 2> $O/bench_paired.err; echo "rc=$?"; tail -c 300 $O/bench_paired.err
python3 -c "
import json
d=json.loads(open('$O/bench_paired.json').read().strip().split('\n')[-1]); c=d['config']
print('%.3g %s  %.1f ms/step  rescued %d (positive %d)  parity %s  cpu %s' % (d['value'], d['unit'], d['ms_per_step'], c['pairs_rescued'], c['rescued_with_positive_score'], d['parity'], d['cpu_baseline']['value']))
print(c['stage_ms'])"
