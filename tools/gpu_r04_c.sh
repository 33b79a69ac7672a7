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
