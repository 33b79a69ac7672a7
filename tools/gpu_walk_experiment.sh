# walk-kernel experiments: each variant is another build of the engine library (tools/build_variant.sh), timed on the headline batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/walkexp
for v in ${VARIANTS:-base}; do
  if [ $v = base ]; then unset VGAMD_ENGINE_LIB; else export VGAMD_ENGINE_LIB=$PWD/build/variants/libvgamd_$v.so; fi
  timeout -s KILL 200 python bench.py --no-cpu --no-e2e --steps 5 --warmup 2 > gpurun_out/walkexp/walk_$v.json 2> gpurun_out/walkexp/walk_$v.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/walkexp/walk_$v.json').read().strip().splitlines()[-1]); o=d['config']['one_stream']; print('$v', o, 'failed', d.get('problems_failed'))"
done
