cd $GRAFT_REPO_ROOT
for T in 16 32 64 128; do
  echo "== VGAMD_HOST_THREADS=$T"
  VGAMD_HOST_THREADS=$T python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('linear pack_s', d.get('pack_seconds'), d['value'])"
  VGAMD_HOST_THREADS=$T python bench.py --workload banded --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('banded e2e', [v for k,v in d['config'].items() if 'end_to_end' in k], d['value'])"
  VGAMD_HOST_THREADS=$T python bench.py --workload gapless --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gapless e2e', [v for k,v in d['config'].items() if 'end_to_end' in k], d['value'])"
  VGAMD_HOST_THREADS=$T python bench.py --workload wfa --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wfa e2e', [v for k,v in d['config'].items() if 'end_to_end' in k], d['value'])"
done
