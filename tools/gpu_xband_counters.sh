cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/xbandpmc
VGAMD_XBAND_SCORES_ONLY=1 timeout -s KILL 300 python bench.py --workload xband --no-cpu > gpurun_out/xbandpmc/scores_only.json 2> gpurun_out/xbandpmc/scores_only.err
python -c "
import json
d=json.loads(open('gpurun_out/xbandpmc/scores_only.json').read().strip().splitlines()[-1]); print('scores only: kernel ms', d['band']['fill_kernel_ms'], 'per s', round(d['value']))"
WORKLOAD=xband READS=200000 KERNELS=xdrop_band bash tools/pmc_gapless.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM" "FETCH_SIZE" "WRITE_SIZE" | cut -c1-300
