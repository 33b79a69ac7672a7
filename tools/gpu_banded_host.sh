# the host half of vgk_banded_align by host threads (tools/banded_host_threads.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/banded_host
nproc > gpurun_out/banded_host/nproc.txt
timeout -s KILL 400 python tools/banded_host_threads.py 100000 > gpurun_out/banded_host/threads.txt 2> gpurun_out/banded_host/threads.err
cat gpurun_out/banded_host/threads.txt; grep -A9 "threads" gpurun_out/banded_host/threads.err | cut -c1-100 | tail -70
