# the X-drop band kernel built for 2 / 3 / 4 wavefronts per SIMD (VGK_XB_OCC; tools/build_variant.sh xb3 -DVGK_XB_OCC=3 ...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/xband
timeout -s KILL 300 python -m pytest tests/test_xdrop_band.py -m gpu -x -q 2>&1 | tail -2
timeout -s KILL 500 python tools/xband_variants.py 200000 default > gpurun_out/xband/variants.txt 2> gpurun_out/xband/variants.err
cat gpurun_out/xband/variants.txt; tail -3 gpurun_out/xband/variants.err
