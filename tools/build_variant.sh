#!/bin/bash
# Build another copy of the engine library with extra -D flags, for kernel experiments on the GPU box:
#   tools/build_variant.sh NAME -DVGK_TB_TILE=8 ...   ->  build/variants/libvgamd_NAME.so   (select with VGAMD_ENGINE_LIB)
# build/ is git-ignored; the .so travels with gpurun.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
out=build/variants/$name; mkdir -p $out
for f in vg_amd/csrc/*.cpp; do g++ -O3 -std=c++17 -fPIC -Iinclude -Wall "$@" -c $f -o $out/$(basename $f .cpp).o & done
for f in vg_amd/csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Wno-unused-function -Wno-unused-value "$@" -c $f -o $out/$(basename $f .hip)_hip.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libvgamd_$name.so $out/*.o
echo build/variants/libvgamd_$name.so
