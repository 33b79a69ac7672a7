#!/bin/bash
# The emulator library built with AddressSanitizer, and a pytest selection run against it: the lane code the kernels share with the emulator
# (gssw_device.hpp, banded_geom_device.hpp, ...) and the host passes around them, checked for reads and writes outside their arenas.
# usage: tools/emu_asan.sh tests/test_gssw_emu_parity.py -k "spec or dags"      (test infrastructure only; restores the normal library)
#        SAN=undefined tools/emu_asan.sh ...   the same under UndefinedBehaviorSanitizer (alignment and vptr checks off: the lane code reads
#        packed arenas through casts by design); its reports go to $OUT/ubsan.log.* — none is the expected result
# (tests that expect an exception out of the host shim abort under the preloaded ASan runtime — its __cxa_throw interceptor finds no real one
# in an uninstrumented library — deselect them: -k "not shim")
set -e
SAN=${SAN:-address}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=${TMPDIR:-/tmp}/vgamd_$SAN; mkdir -p $OUT
FLAGS="-fsanitize=$SAN"; [ $SAN = undefined ] && FLAGS="$FLAGS -fno-sanitize=alignment,vptr"; RT=$([ $SAN = undefined ] && echo libubsan.so || echo libasan.so)
ls $ROOT/vg_amd/csrc/*.cpp $ROOT/tests/emu/backend_emu.cpp | xargs -P 8 -I{} sh -c "g++ -O1 -g -std=c++17 -fPIC $FLAGS -fno-omit-frame-pointer -I$ROOT/include -I$ROOT/vg_amd/csrc -c {} -o $OUT/\$(basename {} .cpp).o"
g++ -shared -fsanitize=$SAN -o $OUT/libvgamd_emu_asan.so $OUT/*.o -lpthread
make -s -C $ROOT emu
cp $ROOT/tests/emu/libvgamd_emu.so $OUT/emu_backup.so
trap 'cp $OUT/emu_backup.so $ROOT/tests/emu/libvgamd_emu.so' EXIT
cp $OUT/libvgamd_emu_asan.so $ROOT/tests/emu/libvgamd_emu.so; touch $ROOT/tests/emu/libvgamd_emu.so
cd $ROOT && ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/ubsan.log LD_PRELOAD=$(gcc -print-file-name=$RT) python -m pytest -x -q -m "not gpu" -p no:cacheprovider "$@"
