#!/bin/bash
# The host half of the library (scene builders, GLB reader, BVH builder and its thread pool, controller, C wrappers)
# under AddressSanitizer + UBSan, then ThreadSanitizer, on the CPU: the host tests and a build of the 1.07 M-triangle
# scene's tree.  The GPU half is compiled as usual and never launched here.   tools/host_sanitizers.sh
set -e
cd "$(dirname "$0")/.."
FLAGS="-std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter -fno-omit-frame-pointer"
GCCLIB() { g++ -print-file-name=$1; }
big_scene='from raytracer_amd import capi
from raytracer_amd.renderer import Scene
s = Scene(capi.RT_SCENE_TEAPOT_GRID); print(s.bvh_info(), s.wide_bvh_info(), hex(s.bvh_hash()))'
make -C raytracer_amd/csrc -j8 LIBNAME=libraytracer_amd_asan.so OBJDIR=build_asan CXXFLAGS="$FLAGS -fsanitize=address,undefined" > /dev/null
LD_PRELOAD="$(GCCLIB libasan.so) $(GCCLIB libubsan.so) $LD_PRELOAD" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  RT_LIB_PATH=$PWD/raytracer_amd/lib/libraytracer_amd_asan.so python -m pytest tests -x -q -m "not gpu"
LD_PRELOAD="$(GCCLIB libasan.so) $(GCCLIB libubsan.so) $LD_PRELOAD" ASAN_OPTIONS=detect_leaks=0 RT_LIB_PATH=$PWD/raytracer_amd/lib/libraytracer_amd_asan.so python -c "$big_scene"
make -C raytracer_amd/csrc -j8 LIBNAME=libraytracer_amd_tsan.so OBJDIR=build_tsan CXXFLAGS="$FLAGS -fsanitize=thread" > /dev/null
LD_PRELOAD="$(GCCLIB libtsan.so) $LD_PRELOAD" TSAN_OPTIONS="report_signal_unsafe=0" RT_LIB_PATH=$PWD/raytracer_amd/lib/libraytracer_amd_tsan.so python -c "$big_scene"
rm -rf raytracer_amd/csrc/build_asan raytracer_amd/csrc/build_tsan raytracer_amd/lib/libraytracer_amd_asan.so raytracer_amd/lib/libraytracer_amd_tsan.so
echo "host sanitizers: clean"
