#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the C ABI (device code is not
# instrumented) and of the C++ facade test, which drives reset / step / state / statistics calls
# against the oracle.  Build here (no GPU needed):  tools/asan_host_check.sh build
# Run on a GPU box:                                  tools/asan_host_check.sh run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/tools/_asan
HIPCC=/opt/rocm/bin/hipcc
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
if [ "$1" = "build" ]; then
  mkdir -p $OUT
  objs=""
  for f in gx_api.cpp gx_classic.hip gx_acrobot.hip gx_classic64.hip gx_acrobot64.hip gx_tabular.hip gx_blackjack.hip gx_pong.hip gx_breakout.hip gx_invaders.hip gx_stats.hip gx_hostpath.hip; do
    x=""; [ "${f##*.}" = "cpp" ] && x="-x hip"
    $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $SAN $x -c $R/gym_amd/csrc/$f -o $OUT/${f%.*}.o &
    objs="$objs $OUT/${f%.*}.o"
  done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -o $OUT/libgymx.so $objs -ldl
  $HIPCC -x c++ -std=c++17 -O1 -g -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer -I$R/include -I$R \
    $R/tests/cpp/host_facade_test.cpp -o $OUT/host_facade_test -L$OUT -lgymx -L$R/oracle/_build -lgym_oracle \
    -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../oracle/_build' -Wl,-rpath,/opt/rocm/lib
  echo built $OUT/host_facade_test
else
  # the HIP runtime keeps allocations for the life of the process: leak checking off
  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $OUT/host_facade_test
fi
