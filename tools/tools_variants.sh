#!/bin/bash
# Builds a variant of libgymx.so for A/B runs: tools/tools_variants.sh <name> <source file> [-Dflags...]
# Sources are taken from $VAR_SRC (default: gym_amd/csrc); only the named file is rebuilt, the
# other objects come from the last regular build.  The library lands in /tmp/gymx_variants (copy
# it under gpurun_in/ or similar to take it to a GPU box).
set -e
cd "$(dirname "$0")/.."
SRC=${VAR_SRC:-gym_amd/csrc}
OUT=${VAR_OUT:-gym_amd/lib/variants}
mkdir -p "$OUT"
name=$1; file=$2; shift 2
base=$(basename "$file" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c "$SRC/$file" -o /tmp/variant_${name}.o
objs=""
for o in gym_amd/lib/obj/*.o; do
  if [ "$(basename "$o" .o)" != "$base" ]; then objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgymx_$name.so" /tmp/variant_${name}.o $objs -ldl
echo built $name
