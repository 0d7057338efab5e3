#!/bin/bash
# Builds a variant of libgymx.so for A/B runs: tools/tools_variants.sh <name> [-Dflags...]
# Sources are taken from $VAR_SRC (default: gym_amd/csrc); only gx_classic.hip is rebuilt.
set -e
cd "$(dirname "$0")/.."
SRC=${VAR_SRC:-gym_amd/csrc}
mkdir -p gym_amd/lib/variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c $SRC/gx_classic.hip -o /tmp/variant_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gym_amd/lib/variants/libgymx_$name.so /tmp/variant_${name}.o gym_amd/lib/obj/gx_api.o gym_amd/lib/obj/gx_acrobot64.o gym_amd/lib/obj/gx_tabular.o gym_amd/lib/obj/gx_pong.o gym_amd/lib/obj/gx_stats.o -ldl
echo built $name
