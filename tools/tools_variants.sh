#!/bin/bash
# builds variant libraries: tools_variants.sh name "-Dflags" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p gym_amd/lib/variants
name=$1; shift
flags="$@"
objs=""
for f in gx_classic.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $flags -c gym_amd/csrc/$f -o gym_amd/lib/variants/${name}_${f%.*}.o
  objs="$objs gym_amd/lib/variants/${name}_${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gym_amd/lib/variants/libgymx_$name.so $objs gym_amd/lib/obj/gx_api.o gym_amd/lib/obj/gx_acrobot64.o gym_amd/lib/obj/gx_tabular.o gym_amd/lib/obj/gx_pong.o gym_amd/lib/obj/gx_stats.o -ldl
echo built $name
