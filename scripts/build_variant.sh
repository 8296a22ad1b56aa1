#!/bin/bash
# usage: build_variant.sh NAME "-DFA_BLOCK2D=256 ..."   -> variants/NAME.so (experiment builds, not shipped)
set -e
cd "$(dirname "$0")/../falcon_amd/csrc"
out=../../variants/$1.so; mkdir -p ../../variants
objs=""
for f in fa_api fa_constraints fa_patches fa_tiles fa_kernels_quad4 fa_kernels_tri3 fa_kernels_hex8 fa_kernels_ho2d fa_kernels_hex20 fa_kernels_emat; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics $2 -c $f.hip -o /tmp/v_$1_$f.o 2>/dev/null &
  objs="$objs /tmp/v_$1_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
echo built $out
