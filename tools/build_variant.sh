#!/bin/bash
# Alternative build of libo2m_hip.so for same-box A/B runs (O2M_HIP_LIB=<path>):  tools/build_variant.sh <out.so> [-DFLAG=..] ...
out=$1; shift
src=one_to_many_gan_amd/csrc
mkdir -p "$(dirname "$out")" build/variants/obj
objs=""
for f in conv_igemm conv_direct conv_wgrad pointwise style ada; do
  o=build/variants/obj/$(basename "$out" .so)_$f.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value "$@" -c $src/$f.hip -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs && echo "built $out"
