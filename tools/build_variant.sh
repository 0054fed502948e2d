#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG | -fflag ...]  -> stereo_toolbox_amd/lib/libstx_hip_NAME.so: another build of the same
# sources for A/B measurements (STX_HIP_LIB=<path> selects it; lib/ is git-ignored but travels to the GPU box).
# A -ffp-contract=... argument replaces the default (fast).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p stereo_toolbox_amd/lib/obj_$name
C="-ffp-contract=fast"
for a in "$@"; do case $a in -ffp-contract=*) C=$a;; esac; done
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $C -munsafe-fp-atomics -Wno-unused-result -I stereo_toolbox_amd/csrc"
for s in stereo_toolbox_amd/csrc/*.hip; do
  o=stereo_toolbox_amd/lib/obj_$name/$(basename ${s%.hip}).o
  /opt/rocm/bin/hipcc $F "$@" -c $s -o $o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC stereo_toolbox_amd/lib/obj_$name/*.o -o stereo_toolbox_amd/lib/libstx_hip_$name.so
echo stereo_toolbox_amd/lib/libstx_hip_$name.so
