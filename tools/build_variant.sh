#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]  -> variants/libstx_NAME.so (tuning A/B builds; git-ignored)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants/obj_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -Wno-unused-result -I stereo_toolbox_amd/csrc"
for s in stereo_toolbox_amd/csrc/*.hip; do
  o=variants/obj_$name/$(basename ${s%.hip}).o
  if [ "$(basename $s)" = "conv3d.hip" ] || [ ! -f $o ]; then /opt/rocm/bin/hipcc $F "$@" -c $s -o $o 2>/dev/null & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/obj_$name/*.o -o variants/libstx_$name.so
echo variants/libstx_$name.so
