#!/bin/bash
# Build diagnostic variants of libns_hip.so HERE (cross-compile) so GPU-box minutes are not spent compiling:
#   scripts/build_variants.sh name1:"-DFLAG=1" name2:"-DFLAG=2" ...   ->  variants/libns_hip_<name>.so
set -e
cd "$(dirname "$0")/../neural-speed_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++20 $flags -fPIC --offload-arch=gfx950 -c ns_kernels.hip -o /tmp/nsk_$name.o &
  /opt/rocm/bin/hipcc -O3 -std=c++20 $flags -fPIC --offload-arch=gfx950 -c ns_decode.hip -o /tmp/nsd_$name.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libns_hip_$name.so ns_api.o ns_blob.o /tmp/nsk_$name.o /tmp/nsd_$name.o ns_quant.o
  echo built variants/libns_hip_$name.so
done
