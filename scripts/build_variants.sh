#!/bin/bash
# Build diagnostic variants of libns_hip.so HERE (cross-compile) so GPU-box minutes are not spent compiling:
#   scripts/build_variants.sh name1:"-DFLAG=1" name2:"-DFLAG=2" ...   ->  variants/libns_hip_<name>.so
set -e
cd "$(dirname "$0")/../neural-speed_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  # FILES = which kernel sources get the flags (the others are linked from the regular build)
  objs=""
  for f in ns_kernels ns_gemv ns_gemvs ns_gemm ns_attn; do
    if [[ " ${FILES:-ns_kernels ns_gemv ns_gemm} " == *" $f "* ]]; then
      /opt/rocm/bin/hipcc -O3 -std=c++20 $flags -fPIC --offload-arch=gfx950 -c $f.hip -o /tmp/${f}_$name.o &
      objs="$objs /tmp/${f}_$name.o"
    else
      objs="$objs $f.o"
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libns_hip_$name.so ns_api.o ns_blob.o ns_split.o ns_tp.o ns_route.o $objs ns_quant.o ns_p2p.o ns_i8ref.o ns_i8g2_n4.o ns_i8g2_n2.o ns_i8g2_n1.o ns_i8g2_b2.o ns_i8g2_b1.o ns_moe.o ns_device.o -ldl
  echo built variants/libns_hip_$name.so
done
