#!/bin/bash
# Diagnostic variants of the decode engine, cross-compiled HERE: scripts/build_engine_variants.sh name:"-DFLAG" ...
#   -> variants/<name>/libns_hip.so ; run with LD_LIBRARY_PATH=variants/<name> scripts/ubench/engine_bench ...
set -e
cd "$(dirname "$0")/../neural-speed_amd/csrc"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  mkdir -p ../../variants/$name
  /opt/rocm/bin/hipcc -O3 -std=c++20 $flags -fPIC --offload-arch=gfx950 -c ns_engine.hip -o /tmp/ns_engine_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/$name/libns_hip.so ns_api.o ns_blob.o ns_tp.o ns_kernels.o ns_gemv.o \
     ns_gemm.o ns_attn.o ns_quant.o ns_p2p.o ns_i8ref.o ns_moe.o /tmp/ns_engine_$name.o -ldl
  echo built variants/$name/libns_hip.so
done
