#!/bin/bash
# builds scripts/ubench/chain_bench and scripts/ubench/engine_bench against the in-tree libns_hip.so
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/ubench/chain_bench.cpp \
  -Lneural-speed_amd -lns_hip -Wl,-rpath,'$ORIGIN/../../neural-speed_amd' -o scripts/ubench/chain_bench
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/ubench/engine_bench.cpp \
  -Lneural-speed_amd -lns_hip -Wl,-rpath,'$ORIGIN/../../neural-speed_amd' -o scripts/ubench/engine_bench
