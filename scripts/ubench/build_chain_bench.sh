#!/bin/bash
# builds scripts/ubench/chain_bench against the in-tree libns_hip.so (run from the repo root)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/ubench/chain_bench.cpp \
  -Lneural-speed_amd -lns_hip -Wl,-rpath,'$ORIGIN/../../neural-speed_amd' -o scripts/ubench/chain_bench
