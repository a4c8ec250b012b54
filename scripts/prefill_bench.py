#!/usr/bin/env python3
"""Prefill GEMM throughput alone (one Llama-2-7B layer's seven GEMMs at M rows, int4 and int8 weights): bench.py's
prefill legs without the decode chain.  Usage: scripts/prefill_bench.py [M ...]   (default 2048)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
torch.cuda.set_device(0)
chain = bench.Chain(pkg, 1, 0, 1)
res = {}
for m in [int(a) for a in sys.argv[1:]] or [2048]:
    res["m%d" % m] = {"int4w_tflops": bench.prefill_tflops(chain, pkg, m), "int8w_tflops": bench.prefill_tflops_int8w(chain, pkg, m)}
print(json.dumps(res))
