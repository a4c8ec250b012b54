#!/usr/bin/env python3
"""bench.py's prefill leg (one layer's seven GEMMs at M = 2048, int4 g32) under ns_hip_set_tuning("g3_bm", v):
0 automatic, 258 automatic without the tall wave tiles, 128 / 256 / 257 forced.  Same process, same weights."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import torch
import bench
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
chain = bench.Chain(pkg, 2, 0, 1, keep_host_layer=False)
L.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
out = []
for rep in range(2):
    for v in (0, 258, 128, 257):
        L.ns_hip_set_tuning(b"g3_bm", v)
        out.append({"g3_bm": v, "tflops": bench.prefill_tflops(chain, pkg)})
        print(out[-1], flush=True)
L.ns_hip_set_tuning(b"g3_bm", 0)
print(json.dumps(out))
