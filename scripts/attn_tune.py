#!/usr/bin/env python3
"""Whole-token throughput (bench.py full_token, context 2048) over the split rule of the decode attention kernel:
ns_hip_set_tuning("attn_wg_target" / "attn_min_keys").  Usage: scripts/attn_tune.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import torch
import bench
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
chain = bench.Chain(pkg, bench.CFG["n_layer"], 0, 1, keep_host_layer=False)
L.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
res = []
for wg, mk in [(1024, 128), (512, 128), (256, 128), (2048, 128), (1024, 64), (2048, 64), (1024, 256), (512, 256), (4096, 32)]:
    L.ns_hip_set_tuning(b"attn_wg_target", wg); L.ns_hip_set_tuning(b"attn_min_keys", mk)
    ft, _ = bench.full_token(chain, pkg, 2048, fused=True)
    res.append({"wg_target": wg, "min_keys": mk, "tokens_per_s": ft["tokens_per_s"], "ms": ft["ms_per_token"]})
    print(res[-1], flush=True)
print(json.dumps(res))
