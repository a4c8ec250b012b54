#!/usr/bin/env python3
"""Decode attention (split + merge) alone: 32 layers' worth of distinct fp16 kv-caches (HBM-bound, not cache-bound), one
HIP graph, per-call microseconds and effective TB/s for a sweep of the context-split rule.
Usage: scripts/attn_bench.py [ctx] [heads] [heads_kv] [head_size]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hkv = int(sys.argv[3]) if len(sys.argv) > 3 else heads
hs = int(sys.argv[4]) if len(sys.argv) > 4 else 128
nl = 32
kc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
vc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
q = torch.randn((1, 1, heads, hs), device="cuda")
out = torch.zeros_like(q)
out16 = torch.zeros((1, 1, heads, hs), device="cuda", dtype=torch.float16)
shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
bytes_per_call = 2 * ctx * hkv * hs * 2
res = []
for target, mk in [(1024, 128), (1024, 64), (2048, 64), (2048, 128), (512, 128), (1024, 256), (4096, 32)]:
    L.ns_hip_set_tuning(b"attn_wg_target", target)
    L.ns_hip_set_tuning(b"attn_min_keys", mk)
    ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")

    def step():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for il in range(nl):
            a = pkg.attn_args(q.data_ptr(), kc[il].data_ptr(), vc[il].data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx,
                              hs ** -0.5, pkg.ATTN_CAUSAL)
            a.tmp = ws.data_ptr()
            pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), out16.data_ptr(), st))
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 / nl * 1e3
    res.append({"wg_target": target, "min_keys": mk, "us_per_call": round(us, 2), "TBps": round(bytes_per_call / us / 1e6, 2)})
print(json.dumps({"ctx": ctx, "heads": heads, "heads_kv": hkv, "head_size": hs, "sweep": res}, indent=1))
