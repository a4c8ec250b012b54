#!/usr/bin/env python3
"""Decode attention at 2048 positions: K / V cold (32 distinct caches, 1 GB: every byte from HBM) against hot (ONE cache read 32 times: 33.5 MB,
resident in the 256 MB Infinity Cache) — what a prefetch of a layer's K / V under the QKV projection could buy at most."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
heads = hkv = 32; hs = 128; nl = 32
q = torch.randn((1, 1, heads, hs), device="cuda"); out = torch.zeros_like(q)
shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
for distinct in (32, 1, 4):
    kc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(distinct)]
    vc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(distinct)]
    def step():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for il in range(nl):
            a = pkg.attn_args(q.data_ptr(), kc[il % distinct].data_ptr(), vc[il % distinct].data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
            a.tmp = ws.data_ptr()
            pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), None, st))
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"ctx": ctx, "distinct_caches": distinct, "MB": distinct * 2 * ctx * hkv * hs * 2 / 1e6, "us_per_call": round(e0.elapsed_time(e1) / 30 / nl * 1e3, 2)}), flush=True)
    del kc, vc
