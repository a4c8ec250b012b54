#!/usr/bin/env python3
"""Decode attention (split + merge), Llama-2-7B shape: which part of its 11 us is the ACCESS PATTERN?  32 layers' worth of distinct fp16
caches in one HIP graph, per-call microseconds, for
  layout  position-major [ctx][heads][hs] (a head's rows are 256-byte pieces 8 KB apart: bench.py's full_token so far)
          head-major     [heads][ctx][hs] (one contiguous slab per head: the library-managed cache of mha_dense.h:124-172)
  order   context ranges of one head first (blockIdx.x = range) / heads first (ns_hip_set_tuning("attn_heads_first", 1))
Usage: scripts/r05/attn_layout_ab.py [ctx]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
heads = hkv = 32; hs = 128; nl = 32
q = torch.randn((1, 1, heads, hs), device="cuda")
out = torch.zeros_like(q)
out16 = torch.zeros((1, 1, heads, hs), device="cuda", dtype=torch.float16)
shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
bytes_per_call = 2 * ctx * hkv * hs * 2
res = []
for layout in ("position-major", "head-major"):
    shp = (1, ctx, hkv, hs) if layout == "position-major" else (1, hkv, ctx, hs)
    kc = [torch.randn(shp, device="cuda").half() for _ in range(nl)]
    vc = [torch.randn(shp, device="cuda").half() for _ in range(nl)]
    for order in (0, 1):
        L.ns_hip_set_tuning(b"attn_heads_first", order)

        def step():
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for il in range(nl):
                a = pkg.attn_args(q.data_ptr(), kc[il].data_ptr(), vc[il].data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
                if layout == "head-major":
                    a.step_k_head_num = a.step_v_head_num = ctx * hs
                    a.step_k_sl = a.step_v_sl = hs
                a.tmp = ws.data_ptr()
                pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), out16.data_ptr(), st))
        step(); torch.cuda.synchronize()
        ref = out.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 / nl * 1e3
        res.append({"layout": layout, "heads_first": order, "us_per_call": round(us, 2), "TBps": round(bytes_per_call / us / 1e6, 2),
                    "finite": bool(torch.isfinite(out).all().item())})
    del kc, vc
L.ns_hip_set_tuning(b"attn_heads_first", -1)
print(json.dumps({"ctx": ctx, "heads": heads, "head_size": hs, "cases": res}))
