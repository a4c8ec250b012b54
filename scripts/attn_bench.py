#!/usr/bin/env python3
"""Decode attention (sl_q = 1) of the fused attention operator on Llama-2-7B / Mistral-7B shapes: us and KV GB/s."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
for name, hn, hkv, hs, ctx in (("llama2-7b ctx2048", 32, 32, 128, 2048), ("llama2-7b ctx512", 32, 32, 128, 512),
                               ("mistral-7b ctx4096 (GQA 8)", 32, 8, 128, 4096)):
    q = torch.randn(1, 1, hn, hs, device="cuda")
    k = torch.randn(1, ctx, hkv, hs, device="cuda", dtype=torch.float16)
    v = torch.randn(1, ctx, hkv, hs, device="cuda", dtype=torch.float16)
    d = torch.zeros_like(q)
    a = pkg.attn_args(q.data_ptr(), k.data_ptr(), v.data_ptr(), d.data_ptr(), 1, hn, hkv, hs, 1, ctx, hs ** -0.5, 1)
    f = lambda: pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    byt = 2 * ctx * hkv * hs * 2
    res[name] = {"us": round(us, 2), "kv_bytes": byt, "GBps": round(byt / us / 1e3, 1)}
print(json.dumps(res, indent=1))

# ---- prefill (sl_q = sl_kv, causal): the generic kernel, no matrix cores yet ----
for name, hn, hkv, hs, sl in (("llama2-7b prefill 512", 32, 32, 128, 512), ("llama2-7b prefill 2048", 32, 32, 128, 2048)):
    q = torch.randn(1, sl, hn, hs, device="cuda")
    k = torch.randn(1, sl, hkv, hs, device="cuda", dtype=torch.float16)
    v = torch.randn(1, sl, hkv, hs, device="cuda", dtype=torch.float16)
    d = torch.zeros_like(q)
    a = pkg.attn_args(q.data_ptr(), k.data_ptr(), v.data_ptr(), d.data_ptr(), 1, hn, hkv, hs, sl, sl, hs ** -0.5, 1)
    f = lambda: pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    flops = 2.0 * 2 * hn * hs * sl * sl / 2   # causal: half of QK^T and PV
    print(json.dumps({name: {"ms": round(ms, 3), "TFLOPS_causal": round(flops / ms / 1e9, 2)}}))
