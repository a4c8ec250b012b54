#!/usr/bin/env python3
"""Causal prefill attention alone (attn_mfma_kernel): Llama-2-7B heads (32 x 128), fp16 K / V, fp32 Q / output, one
layer's call at a few prompt lengths; TFLOPS counts the causal half (2 x 2 x heads x hs x sl^2 / 2).
Usage: scripts/attn_prefill_bench.py [sl ...]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
heads, hs = int(os.environ.get("NS_ATTN_BENCH_HEADS", "32")), int(os.environ.get("NS_ATTN_BENCH_HS", "128"))  # GPT-J: 16 x 256
res = {}
for sl in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]:
    q = torch.randn((1, sl, heads, hs), device="cuda")
    k = torch.randn((1, sl, heads, hs), device="cuda").half()
    v = torch.randn((1, sl, heads, hs), device="cuda").half()
    out = torch.zeros_like(q)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = pkg.attn_args(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1, heads, heads, hs, sl, sl, hs ** -0.5, pkg.ATTN_CAUSAL)
    run = lambda: pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res["sl_%d" % sl] = {"ms": round(ms, 4), "tflops_causal": round(2.0 * 2 * heads * hs * sl * sl / 2 / ms / 1e9, 1)}
print(json.dumps(res))
