#!/usr/bin/env python3
"""Causal prefill attention WITH ALiBi (MPT-7B heads: 32 x 128) — the biased form of the 128-row kernel against what served such
calls before (NS_ATTN_MFMA2_ROWS huge: the one-row-per-workgroup generic kernel).  Usage: attn_alibi_prefill_bench.py [sl ...]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
heads, hs = 32, 128
res = {}
for sl in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    q = torch.randn((1, sl, heads, hs), device="cuda")
    k = torch.randn((1, sl, heads, hs), device="cuda").half()
    v = torch.randn((1, sl, heads, hs), device="cuda").half()
    out = torch.zeros_like(q)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = pkg.attn_args(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1, heads, heads, hs, sl, sl, hs ** -0.5, 3)
    run = lambda: pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
    row = {}
    for name, rows in (("generic_kernel", 1 << 30), ("biased_128_row_kernel", 0)):
        L.ns_hip_set_tuning(b"attn_mfma2_rows", rows)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        row[name] = {"ms": round(ms, 4), "tflops_causal": round(2.0 * 2 * heads * hs * sl * sl / 2 / ms / 1e9, 1)}
    res["sl_%d" % sl] = row
print(json.dumps(res))
