#!/usr/bin/env python3
"""Whole Llama-2-7B decode step, device resident, in one HIP graph: per layer rmsnorm*g, fused QKV, RoPE(q, k), kv-cache
append, fused attention, WO + residual, rmsnorm*g, fused FFN, residual; then final norm + lm_head.  Int4 g32 weights,
fp16 kv-cache, batch 1.  Reports tokens/s at a few context lengths next to the GEMM-only chain of bench.py (everything
but the GEMMs is SURVEY §8f territory: first versions, untuned)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
d, ff, heads, hs, vocab, nl = 4096, 11008, 32, 128, 32000, 32
ctx_max = 2048 + 8


def make(n, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * (k ** -0.5)
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt


layers = []
for i in range(nl):
    layers.append(dict(q=make(d, d, 10 * i + 1), k=make(d, d, 10 * i + 2), v=make(d, d, 10 * i + 3), o=make(d, d, 10 * i + 4),
                       w1=make(ff, d, 10 * i + 5), w3=make(ff, d, 10 * i + 6), w2=make(d, ff, 10 * i + 7),
                       g1=torch.ones(d, device="cuda"), g2=torch.ones(d, device="cuda"),
                       kc=torch.randn((1, ctx_max, heads, hs), device="cuda").half(),
                       vc=torch.randn((1, ctx_max, heads, hs), device="cuda").half()))
head = make(vocab, d, 999)
gf = torch.ones(d, device="cuda")
shape = pkg.AttnShape(1, heads, heads, hs, 1, ctx_max)
attn_ws = torch.empty(L.bestla_fusion_attn_workspace_size(C.byref(shape)), dtype=torch.uint8, device="cuda")
x0 = torch.randn(1, d, device="cuda")
f16 = lambda *shape: torch.empty(*shape, device="cuda", dtype=torch.float16)
sh = dict(h=f16(1, d), qkv=f16(3, d), h2=f16(1, d), t2=f16(1, ff))  # fp16 shadows between operators
bufs = dict(h=torch.empty(1, d, device="cuda"), qkv=torch.empty(3, d, device="cuda"), att=torch.empty(1, d, device="cuda"),
            r1=torch.empty(1, d, device="cuda"), h2=torch.empty(1, d, device="cuda"), t2=torch.empty(1, ff, device="cuda"),
            y=torch.empty(1, d, device="cuda"), x=torch.empty(1, d, device="cuda"), logits=torch.empty(1, vocab, device="cuda"))


def step(n_past):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    b = bufs
    xin = x0
    for lw in layers:
        pkg.check(L.ns_hip_norm_mul_h(1, d, True, 1e-5, xin.data_ptr(), lw["g1"].data_ptr(), b["h"].data_ptr(), sh["h"].data_ptr(), st))
        pkg.check(L.ns_hip_fusion_qkv_forward_h(b["h"].data_ptr(), sh["h"].data_ptr(), lw["q"].h, lw["k"].h, lw["v"].h,
                                                b["qkv"].data_ptr(), None, 1, d, d, st))
        q, k, v = b["qkv"][0], b["qkv"][1], b["qkv"][2]
        pkg.check(L.ns_hip_rope_qkv_append(q.data_ptr(), k.data_ptr(), v.data_ptr(), lw["kc"].data_ptr(), lw["vc"].data_ptr(), 1,
                                           heads, heads, hs, n_past, hs, 0, 10000.0, 1.0, 0.0, 1.0, heads * hs, hs, st))
        a = pkg.attn_args(q.data_ptr(), lw["kc"].data_ptr(), lw["vc"].data_ptr(), b["att"].data_ptr(), 1, heads, heads, hs, 1,
                          n_past + 1, hs ** -0.5, pkg.ATTN_CAUSAL)
        a.step_k_bs = a.step_v_bs = ctx_max * heads * hs
        a.tmp = attn_ws.data_ptr()  # caller-provided workspace (mha_dense.h contract): valid under graph capture
        pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
        pkg.check(L.ns_hip_f32f32_forward(b["att"].data_ptr(), lw["o"].h, b["r1"].data_ptr(), 1, d, d, pkg.EPI_ADD, xin.data_ptr(), d, st))
        pkg.check(L.ns_hip_norm_mul_h(1, d, True, 1e-5, b["r1"].data_ptr(), lw["g2"].data_ptr(), b["h2"].data_ptr(), sh["h2"].data_ptr(), st))
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(b["h2"].data_ptr(), sh["h2"].data_ptr(), lw["w1"].h, lw["w3"].h, None,
                                                b["t2"].data_ptr(), sh["t2"].data_ptr(), 1, pkg.EPI_SILU, st))
        # down projection with the residual add as its epilogue (custom::epilogue::Add)
        pkg.check(L.ns_hip_f32f32_forward_h(b["t2"].data_ptr(), sh["t2"].data_ptr(), lw["w2"].h, b["x"].data_ptr(), None, 1, ff, d,
                                            pkg.EPI_ADD, b["r1"].data_ptr(), d, st))
        xin = b["x"]
    pkg.check(L.ns_hip_norm_mul_h(1, d, True, 1e-5, xin.data_ptr(), gf.data_ptr(), b["h"].data_ptr(), sh["h"].data_ptr(), st))
    pkg.check(L.ns_hip_f32f32_forward_h(b["h"].data_ptr(), sh["h"].data_ptr(), head.h, b["logits"].data_ptr(), None, 1, d, vocab,
                                        pkg.EPI_NONE, None, 0, st))


res = {}
for n_past in (127, 511, 2047):
    for _ in range(2):
        step(n_past)  # also sizes the attention scratch outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(n_past)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    res["ctx_%d" % (n_past + 1)] = {"ms_per_token": round(ms, 4), "tokens_per_s": round(1000.0 / ms, 1)}
print(json.dumps(res, indent=1))
