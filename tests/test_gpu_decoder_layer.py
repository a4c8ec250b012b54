"""One Llama-style decoder layer (GQA, RoPE, fp16 kv-cache, SiLU FFN) executed entirely through the device-resident C
ABI — rmsnorm, quantized GEMMs, RoPE, fused attention, fused FFN, residual adds — against an fp64 numpy model of the
same layer built from the oracle-dequantized weights.  Prefill of a short prompt, then two decode steps on the cache."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rms(x, g, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * g


def _rope(x, pos0, base):  # x [seq][heads][hs], mode 0 (adjacent pairs), closed form
    seq, h, hs = x.shape
    out = x.copy()
    ts = base ** (-2.0 / hs)
    for i in range(seq):
        th = (pos0 + i) * ts ** np.arange(hs // 2)
        c, s = np.cos(th), np.sin(th)
        x0, x1 = x[i, :, 0::2], x[i, :, 1::2]
        out[i, :, 0::2] = x0 * c - x1 * s
        out[i, :, 1::2] = x0 * s + x1 * c
    return out


@pytest.mark.parametrize("fused", [False, True])
def test_llama_layer_prefill_then_decode(L, pkg, nso, fused):
    import torch
    rng = np.random.default_rng(2024)
    d, heads, hkv, hs, ff, ctx, eps, base = 512, 8, 4, 64, 1408, 64, 1e-5, 10000.0
    dkv = hkv * hs
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def qweight(n, k):
        w = (rng.standard_normal((n, k)) * (1.0 / np.sqrt(k))).astype(np.float32)
        blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
        return pkg.Weight.from_host_blob(nso.ptr(blob), st), nso.unpack_fp32(blob).astype(np.float64), blob  # deq is [k][n]

    wq, Wq, _b0 = qweight(d, d)
    wk, Wk, _b1 = qweight(dkv, d)
    wv, Wv, _b2 = qweight(dkv, d)
    wo, Wo, _b3 = qweight(d, d)
    w1, W1, _b4 = qweight(ff, d)
    w3, W3, _b5 = qweight(ff, d)
    w2, W2, _b6 = qweight(d, ff)
    g1 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    g2 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    dg1, dg2 = torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda()
    kc = torch.zeros((1, ctx, hkv, hs), dtype=torch.float16, device="cuda")
    vc = torch.zeros((1, ctx, hkv, hs), dtype=torch.float16, device="cuda")
    kc_ref = np.zeros((ctx, hkv, hs), np.float64)
    vc_ref = np.zeros((ctx, hkv, hs), np.float64)

    def gpu_layer_fused(x_np, n_past):
        """the same layer with the fused device operators: norm*gamma+fp16 shadow, RoPE(q,k)+kv-append, fp16 shadows
        between GEMMs, residual add as the down projection's epilogue"""
        m = x_np.shape[0]
        x = torch.from_numpy(x_np).cuda()
        f16 = lambda *s: torch.empty(*s, device="cuda", dtype=torch.float16)
        h, h16 = torch.empty_like(x), f16(m, d)
        pkg.check(L.ns_hip_norm_mul_h(m, d, True, eps, x.data_ptr(), dg1.data_ptr(), h.data_ptr(), h16.data_ptr(), st))
        q, k, v = torch.empty((m, d), device="cuda"), torch.empty((m, dkv), device="cuda"), torch.empty((m, dkv), device="cuda")
        for wt, out, n in ((wq, q, d), (wk, k, dkv), (wv, v, dkv)):
            pkg.check(L.ns_hip_f32f32_forward_h(h.data_ptr(), h16.data_ptr(), wt.h, out.data_ptr(), None, m, d, n,
                                                pkg.EPI_NONE, None, 0, st))
        pkg.check(L.ns_hip_rope_qkv_append(q.data_ptr(), k.data_ptr(), v.data_ptr(), kc.data_ptr(), vc.data_ptr(), m, heads, hkv,
                                           hs, n_past, hs, 0, base, 1.0, 0.0, 1.0, hkv * hs, hs, st))
        att = torch.empty((m, d), device="cuda")
        a = pkg.attn_args(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), att.data_ptr(), 1, heads, hkv, hs, m, n_past + m,
                          float(hs) ** -0.5, pkg.ATTN_CAUSAL)
        a.step_k_bs = a.step_v_bs = ctx * hkv * hs
        pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
        r1 = torch.empty((m, d), device="cuda")
        pkg.check(L.ns_hip_f32f32_forward(att.data_ptr(), wo.h, r1.data_ptr(), m, d, d, pkg.EPI_ADD, x.data_ptr(), d, st))
        h2, h216 = torch.empty_like(r1), f16(m, d)
        pkg.check(L.ns_hip_norm_mul_h(m, d, True, eps, r1.data_ptr(), dg2.data_ptr(), h2.data_ptr(), h216.data_ptr(), st))
        t2, t216 = torch.empty((m, ff), device="cuda"), f16(m, ff)
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(h2.data_ptr(), h216.data_ptr(), w1.h, w3.h, None, t2.data_ptr(), t216.data_ptr(),
                                                m, pkg.EPI_SILU, st))
        y = torch.empty((m, d), device="cuda")
        pkg.check(L.ns_hip_f32f32_forward_h(t2.data_ptr(), t216.data_ptr(), w2.h, y.data_ptr(), None, m, ff, d, pkg.EPI_ADD,
                                            r1.data_ptr(), d, st))
        torch.cuda.synchronize()
        return y.cpu().numpy()

    def gpu_layer(x_np, n_past):
        if fused:
            return gpu_layer_fused(x_np, n_past)
        m = x_np.shape[0]
        x = torch.from_numpy(x_np).cuda()
        h = torch.empty_like(x)
        pkg.check(L.ns_hip_layernormalization(m, d, True, eps, x.data_ptr(), h.data_ptr(), st))
        pkg.check(L.ns_hip_mul(m, d, h.data_ptr(), dg1.data_ptr(), 0, h.data_ptr(), st))
        q = torch.empty((m, d), device="cuda")
        k = torch.empty((m, dkv), device="cuda")
        v = torch.empty((m, dkv), device="cuda")
        for wt, out, n in ((wq, q, d), (wk, k, dkv), (wv, v, dkv)):
            pkg.check(L.ns_hip_f32f32_forward(h.data_ptr(), wt.h, out.data_ptr(), m, d, n, pkg.EPI_NONE, None, 0, st))
        pkg.check(L.ns_hip_rope_f32(q.data_ptr(), q.data_ptr(), 1, m, heads, hs, n_past, hs, 0, base, 1.0, 0.0, 1.0, st))
        pkg.check(L.ns_hip_rope_f32(k.data_ptr(), k.data_ptr(), 1, m, hkv, hs, n_past, hs, 0, base, 1.0, 0.0, 1.0, st))
        kc[0, n_past:n_past + m] = k.view(m, hkv, hs).half()  # kv-cache append (a copy; glue, not part of the ABI)
        vc[0, n_past:n_past + m] = v.view(m, hkv, hs).half()
        att = torch.empty((m, d), device="cuda")
        a = pkg.attn_args(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), att.data_ptr(), 1, heads, hkv, hs, m, n_past + m,
                          float(hs) ** -0.5, pkg.ATTN_CAUSAL)
        a.step_k_bs = a.step_v_bs = ctx * hkv * hs
        pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(a), st))
        r1 = torch.empty((m, d), device="cuda")  # x + attn * Wo   (custom::epilogue::Add)
        pkg.check(L.ns_hip_f32f32_forward(att.data_ptr(), wo.h, r1.data_ptr(), m, d, d, pkg.EPI_ADD, x.data_ptr(), d, st))
        h2 = torch.empty_like(r1)
        pkg.check(L.ns_hip_layernormalization(m, d, True, eps, r1.data_ptr(), h2.data_ptr(), st))
        pkg.check(L.ns_hip_mul(m, d, h2.data_ptr(), dg2.data_ptr(), 0, h2.data_ptr(), st))
        t2 = torch.empty((m, ff), device="cuda")
        y = torch.empty((m, d), device="cuda")
        pkg.check(L.ns_hip_fusion_ffn3_forward(h2.data_ptr(), w1.h, w2.h, w3.h, None, t2.data_ptr(), y.data_ptr(), m,
                                               pkg.EPI_SILU, st))
        pkg.check(L.ns_hip_add(m, d, y.data_ptr(), r1.data_ptr(), d, y.data_ptr(), st))
        torch.cuda.synchronize()
        return y.cpu().numpy()

    def ref_layer(x_np, n_past):
        m = x_np.shape[0]
        x = x_np.astype(np.float64)
        h = _rms(x, g1, eps)
        q = _rope((h @ Wq).reshape(m, heads, hs), n_past, base)
        k = _rope((h @ Wk).reshape(m, hkv, hs), n_past, base)
        v = (h @ Wv).reshape(m, hkv, hs)
        kc_ref[n_past:n_past + m] = k.astype(np.float16)
        vc_ref[n_past:n_past + m] = v.astype(np.float16)
        att = np.zeros((m, heads, hs))
        for i in range(m):
            nk = n_past + i + 1
            for hh in range(heads):
                s = kc_ref[:nk, hh // (heads // hkv)] @ q[i, hh] * hs ** -0.5
                p = np.exp(s - s.max())
                att[i, hh] = (p / p.sum()) @ vc_ref[:nk, hh // (heads // hkv)]
        r1 = x + att.reshape(m, d) @ Wo
        h2 = _rms(r1, g2, eps)
        a1 = h2 @ W1
        return r1 + ((a1 / (1 + np.exp(-a1))) * (h2 @ W3)) @ W2

    n_past = 0
    for m in (5, 1, 1):  # prompt of 5 tokens, then two single-token decode steps
        x = rng.standard_normal((m, d)).astype(np.float32)
        out, ref = gpu_layer(x, n_past), ref_layer(x, n_past)
        e = nso.rel_l2(out, ref)
        assert e < 2e-3, (m, n_past, e)  # fp16 activations / kv-cache across 7 GEMMs and the attention
        n_past += m
    for w in (wq, wk, wv, wo, w1, w3, w2):
        w.free()
