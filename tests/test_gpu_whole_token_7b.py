"""The fused whole-token path bench.py times (full_token: carried RMS norms, RoPE + kv-append as the QKV launch's epilogue,
split-KV attention with its fp16 shadow, residual adds as epilogues — 6 launches per layer) at the REAL Llama-2-7B widths
(d = 4096, 32 heads x 128, FFN 11008, context 2048) against an fp64 numpy model of the same layer built from the
oracle-dequantized weights: the layer output, the rows appended to the kv cache, and the carried statistics.  (Round 2
checked this path at 7B size only against its own unfused twin.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rms(x, g, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * g


def _rope(x, pos, base):  # x [heads][hs], mode 0 (adjacent pairs)
    h, hs = x.shape
    th = pos * (base ** (-2.0 / hs)) ** np.arange(hs // 2)
    c, s = np.cos(th), np.sin(th)
    out = x.copy()
    out[:, 0::2] = x[:, 0::2] * c - x[:, 1::2] * s
    out[:, 1::2] = x[:, 0::2] * s + x[:, 1::2] * c
    return out


def test_fused_whole_token_layer_at_7b_width_against_fp64_model(L, pkg, nso):
    import torch
    rng = np.random.default_rng(7)
    d, heads, hs, ff, ctx, eps, base = 4096, 32, 128, 11008, 2048, 1e-5, 10000.0
    n_past = ctx - 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def qweight(n, k):
        w = (rng.standard_normal((n, k)) * (1.0 / np.sqrt(k))).astype(np.float32)
        blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
        return pkg.Weight.from_host_blob(nso.ptr(blob), st), nso.unpack_fp32(blob).astype(np.float64), blob  # deq [k][n]

    wq, Wq, _0 = qweight(d, d)
    wk, Wk, _1 = qweight(d, d)
    wv, Wv, _2 = qweight(d, d)
    wo, Wo, _3 = qweight(d, d)
    w1, W1, _4 = qweight(ff, d)
    w3, W3, _5 = qweight(ff, d)
    w2, W2, _6 = qweight(d, ff)
    g1 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    g2 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    g3 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    x0 = rng.standard_normal((1, d)).astype(np.float32)
    kc_h = (rng.standard_normal((ctx + 8, heads, hs)) * 0.5).astype(np.float16)
    vc_h = (rng.standard_normal((ctx + 8, heads, hs)) * 0.5).astype(np.float16)

    dev = "cuda"
    f32 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
    f16 = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float16)
    dg1, dg2, dg3 = (torch.from_numpy(g).cuda() for g in (g1, g2, g3))
    dx0 = torch.from_numpy(x0).cuda()
    kc, vc = torch.from_numpy(kc_h).cuda().unsqueeze(0).contiguous(), torch.from_numpy(vc_h).cuda().unsqueeze(0).contiguous()
    parts = d // 16
    qkv, att, r1, t2, x = f32(3, 1, d), f32(1, d), f32(1, d), f32(1, ff), f32(1, d)
    x0h, atth, r1h, t2h, xh = f16(1, d), f16(1, d), f16(1, d), f16(1, ff), f16(1, d)
    ssq_a, ssq_b = f32(1, parts), f32(1, parts)
    tab = f32(1, hs // 2, 2)
    shape = pkg.AttnShape(1, heads, heads, hs, 1, ctx)
    ws = torch.zeros(int(L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device=dev)
    ck = pkg.check
    ck(L.ns_hip_norm_prep(1, d, dx0.data_ptr(), d, dg1.data_ptr(), x0h.data_ptr(), ssq_a.data_ptr(), parts, st))
    ck(L.ns_hip_rope_cos_sin(1, n_past, hs, base, 1.0, 1.0, tab.data_ptr(), st))
    lk = pkg.NormLink(ssq_a.data_ptr(), parts, parts, eps, d, None, None, 0)
    rp = pkg.QkvRope(kc.data_ptr(), vc.data_ptr(), tab.data_ptr(), heads, heads, hs, n_past, hs, 0, heads * hs, hs)
    ck(L.ns_hip_fusion_qkv_rope_forward_x(dx0.data_ptr(), x0h.data_ptr(), wq.h, wk.h, wv.h, qkv.data_ptr(), 1, d, d, C.byref(lk),
                                          C.byref(rp), st))
    a = pkg.attn_args(qkv[0].data_ptr(), kc.data_ptr(), vc.data_ptr(), att.data_ptr(), 1, heads, heads, hs, 1, n_past + 1, hs ** -0.5,
                      pkg.ATTN_CAUSAL)
    a.step_k_bs = a.step_v_bs = (ctx + 8) * heads * hs
    a.tmp = ws.data_ptr()
    ck(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), atth.data_ptr(), st))
    lk = pkg.NormLink(None, 0, 0, 0.0, 0, dg2.data_ptr(), ssq_b.data_ptr(), parts)
    ck(L.ns_hip_f32f32_forward_x(att.data_ptr(), atth.data_ptr(), wo.h, r1.data_ptr(), r1h.data_ptr(), 1, d, d, pkg.EPI_ADD, dx0.data_ptr(), d,
                                 C.byref(lk), st))
    lk = pkg.NormLink(ssq_b.data_ptr(), parts, parts, eps, d, None, None, 0)
    ck(L.ns_hip_fusion_ffn3_gateup_x(r1.data_ptr(), r1h.data_ptr(), w1.h, w3.h, None, t2.data_ptr(), t2h.data_ptr(), 1, pkg.EPI_SILU,
                                     C.byref(lk), st))
    lk = pkg.NormLink(None, 0, 0, 0.0, 0, dg3.data_ptr(), ssq_a.data_ptr(), parts)
    ck(L.ns_hip_f32f32_forward_x(t2.data_ptr(), t2h.data_ptr(), w2.h, x.data_ptr(), xh.data_ptr(), 1, ff, d, pkg.EPI_ADD, r1.data_ptr(), d,
                                 C.byref(lk), st))
    torch.cuda.synchronize()

    # ---- fp64 model of the same layer ----
    xr = x0.astype(np.float64)
    h = _rms(xr, g1, eps)
    q = _rope((h @ Wq).reshape(heads, hs), n_past, base)
    k_new = _rope((h @ Wk).reshape(heads, hs), n_past, base)
    v_new = (h @ Wv).reshape(heads, hs)
    K = kc_h[:ctx].astype(np.float64)
    V = vc_h[:ctx].astype(np.float64)
    K[n_past], V[n_past] = k_new, v_new
    o = np.zeros((heads, hs))
    for hh in range(heads):
        s = K[:, hh] @ q[hh] * hs ** -0.5
        p = np.exp(s - s.max())
        o[hh] = (p / p.sum()) @ V[:, hh]
    r1_ref = xr + o.reshape(1, d) @ Wo
    h2 = _rms(r1_ref, g2, eps)
    gt, up = h2 @ W1, h2 @ W3
    t2_ref = gt / (1.0 + np.exp(-gt)) * up
    x_ref = r1_ref + t2_ref @ W2

    rel = lambda got, ref: float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref))
    kc_out, vc_out = kc[0, n_past].float().cpu().numpy(), vc[0, n_past].float().cpu().numpy()
    assert rel(kc_out, k_new) < 1.5e-3 and rel(vc_out, v_new) < 1.5e-3      # fp16 cache rows (half an fp16 ulp on top of the GEMV)
    assert rel(att.cpu().numpy(), o.reshape(1, d)) < 2e-3                   # attention over 2048 fp16 positions
    assert rel(r1.cpu().numpy(), r1_ref) < 1e-3
    assert rel(t2.cpu().numpy(), t2_ref) < 2e-3
    assert rel(x.cpu().numpy(), x_ref) < 1e-3, rel(x.cpu().numpy(), x_ref)
    # the carried statistics the NEXT layer's QKV would divide by: sum of squares of the layer output per 16-column tile, and
    # the shadow gamma3 . x
    assert rel(ssq_a.cpu().numpy().sum(), (x_ref ** 2).sum()) < 2e-3
    assert rel(xh.float().cpu().numpy(), x_ref * g3) < 2e-3
    for w in (wq, wk, wv, wo, w1, w3, w2):
        w.free()
