"""The RMS norm carried across operators (ns_norm_link, include/ns_bestla.h): the producer GEMV (residual add epilogue)
emits the gamma-scaled fp16 shadow and per-tile sums of squares, the consumer GEMV divides by the row's rms.  Checked
against numpy fp64 on the oracle-dequantized weights (rms_norm: ne_compute_forward_rms_norm_f32, ne_layers.c; then
ne_mul by gamma, then the GEMM), and against the unfused device operators."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _rms(x, g, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * g


def _w(pkg, nso, rng, n, k, st, qt=None, bs=32, asym=False):
    w = (rng.standard_normal((n, k)) * (1.0 / np.sqrt(k))).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4 if qt is None else qt, nso.BF16, asym, nso.CORE_AVX512_VNNI_KB)
    return pkg.Weight.from_host_blob(nso.ptr(blob), st), nso.unpack_fp32(blob).astype(np.float64), blob


@pytest.mark.parametrize("m,d,ff", [(1, 512, 1408), (3, 512, 1408), (16, 512, 1408), (1, 4096, 2816), (7, 4096, 2816),
                                    (1, 384, 640), (5, 384, 640), (16, 384, 640)])
def test_carried_rms_norm_chain(L, pkg, nso, m, d, ff):
    """x0 -> prep(g1) -> qkv_x ; attn stand-in -> wo_x(+x0, emits g2 shadow/ssq) -> gateup_x -> down_x(+r1, emits g3) ->
    head_x: every consumer against numpy fp64, and the shadows / partial sums against their definitions."""
    import torch
    rng = np.random.default_rng(d + m)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    eps = 1e-5
    wq, Wq, _0 = _w(pkg, nso, rng, d, d, st)
    wk, Wk, _1 = _w(pkg, nso, rng, d, d, st)
    wv, Wv, _2 = _w(pkg, nso, rng, d, d, st)
    wo, Wo, _3 = _w(pkg, nso, rng, d, d, st)
    w1, W1, _4 = _w(pkg, nso, rng, ff, d, st)
    w3, W3, _5 = _w(pkg, nso, rng, ff, d, st)
    w2, W2, _6 = _w(pkg, nso, rng, d, ff, st)
    wh, Wh, _7 = _w(pkg, nso, rng, 1000, d, st)
    g = [(1.0 + 0.2 * rng.standard_normal(d)).astype(np.float32) for _ in range(3)]
    dg = [torch.from_numpy(x).cuda() for x in g]
    x0 = (rng.standard_normal((m, d)) * 3.0).astype(np.float32)
    att = rng.standard_normal((m, d)).astype(np.float32)
    parts = (d + 15) // 16
    stride = (parts + 3) & ~3
    f32 = lambda *s: torch.zeros(*s, device="cuda")
    f16 = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.float16)
    dx0, datt = torch.from_numpy(x0).cuda(), torch.from_numpy(att).cuda()
    datt16 = datt.half()
    x16, ssq1 = f16(m, d), f32(m, stride)
    pkg.check(L.ns_hip_norm_prep(m, d, dx0.data_ptr(), d, dg[0].data_ptr(), x16.data_ptr(), ssq1.data_ptr(), stride, st))
    torch.cuda.synchronize()
    assert np.allclose(x16.float().cpu().numpy(), x0 * g[0], rtol=1e-3, atol=1e-3)
    assert np.allclose(ssq1.cpu().numpy()[:, :parts].sum(1), (x0.astype(np.float64) ** 2).sum(1), rtol=1e-5)

    # qkv consumes (x16, ssq1)
    qkv = f32(3, m, d)
    lk = pkg.NormLink(ssq1.data_ptr(), parts, stride, eps, d, None, None, 0)
    pkg.check(L.ns_hip_fusion_qkv_forward_x(dx0.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv.data_ptr(), None, m, d, d,
                                            C.byref(lk), st))
    torch.cuda.synchronize()
    h = _rms(x0.astype(np.float64), g[0], eps)
    for i, W in enumerate((Wq, Wk, Wv)):
        assert nso.rel_l2(qkv[i].cpu().numpy(), h @ W) < TOL, ("qkv", i)
    # the unfused device operators give the same thing (different rounding points: shadow of the normalised row)
    hn, hn16, qkv_u = f32(m, d), f16(m, d), f32(3, m, d)
    pkg.check(L.ns_hip_norm_mul_h(m, d, True, eps, dx0.data_ptr(), dg[0].data_ptr(), hn.data_ptr(), hn16.data_ptr(), st))
    pkg.check(L.ns_hip_fusion_qkv_forward_h(hn.data_ptr(), hn16.data_ptr(), wq.h, wk.h, wv.h, qkv_u.data_ptr(), None, m, d, d, st))
    torch.cuda.synchronize()
    assert nso.rel_l2(qkv.cpu().numpy(), qkv_u.cpu().numpy()) < TOL

    # wo: r1 = att @ Wo + x0, emits g2-scaled shadow + partial sums
    r1, r1_16, ssq2 = f32(m, d), f16(m, d), f32(m, stride)
    lk = pkg.NormLink(None, 0, 0, 0.0, 0, dg[1].data_ptr(), ssq2.data_ptr(), stride)
    pkg.check(L.ns_hip_f32f32_forward_x(datt.data_ptr(), datt16.data_ptr(), wo.h, r1.data_ptr(), r1_16.data_ptr(), m, d, d,
                                        pkg.EPI_ADD, dx0.data_ptr(), d, C.byref(lk), st))
    torch.cuda.synchronize()
    r1_ref = att.astype(np.float64) @ Wo + x0
    assert nso.rel_l2(r1.cpu().numpy(), r1_ref) < TOL
    r1_np = r1.cpu().numpy()
    assert np.allclose(r1_16.float().cpu().numpy(), r1_np * g[1], rtol=2e-3, atol=2e-3)
    assert np.allclose(ssq2.cpu().numpy()[:, :parts].sum(1), (r1_np.astype(np.float64) ** 2).sum(1), rtol=1e-5)

    # gate/up consumes (r1_16, ssq2)
    t2, t2_16 = f32(m, ff), f16(m, ff)
    lk = pkg.NormLink(ssq2.data_ptr(), parts, stride, eps, d, None, None, 0)
    pkg.check(L.ns_hip_fusion_ffn3_gateup_x(r1.data_ptr(), r1_16.data_ptr(), w1.h, w3.h, None, t2.data_ptr(), t2_16.data_ptr(),
                                            m, pkg.EPI_SILU, C.byref(lk), st))
    torch.cuda.synchronize()
    h2 = _rms(r1_np.astype(np.float64), g[1], eps)
    a1 = h2 @ W1
    t2_ref = (a1 / (1 + np.exp(-a1))) * (h2 @ W3)
    assert nso.rel_l2(t2.cpu().numpy(), t2_ref) < TOL

    # down: x = t2 @ W2 + r1, emits g3 shadow + sums; the head consumes them
    x, x_16, ssq3 = f32(m, d), f16(m, d), f32(m, stride)
    lk = pkg.NormLink(None, 0, 0, 0.0, 0, dg[2].data_ptr(), ssq3.data_ptr(), stride)
    pkg.check(L.ns_hip_f32f32_forward_x(t2.data_ptr(), t2_16.data_ptr(), w2.h, x.data_ptr(), x_16.data_ptr(), m, ff, d,
                                        pkg.EPI_ADD, r1.data_ptr(), d, C.byref(lk), st))
    logits = f32(m, 1000)
    lk = pkg.NormLink(ssq3.data_ptr(), parts, stride, eps, d, None, None, 0)
    pkg.check(L.ns_hip_f32f32_forward_x(x.data_ptr(), x_16.data_ptr(), wh.h, logits.data_ptr(), None, m, d, 1000,
                                        pkg.EPI_NONE, None, 0, C.byref(lk), st))
    torch.cuda.synchronize()
    x_np = x.cpu().numpy()
    assert nso.rel_l2(x_np, t2.cpu().numpy().astype(np.float64) @ W2 + r1_np) < TOL
    assert nso.rel_l2(logits.cpu().numpy(), _rms(x_np.astype(np.float64), g[2], eps) @ Wh) < TOL

    # run-to-run identical (fixed summation order)
    logits2 = f32(m, 1000)
    pkg.check(L.ns_hip_f32f32_forward_x(x.data_ptr(), x_16.data_ptr(), wh.h, logits2.data_ptr(), None, m, d, 1000,
                                        pkg.EPI_NONE, None, 0, C.byref(lk), st))
    torch.cuda.synchronize()
    assert torch.equal(logits, logits2)


def test_norm_link_is_refused_outside_its_envelope(L, pkg, nso):
    import torch
    rng = np.random.default_rng(5)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = 256
    w, _, _b = _w(pkg, nso, rng, d, d, st)
    a = torch.zeros(32, d, device="cuda")
    a16 = a.half()
    c = torch.zeros(32, d, device="cuda")
    ssq = torch.zeros(32, 16, device="cuda")
    lk = pkg.NormLink(ssq.data_ptr(), 16, 16, 1e-5, d, None, None, 0)
    # more than 16 rows
    assert L.ns_hip_f32f32_forward_x(a.data_ptr(), a16.data_ptr(), w.h, c.data_ptr(), None, 32, d, d, 0, None, 0, C.byref(lk), st) != 0
    # no fp16 shadow
    assert L.ns_hip_f32f32_forward_x(a.data_ptr(), None, w.h, c.data_ptr(), None, 1, d, d, 0, None, 0, C.byref(lk), st) != 0
    assert b"norm link" in L.ns_hip_last_error()
    # misaligned partial sums
    lk2 = pkg.NormLink(ssq.data_ptr() + 4, 16, 16, 1e-5, d, None, None, 0)
    assert L.ns_hip_f32f32_forward_x(a.data_ptr(), a16.data_ptr(), w.h, c.data_ptr(), None, 1, d, d, 0, None, 0, C.byref(lk2), st) != 0
    # a NULL link is the plain operator
    assert L.ns_hip_f32f32_forward_x(a.data_ptr(), a16.data_ptr(), w.h, c.data_ptr(), None, 1, d, d, 0, None, 0, None, st) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("m,n_past", [(1, 0), (1, 77), (4, 30)])
@pytest.mark.parametrize("heads,hkv,hs", [(8, 8, 64), (8, 2, 128)])
def test_qkv_with_rope_and_kv_append_epilogue(L, pkg, nso, m, n_past, heads, hkv, hs):
    """ns_hip_fusion_qkv_rope_forward_x == ns_hip_fusion_qkv_forward_h followed by ns_hip_rope_qkv_append, bit for bit
    (q in place, k / v in the fp16 cache), with a carried norm on the input."""
    import torch
    rng = np.random.default_rng(heads * hs + m)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d, dkv, ctx, eps = heads * hs, hkv * hs, 128, 1e-5
    wq, Wq, _0 = _w(pkg, nso, rng, d, d, st)
    wk, Wk, _1 = _w(pkg, nso, rng, dkv, d, st)
    wv, Wv, _2 = _w(pkg, nso, rng, dkv, d, st)
    gam = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(d)).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
    parts = d // 16
    x16, ssq = torch.zeros(m, d, device="cuda", dtype=torch.float16), torch.zeros(m, parts, device="cuda")
    pkg.check(L.ns_hip_norm_prep(m, d, x.data_ptr(), d, gam.data_ptr(), x16.data_ptr(), ssq.data_ptr(), parts, st))
    lk = pkg.NormLink(ssq.data_ptr(), parts, parts, eps, d, None, None, 0)
    ldc = d  # the three outputs are laid out [3][m][ldc]
    # two steps: separate operators
    qkv_a = torch.zeros(3, m, ldc, device="cuda")
    kc_a = torch.zeros(1, ctx, hkv, hs, device="cuda", dtype=torch.float16)
    vc_a = torch.zeros_like(kc_a)
    pkg.check(L.ns_hip_fusion_qkv_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_a.data_ptr(), None, m, d, ldc,
                                            C.byref(lk), st))
    q_a = qkv_a[0].contiguous()
    k_a = qkv_a[1][:, :dkv].contiguous()
    v_a = qkv_a[2][:, :dkv].contiguous()
    pkg.check(L.ns_hip_rope_qkv_append(q_a.data_ptr(), k_a.data_ptr(), v_a.data_ptr(), kc_a.data_ptr(), vc_a.data_ptr(), m, heads,
                                       hkv, hs, n_past, hs, 0, 10000.0, 1.0, 0.0, 1.0, hkv * hs, hs, st))
    # one launch
    qkv_b = torch.zeros(3, m, ldc, device="cuda")
    kc_b, vc_b = torch.zeros_like(kc_a), torch.zeros_like(kc_a)
    tab = torch.zeros(m, hs // 2, 2, device="cuda")
    pkg.check(L.ns_hip_rope_cos_sin(m, n_past, hs, 10000.0, 1.0, 1.0, tab.data_ptr(), st))
    rp = pkg.QkvRope(kc_b.data_ptr(), vc_b.data_ptr(), tab.data_ptr(), heads, hkv, hs, n_past, hs, 0, hkv * hs, hs)
    pkg.check(L.ns_hip_fusion_qkv_rope_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_b.data_ptr(), m, d, ldc,
                                                 C.byref(lk), C.byref(rp), st))
    torch.cuda.synchronize()
    assert torch.equal(qkv_b[0], q_a)
    assert torch.equal(kc_b, kc_a) and torch.equal(vc_b, vc_a)
    assert torch.count_nonzero(kc_b[0, n_past:n_past + m]) > 0 and torch.count_nonzero(kc_b[0, :n_past]) == 0
    # and against fp64: rms norm -> GEMM -> rope (closed form)
    h = _rms(x.cpu().numpy().astype(np.float64), gam.cpu().numpy(), eps)
    kr = (h @ Wk).reshape(m, hkv, hs)
    ts = 10000.0 ** (-2.0 / hs)
    ref = kr.copy()
    for i in range(m):
        th = (n_past + i) * ts ** np.arange(hs // 2)
        c, s = np.cos(th), np.sin(th)
        ref[i, :, 0::2] = kr[i, :, 0::2] * c - kr[i, :, 1::2] * s
        ref[i, :, 1::2] = kr[i, :, 0::2] * s + kr[i, :, 1::2] * c
    assert nso.rel_l2(kc_b[0, n_past:n_past + m].float().cpu().numpy(), ref) < 2e-3
    # refused: NeoX mode, wrong head geometry
    rp2 = pkg.QkvRope(kc_b.data_ptr(), vc_b.data_ptr(), tab.data_ptr(), heads, hkv, hs, n_past, hs, 2, hkv * hs, hs)
    assert L.ns_hip_fusion_qkv_rope_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_b.data_ptr(), m, d, ldc,
                                              C.byref(lk), C.byref(rp2), st) != 0
    rp3 = pkg.QkvRope(kc_b.data_ptr(), vc_b.data_ptr(), tab.data_ptr(), heads + 1, hkv, hs, n_past, hs, 0, hkv * hs, hs)
    assert L.ns_hip_fusion_qkv_rope_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_b.data_ptr(), m, d, ldc,
                                              C.byref(lk), C.byref(rp3), st) != 0
    L.ns_hip_reset_error()


@pytest.mark.parametrize("m,heads,hkv,hs,qt", [(300, 8, 8, 128, None), (2048, 8, 2, 128, None), (77, 16, 4, 64, "s8"), (130, 4, 4, 96, None)])
def test_qkv_rope_cache_append_as_the_tiled_gemms_epilogue(L, pkg, nso, m, heads, hkv, hs, qt):
    """ns_hip_fusion_qkv_rope_forward_x at PREFILL size (round 5): the fused-QKV launch of the tiled GEMM rotates q and k and writes k / v to the fp16 cache
    in its epilogue — the same bits as three GEMMs + ns_hip_rope_qkv_append, with and without the fp32 k / v tensors (NS_QKV_ROPE_KV_CACHE_ONLY)."""
    import torch
    rng = np.random.default_rng(m + hs)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d, dkv, n_past = heads * hs, hkv * hs, 19
    if dkv % 128 or d % 128:
        pytest.skip("matrix widths must be whole 128-column blocks")
    ctx = n_past + m + 5
    q8 = nso.S8 if qt == "s8" else None
    wq, _a, _0 = _w(pkg, nso, rng, d, d, st, q8)
    wk, Wk, _1 = _w(pkg, nso, rng, dkv, d, st, q8)
    wv, _b, _2 = _w(pkg, nso, rng, dkv, d, st, q8)
    x = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
    x16 = x.half()
    # separate operators: the fused-QKV GEMM launch (the same tiles, never split along K), then RoPE + append on packed copies of its outputs
    ldc = d
    qkv_a = torch.zeros(3, m, ldc, device="cuda")
    pkg.check(L.ns_hip_fusion_qkv_forward_h(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_a.data_ptr(), None, m, d, ldc, st))
    q_a, k_a, v_a = qkv_a[0].contiguous(), qkv_a[1][:, :dkv].contiguous(), qkv_a[2][:, :dkv].contiguous()
    v_raw = v_a.clone()
    kc_a = torch.zeros(1, ctx, hkv, hs, device="cuda", dtype=torch.float16)
    vc_a = torch.zeros_like(kc_a)
    pkg.check(L.ns_hip_rope_qkv_append(q_a.data_ptr(), k_a.data_ptr(), v_a.data_ptr(), kc_a.data_ptr(), vc_a.data_ptr(), m, heads, hkv, hs, n_past, hs, 0,
                                       10000.0, 1.0, 0.0, 1.0, hkv * hs, hs, st))
    tab = torch.zeros(m, hs // 2, 2, device="cuda")
    pkg.check(L.ns_hip_rope_cos_sin(m, n_past, hs, 10000.0, 1.0, 1.0, tab.data_ptr(), st))
    for flags in (0, 1):
        qkv_b = torch.full((3, m, ldc), 7.0, device="cuda")
        kc_b, vc_b = torch.zeros_like(kc_a), torch.zeros_like(kc_a)
        rp = pkg.QkvRope(kc_b.data_ptr(), vc_b.data_ptr(), tab.data_ptr(), heads, hkv, hs, n_past, hs, 0, hkv * hs, hs, flags)
        pkg.check(L.ns_hip_fusion_qkv_rope_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_b.data_ptr(), m, d, ldc, None, C.byref(rp), st))
        torch.cuda.synchronize()
        assert torch.equal(qkv_b[0], q_a), flags
        assert torch.equal(kc_b, kc_a) and torch.equal(vc_b, vc_a), flags
        assert torch.count_nonzero(kc_b[0, n_past:n_past + m]) > 0 and torch.count_nonzero(kc_b[0, :n_past]) == 0
        if flags == 0:  # k comes out rotated, v as it is — the fp32 tensors a graph may read
            kr = kc_a[0, n_past:n_past + m].reshape(m, dkv).float()
            assert torch.equal(qkv_b[1][:, :dkv].half().float(), kr) and torch.equal(qkv_b[2][:, :dkv], v_raw)
        else:
            assert bool((qkv_b[1] == 7.0).all()) and bool((qkv_b[2] == 7.0).all())
    # against fp64: GEMM -> rope (closed form)
    kr = (x16.float().cpu().numpy().astype(np.float64) @ Wk).reshape(m, hkv, hs)
    ts = 10000.0 ** (-2.0 / hs)
    ref = kr.copy()
    for i in range(m):
        th = (n_past + i) * ts ** np.arange(hs // 2)
        c, s = np.cos(th), np.sin(th)
        ref[i, :, 0::2] = kr[i, :, 0::2] * c - kr[i, :, 1::2] * s
        ref[i, :, 1::2] = kr[i, :, 0::2] * s + kr[i, :, 1::2] * c
    assert nso.rel_l2(kc_b[0, n_past:n_past + m].float().cpu().numpy(), ref) < 2e-3
    # refused at this size: NeoX pairs (they are 64 columns apart: other lanes' columns)
    rp2 = pkg.QkvRope(kc_b.data_ptr(), vc_b.data_ptr(), tab.data_ptr(), heads, hkv, hs, n_past, hs, 2, hkv * hs, hs, 0)
    assert L.ns_hip_fusion_qkv_rope_forward_x(x.data_ptr(), x16.data_ptr(), wq.h, wk.h, wv.h, qkv_b.data_ptr(), m, d, ldc, None, C.byref(rp2), st) != 0
    L.ns_hip_reset_error()


def test_norm_with_the_fp16_result_only_and_gemms_on_fp16_only_activations(L, pkg, nso):
    """Prefill plumbing (round 5): ns_hip_norm_mul_h with dOut = NULL writes the fp16 shadow alone; the fused gate/up and QKV + RoPE launches of the tiled GEMM
    take dA = NULL (fp16 activations as they are) — the same bits as with the fp32 tensors present."""
    import torch
    rng = np.random.default_rng(77)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    m, d, ff, heads, hs = 200, 512, 1408, 4, 128
    gam = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(d)).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()
    h32, h16a, h16b = torch.zeros(m, d, device="cuda"), torch.zeros(m, d, device="cuda", dtype=torch.float16), torch.zeros(m, d, device="cuda", dtype=torch.float16)
    pkg.check(L.ns_hip_norm_mul_h(m, d, True, 1e-5, x.data_ptr(), gam.data_ptr(), h32.data_ptr(), h16a.data_ptr(), st))
    pkg.check(L.ns_hip_norm_mul_h(m, d, True, 1e-5, x.data_ptr(), gam.data_ptr(), None, h16b.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(h16a, h16b) and torch.equal(h16a, h32.half())
    w1, _a, _0 = _w(pkg, nso, rng, ff, d, st)
    w3, _b, _1 = _w(pkg, nso, rng, ff, d, st)
    t_a, t_b = torch.zeros(m, ff, device="cuda", dtype=torch.float16), torch.zeros(m, ff, device="cuda", dtype=torch.float16)
    pkg.check(L.ns_hip_fusion_ffn3_gateup_h(h32.data_ptr(), h16a.data_ptr(), w1.h, w3.h, None, None, t_a.data_ptr(), m, pkg.EPI_SILU, st))
    pkg.check(L.ns_hip_fusion_ffn3_gateup_h(None, h16a.data_ptr(), w1.h, w3.h, None, None, t_b.data_ptr(), m, pkg.EPI_SILU, st))
    torch.cuda.synchronize()
    assert torch.equal(t_a, t_b) and torch.count_nonzero(t_b) > 0
    wq, _c, _2 = _w(pkg, nso, rng, d, d, st)
    wk, _d, _3 = _w(pkg, nso, rng, d, d, st)
    wv, _e, _4 = _w(pkg, nso, rng, d, d, st)
    tab = torch.zeros(m, hs // 2, 2, device="cuda")
    pkg.check(L.ns_hip_rope_cos_sin(m, 0, hs, 10000.0, 1.0, 1.0, tab.data_ptr(), st))
    outs = []
    for a32 in (h32.data_ptr(), None):
        q = torch.zeros(3, m, d, device="cuda")
        kc, vc = torch.zeros(1, m, heads, hs, device="cuda", dtype=torch.float16), torch.zeros(1, m, heads, hs, device="cuda", dtype=torch.float16)
        rp = pkg.QkvRope(kc.data_ptr(), vc.data_ptr(), tab.data_ptr(), heads, heads, hs, 0, hs, 0, heads * hs, hs, 1)
        pkg.check(L.ns_hip_fusion_qkv_rope_forward_x(a32, h16a.data_ptr(), wq.h, wk.h, wv.h, q.data_ptr(), m, d, d, None, C.byref(rp), st))
        torch.cuda.synchronize()
        outs.append((q[0].clone(), kc, vc))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and torch.count_nonzero(outs[1][1]) > 0
    # fp16-only activations at DECODE size have no kernel that takes them: refused, not guessed
    assert L.ns_hip_fusion_ffn3_gateup_h(None, h16a.data_ptr(), w1.h, w3.h, None, None, t_b.data_ptr(), 1, pkg.EPI_SILU, st) != 0
    L.ns_hip_reset_error()
