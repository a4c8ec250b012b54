"""RoPE (ne_compute_forward_rope_f32, ne_layers.c:9243-9428): the oracle's restatement against an fp64 closed form on
CPU, and the GPU kernel against the oracle."""
import ctypes as C

import numpy as np
import pytest


def _closed_form(x, n_past, n_dims, mode, base, fscale, attn):
    b, s, h, hs = x.shape
    y = x.astype(np.float64).copy()
    xs = x.astype(np.float64)
    ts = float(np.float32(base)) ** (-2.0 / n_dims)
    for i2 in range(s):
        p = float(n_past + i2)
        if mode == 0:
            idx = np.arange(hs // 2)
            th = fscale * p * ts ** idx
            c, sn = np.cos(th) * attn, np.sin(th) * attn
            x0, x1 = xs[:, i2, :, 0::2], xs[:, i2, :, 1::2]
            y[:, i2, :, 0::2] = x0 * c - x1 * sn
            y[:, i2, :, 1::2] = x0 * sn + x1 * c
        else:
            k = 0
            for ib in range(hs // n_dims):
                for ic in range(0, n_dims, 2):
                    th = fscale * (p * fscale) * ts ** k   # the reference applies freq_scale twice in this branch
                    k += 1
                    i0 = ib * n_dims + ic // 2
                    x0, x1 = xs[:, i2, :, i0], xs[:, i2, :, i0 + n_dims // 2]
                    y[:, i2, :, i0] = x0 * np.cos(th) * attn - x1 * np.sin(th) * attn
                    y[:, i2, :, i0 + n_dims // 2] = x0 * np.sin(th) * attn + x1 * np.cos(th) * attn
    return y


CASES = [(1, 1, 32, 128, 17, 128, 0, 10000.0, 1.0, 1.0), (2, 5, 4, 64, 0, 64, 2, 10000.0, 1.0, 1.0),
         (1, 3, 8, 128, 2000, 64, 2, 1000000.0, 0.25, 1.3), (1, 4, 2, 80, 9, 80, 0, 10000.0, 0.5, 1.0)]


@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,fscale,attn", CASES)
def test_rope_oracle_matches_closed_form(nso, b, s, h, hs, n_past, n_dims, mode, base, fscale, attn):
    x = np.random.default_rng(hs + n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = _closed_form(x, n_past, n_dims, mode, base, fscale, attn)
    out = nso.rope_f32(x, n_past, n_dims, mode, base, fscale, attn)
    # sequential fp32 theta products + fp32 sin/cos at |theta| up to a few thousand: ~1e-4 absolute at worst
    assert np.max(np.abs(out - ref)) < 2e-3
    assert nso.rel_l2(out, ref) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,fscale,attn", CASES)
def test_rope_gpu_matches_oracle(L, pkg, nso, b, s, h, hs, n_past, n_dims, mode, base, fscale, attn):
    import torch
    x = np.random.default_rng(hs * 3 + n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = nso.rope_f32(x, n_past, n_dims, mode, base, fscale, attn)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros_like(dx)
    pkg.check(L.ns_hip_rope_f32(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, fscale, 0.0, attn, st))
    pkg.check(L.ns_hip_rope_f32(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, fscale, 0.0, attn, st))
    torch.cuda.synchronize()
    out = dy.cpu().numpy()
    assert np.array_equal(out, dx.cpu().numpy())  # in place == out of place
    # same theta bit for bit; only the device sinf / cosf differ from the host libm
    assert np.max(np.abs(out - ref)) < 1e-5 * max(1.0, float(np.abs(x).max()))
    # unsupported modes are refused loudly
    assert L.ns_hip_rope_f32(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, 4, base, fscale, 0.0, attn, st) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2])
def test_rope_qkv_append_equals_separate_ops(L, pkg, nso, mode):
    import torch
    seq, heads, hkv, hs, n_past, ctx = 3, 8, 2, 64, 5, 16
    rng = np.random.default_rng(mode + 1)
    q = rng.standard_normal((1, seq, heads, hs)).astype(np.float32)
    k = rng.standard_normal((1, seq, hkv, hs)).astype(np.float32)
    v = rng.standard_normal((1, seq, hkv, hs)).astype(np.float32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dq, dk, dv = torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
    rq, rk = dq.clone(), dk.clone()
    pkg.check(L.ns_hip_rope_f32(rq.data_ptr(), rq.data_ptr(), 1, seq, heads, hs, n_past, hs, mode, 10000.0, 1.0, 0.0, 1.0, st))
    pkg.check(L.ns_hip_rope_f32(rk.data_ptr(), rk.data_ptr(), 1, seq, hkv, hs, n_past, hs, mode, 10000.0, 1.0, 0.0, 1.0, st))
    kc = torch.full((ctx, hkv, hs), 9.0, dtype=torch.float16, device="cuda")
    vc = torch.full((ctx, hkv, hs), 9.0, dtype=torch.float16, device="cuda")
    pkg.check(L.ns_hip_rope_qkv_append(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), kc.data_ptr(), vc.data_ptr(), seq, heads,
                                       hkv, hs, n_past, hs, mode, 10000.0, 1.0, 0.0, 1.0, hkv * hs, hs, st))
    torch.cuda.synchronize()
    assert torch.equal(dq, rq)
    assert torch.equal(kc[n_past:n_past + seq], rk[0].half())
    assert torch.equal(vc[n_past:n_past + seq], dv[0].half())
    assert torch.all(kc[:n_past] == 9.0) and torch.all(vc[n_past + seq:] == 9.0)  # nothing else touched


@pytest.mark.gpu
@pytest.mark.parametrize("mode,hs,n_dims,fscale,attn", [(0, 128, 128, 1.0, 1.0), (2, 64, 64, 0.5, 1.25), (2, 128, 64, 1.0, 1.0), (0, 80, 80, 1.0, 1.0)])
def test_rope_qkv_append_prompt_sized_form_equals_the_row_by_row_form(L, pkg, nso, mode, hs, n_dims, fscale, attn):
    """from 16 rows on the call takes the table + streaming kernels (round 4); one row at a time it takes the per-pair kernel:
    the same bits in q, the K cache and the V cache, GQA and a strided cache included; head size 80 (pairs not a multiple of 4
    per 16 bytes of cache row) stays on the per-pair kernel and must agree with itself"""
    import torch
    seq, heads, hkv, n_past, ctx = 37, 8, 2, 11, 64
    g = torch.Generator(device="cuda").manual_seed(hs + mode)
    q = torch.randn((1, seq, heads, hs), generator=g, device="cuda")
    k = torch.randn((1, seq, hkv, hs), generator=g, device="cuda")
    v = torch.randn((1, seq, hkv, hs), generator=g, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = 8  # the cache rows carry 8 unused halves per head: strides differ from the dense ones
    def run(rows_per_call):
        qq = q.clone()
        kc = torch.full((ctx, hkv, hs + pad), 9.0, dtype=torch.float16, device="cuda")
        vc = torch.full((ctx, hkv, hs + pad), 9.0, dtype=torch.float16, device="cuda")
        for r0 in range(0, seq, rows_per_call):
            n = min(rows_per_call, seq - r0)
            pkg.check(L.ns_hip_rope_qkv_append(qq[0, r0].data_ptr(), k[0, r0].data_ptr(), v[0, r0].data_ptr(), kc.data_ptr(), vc.data_ptr(), n,
                                               heads, hkv, hs, n_past + r0, n_dims, mode, 10000.0, fscale, 0.0, attn, hkv * (hs + pad), hs + pad, st))
        torch.cuda.synchronize()
        return qq, kc, vc
    a, b = run(seq), run(1)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.all(a[1][:, :, hs:] == 9.0) and torch.all(a[1][:n_past] == 9.0) and torch.all(a[2][n_past + seq:] == 9.0)
    assert torch.equal(a[2][n_past:n_past + seq, :, :hs], v[0].half())


# ---------------------------------------------------------------------------------------------- YaRN extrapolation mix
def _yarn_closed_form(x, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow):
    """rope_yarn (ne_layers.c:9196-9217) in fp64, written independently of the oracle's loop structure"""
    b, s, h, hs = x.shape
    xs = x.astype(np.float64)
    y = xs.copy()
    ts = float(np.float32(base)) ** (-2.0 / n_dims)

    def corr_dim(n_rot):
        return n_dims * np.log(n_orig / (n_rot * 2 * np.pi)) / (2 * np.log(base))
    lo = max(0.0, np.floor(corr_dim(bfast)))
    hi = min(n_dims - 1.0, np.ceil(corr_dim(bslow)))
    mscale = attn * (1.0 + 0.1 * np.log(1.0 / fscale))

    def theta(extrap, i0):
        interp = fscale * extrap
        t = (int(i0 / 2) - lo) / max(0.001, hi - lo)      # C integer division truncates toward zero
        mix = (1.0 - min(1.0, max(0.0, t))) * ext
        return interp * (1 - mix) + extrap * mix
    for i2 in range(s):
        p = float(n_past + i2)
        if mode == 0:
            for k in range(hs // 2):
                th = theta(p * ts ** k, 2 * k)
                c, sn = np.cos(th) * mscale, np.sin(th) * mscale
                x0, x1 = xs[:, i2, :, 2 * k], xs[:, i2, :, 2 * k + 1]
                y[:, i2, :, 2 * k] = x0 * c - x1 * sn
                y[:, i2, :, 2 * k + 1] = x0 * sn + x1 * c
        else:
            k = 0
            for ib in range(hs // n_dims):
                for ic in range(0, n_dims, 2):
                    cur_rot = int(-ic / n_dims - ib)      # (int) of a float: toward zero
                    th = theta(p * fscale * ts ** k, cur_rot)
                    k += 1
                    c, sn = np.cos(th) * mscale, np.sin(th) * mscale
                    i0 = ib * n_dims + ic // 2
                    x0, x1 = xs[:, i2, :, i0], xs[:, i2, :, i0 + n_dims // 2]
                    y[:, i2, :, i0] = x0 * c - x1 * sn
                    y[:, i2, :, i0 + n_dims // 2] = x0 * sn + x1 * c
    return y


YARN_CASES = [(1, 3, 4, 128, 100, 128, 0, 10000.0, 0.25, 4096, 1.0, 1.0, 32.0, 1.0),
              (2, 2, 2, 64, 5000, 64, 0, 10000.0, 0.5, 2048, 0.7, 1.2, 32.0, 1.0),
              (1, 3, 2, 128, 40, 64, 2, 10000.0, 0.25, 4096, 1.0, 1.0, 32.0, 1.0)]


@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,fscale,n_orig,ext,attn,bfast,bslow", YARN_CASES)
def test_rope_yarn_oracle_matches_closed_form(nso, b, s, h, hs, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow):
    x = np.random.default_rng(hs + n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = _yarn_closed_form(x, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow)
    out = nso.rope_f32_yarn(x, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow)
    assert np.max(np.abs(out - ref)) < 3e-3
    assert nso.rel_l2(out, ref) < 3e-4
    # the mix matters: without it the result is measurably different, and ext_factor = 0 is the plain rope
    plain = nso.rope_f32(x, n_past, n_dims, mode, base, fscale, attn)
    assert nso.rel_l2(out, plain) > 1e-2
    assert np.array_equal(nso.rope_f32_yarn(x, n_past, n_dims, mode, base, fscale, n_orig, 0.0, attn, bfast, bslow), plain)


@pytest.mark.gpu
@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,fscale,n_orig,ext,attn,bfast,bslow", YARN_CASES)
def test_rope_yarn_gpu_matches_oracle(L, pkg, nso, b, s, h, hs, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow):
    import torch
    x = np.random.default_rng(hs + n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros_like(dx)
    pkg.check(L.ns_hip_rope_f32_yarn(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, fscale, n_orig, ext,
                                     attn, bfast, bslow, None))
    torch.cuda.synchronize()
    ref = nso.rope_f32_yarn(x, n_past, n_dims, mode, base, fscale, n_orig, ext, attn, bfast, bslow)
    # theta, the ramp and the mix are the same fp32 operations; only sinf / cosf differ from the host libm
    assert np.max(np.abs(dy.cpu().numpy() - ref)) < 2e-5 * max(1.0, attn * 1.3) * np.max(np.abs(x))


# ---------------------------------------------------------------------------------------------- long-rope (mode 0x10)
def _longrope_closed_form(x, n_past, n_dims, base, fscale, attn, factors, scale):
    b, s, h, hs = x.shape
    xs = x.astype(np.float64)
    y = xs.copy()
    ts = float(np.float32(base)) ** (-2.0 / n_dims)
    for i2 in range(s):
        p = float(n_past + i2)
        k = 0
        for ib in range(hs // n_dims):
            for ic in range(0, n_dims, 2):
                th = fscale * (p * fscale * ts ** k) / float(factors[ic // 2])
                k += 1
                c, sn = np.cos(th) * attn * scale, np.sin(th) * attn * scale
                i0 = ib * n_dims + ic // 2
                x0, x1 = xs[:, i2, :, i0], xs[:, i2, :, i0 + n_dims // 2]
                y[:, i2, :, i0] = x0 * c - x1 * sn
                y[:, i2, :, i0 + n_dims // 2] = x0 * sn + x1 * c
    return y


@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims", [(1, 3, 4, 96, 50, 96), (2, 2, 2, 128, 700, 64)])
def test_rope_longrope_oracle_and_gpu(request, nso, b, s, h, hs, n_past, n_dims):
    rng = np.random.default_rng(n_dims + n_past)
    x = rng.standard_normal((b, s, h, hs)).astype(np.float32)
    factors = rng.uniform(1.0, 8.0, n_dims // 2).astype(np.float32)
    base, fscale, attn, scale = 10000.0, 0.5, 1.0, 1.19
    ref = _longrope_closed_form(x, n_past, n_dims, base, fscale, attn, factors, scale)
    out = nso.rope_f32_longrope(x, n_past, n_dims, base, fscale, 4096, 0.0, attn, 32.0, 1.0, factors, scale)
    assert np.max(np.abs(out - ref)) < 2e-3 and nso.rel_l2(out, ref) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("ext", [0.0, 1.0])
def test_rope_longrope_gpu_matches_oracle(L, pkg, nso, ext):
    import torch
    rng = np.random.default_rng(5)
    b, s, h, hs, n_past, n_dims = 1, 3, 4, 96, 50, 96
    x = rng.standard_normal((b, s, h, hs)).astype(np.float32)
    factors = rng.uniform(1.0, 8.0, n_dims // 2).astype(np.float32)
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(factors).cuda()
    dy = torch.zeros_like(dx)
    pkg.check(L.ns_hip_rope_f32_longrope(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, 10000.0, 0.5, 4096, ext, 1.0,
                                         32.0, 1.0, df.data_ptr(), 1.19, None))
    torch.cuda.synchronize()
    ref = nso.rope_f32_longrope(x, n_past, n_dims, 10000.0, 0.5, 4096, ext, 1.0, 32.0, 1.0, factors, 1.19)
    assert np.max(np.abs(dy.cpu().numpy() - ref)) < 4e-5 * np.max(np.abs(x))


# ---------------------------------------------------------------------------------------------- GLM branch (mode & 4)
def _glm_closed_form(x, n_past, n_dims, mode, base, prompt_size, pads):
    b, s, h, hs = x.shape
    y = x.astype(np.float64).copy()
    xs = x.astype(np.float64)
    ts = float(np.float32(base)) ** (-2.0 / n_dims)
    skip = bool(mode & 1)
    idx = np.arange(hs // 4)
    for i3 in range(b):
        for i2 in range(n_past if skip else 0, s):
            p = i2 if skip else n_past + i2
            tb = min(max(p - pads[i3], 0), prompt_size - 2 - pads[i3])
            bt = max(p - (prompt_size - 2), 0)
            th, bth = tb * ts ** idx, bt * ts ** idx
            x0, x1 = xs[i3, i2, :, idx], xs[i3, i2, :, idx + n_dims // 2]          # [hs/4][heads]
            x2, x3 = xs[i3, i2, :, idx + n_dims], xs[i3, i2, :, idx + n_dims // 2 * 3]
            c, sn, cb, sb = (np.cos(th)[:, None], np.sin(th)[:, None], np.cos(bth)[:, None], np.sin(bth)[:, None])
            y[i3, i2, :, idx] = x0 * c - x1 * sn
            y[i3, i2, :, idx + n_dims // 2] = x0 * sn + x1 * c
            y[i3, i2, :, idx + n_dims] = x2 * cb - x3 * sb
            y[i3, i2, :, idx + n_dims // 2 * 3] = x2 * sb + x3 * cb
    return y


GLM_CASES = [  # batch, seq, heads, head_size, n_past, n_dims, mode, base, prompt_size, n_padding
    (1, 6, 4, 128, 0, 64, 4, 10000.0, 6, [0]),           # prompt pass: every token inside the prompt
    (2, 1, 8, 128, 9, 64, 4, 10000.0, 7, [0, 3]),        # decode step past the prompt: block position counts up, padding per batch
    (1, 5, 2, 64, 2, 32, 5, 10000.0, 4, [1]),            # skip form: rows below n_past untouched
]


@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,psize,pads", GLM_CASES)
def test_rope_glm_oracle_matches_closed_form(nso, b, s, h, hs, n_past, n_dims, mode, base, psize, pads):
    x = np.random.default_rng(hs + n_past + mode).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = _glm_closed_form(x, n_past, n_dims, mode, base, psize, pads)
    out = nso.rope_f32_glm(x, n_past, n_dims, mode, base, psize, pads)
    assert np.max(np.abs(out - ref)) < 1e-4
    if mode & 1:
        assert np.array_equal(out[:, :n_past], x[:, :n_past])


@pytest.mark.gpu
@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,psize,pads", GLM_CASES)
def test_rope_glm_gpu_matches_oracle(L, pkg, nso, b, s, h, hs, n_past, n_dims, mode, base, psize, pads):
    import torch
    x = np.random.default_rng(hs * 5 + n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = nso.rope_f32_glm(x, n_past, n_dims, mode, base, psize, pads)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pad = np.ascontiguousarray(pads, np.int32)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros_like(dx)
    pkg.check(L.ns_hip_rope_f32_glm(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, psize, nso.ptr(pad), st))
    pkg.check(L.ns_hip_rope_f32_glm(dx.data_ptr(), dx.data_ptr(), b, s, h, hs, n_past, n_dims, mode, base, psize, nso.ptr(pad), st))
    torch.cuda.synchronize()
    out = dy.cpu().numpy()
    assert np.array_equal(out, dx.cpu().numpy())  # in place == out of place
    assert np.max(np.abs(out - ref)) < 1e-5 * max(1.0, float(np.abs(x).max()))
    assert L.ns_hip_rope_f32_glm(dx.data_ptr(), dy.data_ptr(), b, s, h, hs, n_past, n_dims, 2, base, psize, nso.ptr(pad), st) != 0
    L.ns_hip_reset_error()


# ------------------------------------------------------------------ the oracle pinned to the REAL reference operator
# oracle/_ref/libne_ref.so is the reference's own ne_layers.c compiled from where it lies (oracle/Makefile `neref`) and
# driven through its graph builder + ne_graph_compute by oracle/ne_ref_harness.c: ne_compute_forward_rope_f32 itself.
@pytest.fixture(scope="module")
def neref(nso):
    if nso.neref() is None:
        pytest.skip("oracle/_ref/libne_ref.so not built (reference tree absent)")
    return nso


PIN_SHAPES = [(1, 1, 32, 128, 17), (2, 5, 4, 64, 0), (1, 3, 8, 128, 2000), (1, 4, 2, 80, 9)]


@pytest.mark.parametrize("b,s,h,hs,n_past", PIN_SHAPES)
@pytest.mark.parametrize("mode,fscale", [(0, 1.0), (0, 0.25), (2, 1.0), (2, 0.5)])
def test_rope_oracle_equals_reference_operator(neref, b, s, h, hs, n_past, mode, fscale):
    x = np.random.default_rng(hs + n_past + mode).standard_normal((b, s, h, hs)).astype(np.float32)
    for n_dims in {hs, hs // 2} if mode == 2 else {hs}:
        ref = neref.neref_rope(x, n_past, n_dims, mode, 10000.0, fscale)
        out = neref.rope_f32(x, n_past, n_dims, mode, 10000.0, fscale, 1.0)
        if mode == 2 and hs % n_dims:   # dims the NeoX loop does not visit: in place, they keep x
            out = np.where(out == 0, x, out)
        assert np.array_equal(out, ref)   # same sequential fp32 products, same libm: bit for bit


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("ext,fscale", [(1.0, 0.25), (0.5, 0.5), (0.0, 0.25)])
def test_rope_yarn_oracle_equals_reference_operator(neref, mode, ext, fscale):
    b, s, h, hs, n_past, n_orig = 1, 4, 4, 128, 3000, 4096
    x = np.random.default_rng(int(ext * 10) + mode).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = neref.neref_rope(x, n_past, hs, mode, 10000.0, fscale, n_orig_ctx=n_orig, ext_factor=ext, attn_factor=1.2,
                           beta_fast=32.0, beta_slow=1.0)
    out = neref.rope_f32_yarn(x, n_past, hs, mode, 10000.0, fscale, n_orig, ext, 1.2, 32.0, 1.0)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("ext", [0.0, 1.0])
def test_rope_longrope_oracle_equals_reference_operator(neref, ext):
    b, s, h, hs, n_past = 1, 3, 4, 96, 5000
    rng = np.random.default_rng(7)
    x = rng.standard_normal((b, s, h, hs)).astype(np.float32)
    factors = (1.0 + rng.random(hs // 2) * 3).astype(np.float32)
    ref = neref.neref_rope(x, n_past, hs, 0x10, 10000.0, 0.5, n_orig_ctx=4096, ext_factor=ext, attn_factor=1.0,
                           beta_fast=32.0, beta_slow=1.0, factors=factors, scale_factor=1.19)
    out = neref.rope_f32_longrope(x, n_past, hs, 10000.0, 0.5, 4096, ext, 1.0, 32.0, 1.0, factors, 1.19)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("b,s,h,hs,n_past,n_dims,mode,base,psize,pads", GLM_CASES)
def test_rope_glm_oracle_equals_reference_operator(neref, b, s, h, hs, n_past, n_dims, mode, base, psize, pads):
    x = np.random.default_rng(hs + 11 * n_past).standard_normal((b, s, h, hs)).astype(np.float32)
    ref = neref.neref_rope(x, n_past, n_dims, mode, base, 1.0, prompt_size=psize, n_padding=pads)
    out = neref.rope_f32_glm(x, n_past, n_dims, mode, base, psize, pads)
    assert np.array_equal(out, ref)
