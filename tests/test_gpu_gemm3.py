"""gemm3_kernel (prefill GEMM, M >= 192) on both row-tile heights: ragged N / K / M, asymmetric and fp32-scale formats,
8-bit and 4-bit-float weights, every epilogue with and without aligned outputs, the fp16 shadow, the split-K path —
against the oracle's fp64 GEMM on the same blob."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3

CASES = [  # qtype, scale, asym, core, group, n, k, m
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 256, 512, 256),
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 263, 448, 193),    # ragged N, K = 3.5 x 128 (odd chunk count), ragged M
    ("S4", "F32", True, "CORE_AVX512F", 128, 400, 1024, 300),
    ("S4", "F16", False, "CORE_AVX512F", 64, 129, 2048, 512),          # few tiles: split-K
    ("S8", "BF16", False, "CORE_AVX512F", 32, 384, 576, 257),          # K = 9 x 64
    ("S8", "F32", True, "CORE_AVX512F", 64, 144, 1024, 640),
    ("F4_NF4", "BF16", False, "CORE_AVX512F", 128, 272, 768, 200),
    ("F4_E2M1", "F32", False, "CORE_AVX512F", 64, 208, 640, 320),
    ("S3", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 160, 384, 192),
]


@pytest.mark.parametrize("bm", [128, 256])
@pytest.mark.parametrize("qt,st,asym,core,bs,n,k,m", CASES)
def test_gemm3_host_api(L, pkg, nso, bm, qt, st, asym, core, bs, n, k, m):
    rng = np.random.default_rng(n * 7 + k + m + bm)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    qtype = getattr(nso, qt) if hasattr(nso, qt) else nso.INT_TYPES[int(qt[1:])]
    blob = nso.quant_pack(w, bs, qtype, getattr(nso, st), asym, getattr(nso, core))
    ref = nso.gemm_f64(a, blob)
    assert L.ns_hip_set_tuning(b"g3_bm", bm) == 0
    try:
        out = np.zeros((m, n), np.float32)
        L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
        assert nso.rel_l2(out, ref) < TOL, (qt, bm, nso.rel_l2(out, ref))
    finally:
        L.ns_hip_set_tuning(b"g3_bm", 0)
        L.ns_hip_cache_clear()


@pytest.mark.parametrize("bm", [128, 256])
@pytest.mark.parametrize("epi", ["none", "add", "mul", "add_gelu", "gelu", "silu"])
@pytest.mark.parametrize("aligned", [True, False])
def test_gemm3_epilogues_and_shadow(L, pkg, nso, bm, epi, aligned):
    """device API with the fp16 shadow in and out; `aligned` = leading dimensions that allow the float4 row stores"""
    import torch
    n, k, m = 320, 512, 260
    ldc = n if aligned else n + 1
    rng = np.random.default_rng(len(epi) + bm)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wt = pkg.Weight.from_host_blob(nso.ptr(blob), st)
    a = rng.standard_normal((m, k)).astype(np.float32)
    d = rng.standard_normal((m, ldc)).astype(np.float32)
    da, dd = torch.from_numpy(a).cuda(), torch.from_numpy(d).cuda()
    da16 = da.half()
    dc = torch.full((m, ldc), -7.0, device="cuda")
    dc16 = torch.zeros((m, ldc), device="cuda", dtype=torch.float16)
    code = {"none": pkg.EPI_NONE, "add": pkg.EPI_ADD, "mul": pkg.EPI_MUL, "add_gelu": pkg.EPI_ADD_GELU, "gelu": pkg.EPI_GELU,
            "silu": pkg.EPI_SILU}[epi]
    assert L.ns_hip_set_tuning(b"g3_bm", bm) == 0
    try:
        pkg.check(L.ns_hip_f32f32_forward_h(da.data_ptr(), da16.data_ptr(), wt.h, dc.data_ptr(), dc16.data_ptr(), m, k, ldc, code,
                                            dd.data_ptr() if epi in ("add", "mul", "add_gelu") else None, ldc, st))
        torch.cuda.synchronize()
    finally:
        L.ns_hip_set_tuning(b"g3_bm", 0)
    g = nso.gemm_f64(a, blob)
    gelu = lambda x: 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))
    dv = d[:, :n].astype(np.float64)
    ref = {"none": g, "add": g + dv, "mul": g * dv, "add_gelu": gelu(g + dv), "gelu": gelu(g), "silu": g / (1 + np.exp(-g))}[epi]
    out = dc.cpu().numpy()
    assert nso.rel_l2(out[:, :n], ref) < 2e-3 if epi == "mul" else nso.rel_l2(out[:, :n], ref) < TOL
    if not aligned:
        assert np.all(out[:, n] == -7.0)  # nothing written past N
    assert np.allclose(dc16.float().cpu().numpy()[:, :n], out[:, :n], rtol=2e-3, atol=2e-3)


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the cross-wave output epilogue, fp16-only outputs, and gate / up tile pairs (the FFN at GEMM size)
# ---------------------------------------------------------------------------------------------------------------------
def _silu(x):
    return x / (1.0 + np.exp(-x))


def _gelu(x):
    return 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))


@pytest.mark.parametrize("bm", [64, 128, 257])
@pytest.mark.parametrize("epi", ["none", "add", "silu"])
def test_cross_wave_epilogue_writes_the_bits_of_the_per_wave_one(L, pkg, nso, bm, epi):
    """ns_hip_set_tuning("g3_wide", 0 / 1): the same accumulators leave through two epilogues — every fp32 and fp16 output bit equal;
    full column blocks (float4 / 16-byte fp16 stores) and a ragged last block (element stores) in one launch"""
    import torch
    n, k, m = 128 * 3 + 48, 512, 64 if bm == 64 else 300
    rng = np.random.default_rng(bm + len(epi))
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wt = pkg.Weight.from_host_blob(nso.ptr(blob), st)
    ldc = n + 8 - n % 8
    da = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).cuda()
    da16 = da.half()
    dd = torch.from_numpy(rng.standard_normal((m, ldc)).astype(np.float32)).cuda()
    code = {"none": pkg.EPI_NONE, "add": pkg.EPI_ADD, "silu": pkg.EPI_SILU}[epi]
    outs = []
    assert L.ns_hip_set_tuning(b"g3_bm", bm) == 0
    try:
        for wide in (0, 1):
            assert L.ns_hip_set_tuning(b"g3_wide", wide) == 0
            dc = torch.full((m, ldc), -7.0, device="cuda")
            dc16 = torch.full((m, ldc), -7.0, device="cuda", dtype=torch.float16)
            pkg.check(L.ns_hip_f32f32_forward_h(da.data_ptr(), da16.data_ptr(), wt.h, dc.data_ptr(), dc16.data_ptr(), m, k, ldc, code,
                                                dd.data_ptr() if epi == "add" else None, ldc, st))
            torch.cuda.synchronize()
            outs.append((dc.cpu().numpy(), dc16.cpu().numpy()))
    finally:
        L.ns_hip_set_tuning(b"g3_wide", -1)
        L.ns_hip_set_tuning(b"g3_bm", 0)
        wt.free()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.all(outs[1][0][:, n:] == -7.0) and np.all(outs[1][1][:, n:] == -7.0)  # nothing written past N
    g = nso.gemm_f64(da.cpu().numpy(), blob)
    ref = {"none": g, "add": g + dd.cpu().numpy()[:, :n].astype(np.float64), "silu": _silu(g)}[epi]
    assert nso.rel_l2(outs[1][0][:, :n], ref) < TOL


DUAL_CASES = [  # qtype, scale, asym, core, group, n (ffn width), k, m, activation
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 448, 512, 300, "silu"),     # 7 column blocks of 64
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 424, 448, 193, "silu"),     # ragged N (last block 40 wide), odd chunk count, ragged M
    ("S4", "F32", True, "CORE_AVX512F", 128, 320, 1024, 130, "gelu"),            # zero points + fp32 scales, Gelu_Mul
    ("S8", "BF16", False, "CORE_AVX512F", 32, 384, 576, 257, "silu"),
    ("F4_NF4", "BF16", False, "CORE_AVX512F", 128, 256, 768, 200, "silu"),
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 4096, 256, 40, "silu"),      # 64-row tile (17 .. 64 rows on a wide output)
]


@pytest.mark.parametrize("qt,sdt,asym,core,bs,n,k,m,actn", DUAL_CASES)
def test_ffn_gate_up_at_gemm_size_is_one_launch_on_tile_pairs(L, pkg, nso, qt, sdt, asym, core, bs, n, k, m, actn):
    """bestla_fusion_FFN_{SiLu,Gelu_Mul}_f32f32_forward's first half at GEMM size (ip_fusion_ffn.cpp:364-406): gemm3_kernel on gate / up
    tile pairs, act(A W1) * (A W3) formed in registers.  tmp1 (fp32), tmp2 (fp32) and tmp2's fp16 shadow are each written only when a
    pointer is handed over; all three against the oracle's fp64 GEMMs on the same blobs, and the fp16-only call against the full one"""
    import torch
    rng = np.random.default_rng(n + k + m)
    qtype = getattr(nso, qt) if hasattr(nso, qt) else nso.INT_TYPES[int(qt[1:])]
    blobs = [nso.quant_pack((rng.standard_normal((n, k)) * 0.05).astype(np.float32), bs, qtype, getattr(nso, sdt), asym, getattr(nso, core))
             for _ in range(2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w1, w3 = (pkg.Weight.from_host_blob(nso.ptr(b), st) for b in blobs)
    a = rng.standard_normal((m, k)).astype(np.float32)
    da = torch.from_numpy(a).cuda()
    da16 = da.half()
    act = {"silu": pkg.EPI_SILU, "gelu": pkg.EPI_GELU}[actn]
    t1 = torch.full((m, n), -7.0, device="cuda")
    t2 = torch.full((m, n), -7.0, device="cuda")
    t216 = torch.full((m, n), -7.0, device="cuda", dtype=torch.float16)
    only16 = torch.full((m, n), -7.0, device="cuda", dtype=torch.float16)
    try:
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(da.data_ptr(), da16.data_ptr(), w1.h, w3.h, t1.data_ptr(), t2.data_ptr(), t216.data_ptr(), m, act, st))
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(da.data_ptr(), da16.data_ptr(), w1.h, w3.h, None, None, only16.data_ptr(), m, act, st))
        torch.cuda.synchronize()
    finally:
        w1.free(), w3.free()
    g, u = nso.gemm_f64(a, blobs[0]), nso.gemm_f64(a, blobs[1])
    ag = {"silu": _silu, "gelu": _gelu}[actn](g)
    assert nso.rel_l2(t1.cpu().numpy(), ag) < TOL
    assert nso.rel_l2(t2.cpu().numpy(), ag * u) < 2e-3  # a product of two rounded factors, like the Mul epilogue above
    assert torch.equal(t216, t2.half()) and torch.equal(only16, t216)


@pytest.mark.parametrize("qt,n,k,m", [("S4", 704, 512, 300), ("S8", 384, 512, 150)])
def test_ffn3_at_gemm_size_keeps_its_intermediate_in_fp16_only(L, pkg, nso, qt, n, k, m):
    """ns_hip_fusion_ffn3_forward_h with no temporaries handed over (the reference's graph treats tmp1 / tmp2 as scratch): gate / up
    pairs -> fp16 intermediate in per-stream scratch -> down projection on it; against the fp64 chain of the oracle and against
    the same call with every temporary requested (same kernels, same bits)"""
    import torch
    rng = np.random.default_rng(n + m)
    mk = lambda nn, kk: nso.quant_pack((rng.standard_normal((nn, kk)) * 0.05).astype(np.float32), 32, getattr(nso, qt), nso.BF16, False,
                                       nso.CORE_AVX512_VNNI_KB if qt == "S4" else nso.CORE_AVX512F)
    b1, b3, b2 = mk(n, k), mk(n, k), mk(k, n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w1, w3, w2 = (pkg.Weight.from_host_blob(nso.ptr(b), st) for b in (b1, b3, b2))
    a = rng.standard_normal((m, k)).astype(np.float32)
    da = torch.from_numpy(a).cuda()
    da16 = da.half()
    out = torch.full((m, k), -7.0, device="cuda")
    out16 = torch.full((m, k), -7.0, device="cuda", dtype=torch.float16)
    out_b = torch.full((m, k), -7.0, device="cuda")
    t1, t2 = torch.empty((m, n), device="cuda"), torch.empty((m, n), device="cuda")
    t216 = torch.empty((m, n), device="cuda", dtype=torch.float16)
    try:
        pkg.check(L.ns_hip_fusion_ffn3_forward_h(da.data_ptr(), da16.data_ptr(), w1.h, w2.h, w3.h, None, None, None, out.data_ptr(),
                                                 out16.data_ptr(), m, pkg.EPI_SILU, st))
        pkg.check(L.ns_hip_fusion_ffn3_forward_h(da.data_ptr(), da16.data_ptr(), w1.h, w2.h, w3.h, t1.data_ptr(), t2.data_ptr(), t216.data_ptr(),
                                                 out_b.data_ptr(), None, m, pkg.EPI_SILU, st))
        torch.cuda.synchronize()
    finally:
        w1.free(), w2.free(), w3.free()
    mid = (_silu(nso.gemm_f64(a, b1)) * nso.gemm_f64(a, b3)).astype(np.float32)
    ref = nso.gemm_f64(mid, b2)
    assert nso.rel_l2(out.cpu().numpy(), ref) < 2e-3
    assert torch.equal(out, out_b) and torch.equal(out16, out.half())
