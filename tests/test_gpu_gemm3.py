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
