"""Full-size parity of the kernels the bench times, directly against the oracle (VERDICT r02 "parity evidence holes"):

* gemm3_kernel / i8mfma_kernel at BASELINE config 3's REAL size — M = 2048 through the three Llama-2-7B shapes, int8 and
  int4 weights, both row-tile heights — on a random sample of rows (every column of them) against the oracle's fp64
  GEMM (`nso.gemm_f64`; the reference's bar is ut/bestla_prologue_b.cpp:548-785) and, in the int8-reference mode,
  against the oracle's u8 x s8 GEMM (`nso.gemm_u8s8`, 2e-6).
* gemv_kernel itself — device entry with the fp16 shadow, the path bench.py's chain takes — on all five 7B shapes x
  {int4 g32, int8 g32, NF4 g128, int4 asym f32-scale g128, fp8} at 1 / 8 / 16 rows, every output column.
* config 4 (Mistral-7B NF4 g128, 8 rows) at its full FFN width 14336.

Weights are quantized and packed by the GPU quantizer (bit-exact with the oracle, tests/test_gpu_parity.py); the oracle
reads the same blob."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star


def _device_blob(L, pkg, nso, n, k, qt, st_dt, bs, comp, asym, seed, wscale=0.02):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    dW = torch.randn((n, k), generator=g, device="cuda") * wscale
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, asym, comp, None)
    assert size > 0, pkg.last_error()
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, bs, qt, st_dt, asym, comp, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    wt = pkg.Weight.from_device_blob(dBlob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return blob, wt


def _forward_h(L, pkg, wt, dA, m, k, n, shadow=True):
    """ns_hip_f32f32_forward_h: fp32 A + its fp16 shadow in, fp32 C out (what bench.py's chains call)"""
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dA16 = dA.to(torch.float16) if shadow else None
    dC = torch.full((m, n), 7.0, dtype=torch.float32, device="cuda")
    pkg.check(L.ns_hip_f32f32_forward_h(dA.data_ptr(), dA16.data_ptr() if shadow else None, wt.h, dC.data_ptr(), None, m, k, n, 0,
                                        None, 0, st))
    torch.cuda.synchronize()
    return dC.cpu().numpy()


PREFILL_SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008)]  # (n, k)


@pytest.mark.parametrize("bm", [128, 256, 257])  # 257: 256-row tiles with tall (256 x 32) wave tiles, int4 only
@pytest.mark.parametrize("fmt", ["s8_g32_bf16", "s4_g32_bf16"])
@pytest.mark.parametrize("n,k", PREFILL_SHAPES)
def test_config3_prefill_m2048_full_size(L, pkg, nso, n, k, fmt, bm):
    """gemm3_kernel at M = 2048 on the real 7B shapes: 24 sampled rows (spread over every 256-row block, first and last
    row included), all N columns, against the fp64 GEMM and against the fp64 GEMM fed fp16-rounded activations"""
    import torch
    qt = pkg.S8 if fmt.startswith("s8") else pkg.S4
    comp = pkg.COMP_F32 if fmt.startswith("s8") else pkg.COMP_INT8
    blob, wt = _device_blob(L, pkg, nso, n, k, qt, pkg.BF16, 32, comp, False, seed=n * 3 + k)
    m = 2048
    g = torch.Generator(device="cuda").manual_seed(11)
    dA = torch.randn((m, k), generator=g, device="cuda")
    assert L.ns_hip_set_tuning(b"g3_bm", bm) == 0
    try:
        out = _forward_h(L, pkg, wt, dA, m, k, n)
    finally:
        L.ns_hip_set_tuning(b"g3_bm", 0)
    rng = np.random.default_rng(n + k + bm)
    rows = np.unique(np.concatenate([[0, m - 1, 127, 128, 255, 256], rng.integers(0, m, 18)]))
    a = dA[torch.from_numpy(rows).cuda()].cpu().numpy()
    ref, ref16 = nso.gemm_f64_pair(a, blob)
    e = nso.rel_l2(out[rows], ref)
    assert e < TOL, (fmt, bm, n, k, e)
    e16 = nso.rel_l2(out[rows], ref16)
    assert e16 < 5e-4, (fmt, bm, n, k, e16)
    per_row = np.sqrt(((out[rows] - ref) ** 2).sum(-1) / (ref ** 2).sum(-1))
    assert per_row.max() < TOL, (int(rows[per_row.argmax()]), per_row.max())
    # rows not sampled: finite, and no row left at the fill value
    assert np.isfinite(out).all() and not (out == 7.0).all(axis=1).any()
    wt.free()


@pytest.mark.parametrize("fmt", ["s8_g32_bf16", "s4_g32_bf16"])
@pytest.mark.parametrize("n,k", PREFILL_SHAPES)
def test_config3_prefill_m2048_int8_reference_mode(L, pkg, nso, n, k, fmt):
    """i8mfma_kernel (NS_COMPUTE_REF_INT8) at M = 2048 on the real shapes: the reference's u8 x s8 arithmetic
    (bestla_wrapper.h:768-831, ut/bestla_gemm.cpp:159-190) — sampled rows against the oracle's restatement, 2e-6"""
    import torch
    qt = pkg.S8 if fmt.startswith("s8") else pkg.S4
    blob, wt = _device_blob(L, pkg, nso, n, k, qt, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n * 5 + k)
    m = 2048
    g = torch.Generator(device="cuda").manual_seed(13)
    dA = torch.randn((m, k), generator=g, device="cuda")
    prev = L.ns_hip_set_compute_mode(1)
    try:
        out = _forward_h(L, pkg, wt, dA, m, k, n, shadow=False)
    finally:
        L.ns_hip_set_compute_mode(prev)
    rng = np.random.default_rng(n + k)
    rows = np.unique(np.concatenate([[0, m - 1, 63, 64], rng.integers(0, m, 12)]))
    a = dA[torch.from_numpy(rows).cuda()].cpu().numpy()
    ref = nso.gemm_u8s8(a, blob)
    e = nso.rel_l2(out[rows], ref)
    assert e < 2e-6, (fmt, n, k, e)
    per_row = np.sqrt(((out[rows] - ref) ** 2).sum(-1) / np.maximum((ref.astype(np.float64) ** 2).sum(-1), 1e-30))
    assert per_row.max() < 1e-5, (int(rows[per_row.argmax()]), per_row.max())
    assert np.isfinite(out).all() and not (out == 7.0).all(axis=1).any()
    wt.free()


# format table of the decode kernel: (name, qtype attr, scale attr, group, comp attr, asym)
GEMV_FORMATS = [
    ("int4_g32_bf16", "S4", "BF16", 32, "COMP_INT8", False),      # the headline format (INT4, SPS 4)
    ("int8_g32_bf16", "S8", "BF16", 32, "COMP_F32", False),       # INT8, SPS 2
    ("nf4_g128_bf16", "F4_NF4", "BF16", 128, "COMP_BF16", False),  # F4, SPS 1
    ("int4_asym_g128_f32", "S4", "F32", 128, "COMP_F32", True),   # ASYM, fp32 scales, SPS 1
    ("fp8_e4m3_g32_f32", "F8_E4M3", "F32", 32, "COMP_F32", False),  # F8
]
GEMV_SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096), (12288, 4096)]  # wo, w1, w2, lm_head, qkv-wide
# the headline format on every shape at 1 / 8 / 16 rows; the other instantiations on the three layer shapes at 1 / 16
GEMV_CASES = [(n, k, GEMV_FORMATS[0], m) for (n, k) in GEMV_SHAPES for m in (1, 8, 16)] + \
             [(n, k, f, m) for f in GEMV_FORMATS[1:] for (n, k) in GEMV_SHAPES[:3] for m in (1, 16)]


@pytest.mark.parametrize("n,k,fmt,m", GEMV_CASES, ids=["%s-%dx%d-m%d" % (c[2][0], c[0], c[1], c[3]) for c in GEMV_CASES])
def test_gemv_kernel_full_size_every_format(L, pkg, nso, n, k, fmt, m):
    """gemv_kernel (fp16 shadow in) on the 7B shapes: every output column against the oracle's fp64 GEMM on the same blob.
    The K = 11008 case at 8 / 16 rows exceeds the kernel's 64 KB activation envelope and is served by smallm_kernel — the
    dispatch the library really takes for that call is what is checked."""
    import torch
    name, qt, st_dt, bs, comp, asym = fmt
    if not hasattr(pkg, qt):
        pytest.skip("format constant %s not exported by the package" % qt)
    blob, wt = _device_blob(L, pkg, nso, n, k, getattr(pkg, qt), getattr(pkg, st_dt), bs, getattr(pkg, comp), asym,
                            seed=n * 7 + k + bs)
    g = torch.Generator(device="cuda").manual_seed(m * 100 + 7)
    dA = torch.randn((m, k), generator=g, device="cuda")
    out = _forward_h(L, pkg, wt, dA, m, k, n)
    a = dA.cpu().numpy()
    ref, ref16 = nso.gemm_f64_pair(a, blob)
    e = nso.rel_l2(out, ref)
    assert e < TOL, (name, n, k, m, e)
    e16 = nso.rel_l2(out, ref16)
    assert e16 < (6e-4 if qt.startswith("F4") else 3e-5), (name, n, k, m, e16)
    # no single output off by more than 1e-3 of the output scale: a bad 16-column tile cannot hide in the norm
    rms = np.sqrt(np.mean(ref ** 2))
    assert np.max(np.abs(out - ref)) < 8e-3 * rms, (name, n, k, m, float(np.max(np.abs(out - ref)) / rms))
    wt.free()


@pytest.mark.parametrize("n,k", [(14336, 4096), (4096, 14336), (1024, 4096), (32000, 4096)])
def test_config4_mistral7b_nf4_g128_batch8_full_width(L, pkg, nso, n, k):
    """config 4 at its real FFN width (14336) and vocabulary: NF4 RTN g128 bf16 scales, 8 rows, every column"""
    import torch
    blob, wt = _device_blob(L, pkg, nso, n, k, pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16, False, seed=n + k * 3)
    g = torch.Generator(device="cuda").manual_seed(8)
    dA = torch.randn((8, k), generator=g, device="cuda")
    out = _forward_h(L, pkg, wt, dA, 8, k, n)
    a = dA.cpu().numpy()
    ref = nso.gemm_f64(a, blob)
    e = nso.rel_l2(out, ref)
    assert e < TOL, (n, k, e)
    # the host entry (fp32 activations only) on the same blob
    out2 = np.zeros((8, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out2), 8, n, k, k, n, None)
    assert nso.rel_l2(out2, ref) < TOL
    L.ns_hip_cache_clear()
    wt.free()


A32_CASES = [(n, k, GEMV_FORMATS[0], m) for (n, k) in GEMV_SHAPES[:4] for m in (1, 2, 4)] + \
            [(n, k, f, 1) for f in GEMV_FORMATS[1:] for (n, k) in GEMV_SHAPES[:3]] + \
            [(4096, 4096 + 96, GEMV_FORMATS[0], 1), (1008, 4096, GEMV_FORMATS[0], 3)]  # ragged last k-step / last tile


@pytest.mark.parametrize("n,k,fmt,m", A32_CASES, ids=["%s-%dx%d-m%d" % (c[2][0], c[0], c[1], c[3]) for c in A32_CASES])
def test_gemv_kernel_fp32_activations_only_equals_the_shadow_path(L, pkg, nso, n, k, fmt, m):
    """A caller without an fp16 shadow (the reference's device graph: bestla_device_f32f32_forward, ne_bestla.h:110-112)
    is served by the same streaming kernel, converting its share of A on the way to LDS: BIT FOR BIT what the caller with
    the round-to-nearest shadow gets, and the oracle's fp64 GEMM within the usual tolerance."""
    import torch
    name, qt, st_dt, bs, comp, asym = fmt
    if not hasattr(pkg, qt):
        pytest.skip("format constant %s not exported by the package" % qt)
    blob, wt = _device_blob(L, pkg, nso, n, k, getattr(pkg, qt), getattr(pkg, st_dt), bs, getattr(pkg, comp), asym,
                            seed=n * 5 + k + bs + m)
    g = torch.Generator(device="cuda").manual_seed(m * 10 + 3)
    dA = torch.randn((m, k), generator=g, device="cuda")
    with_shadow = _forward_h(L, pkg, wt, dA, m, k, n, shadow=True)
    fp32_only = _forward_h(L, pkg, wt, dA, m, k, n, shadow=False)
    assert np.isfinite(fp32_only).all()
    assert np.array_equal(with_shadow.view(np.int32), fp32_only.view(np.int32)), \
        (name, n, k, m, float(np.max(np.abs(with_shadow - fp32_only))))
    ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
    assert nso.rel_l2(fp32_only, ref) < TOL, (name, n, k, m)
    assert nso.rel_l2(fp32_only, ref16) < (6e-4 if qt.startswith("F4") else 3e-5), (name, n, k, m)
    wt.free()


I8_SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]


@pytest.mark.parametrize("m", [1, 4])
@pytest.mark.parametrize("n,k", I8_SHAPES)
def test_int8_reference_numerics_decode_full_size(L, pkg, nso, n, k, m):
    """NS_COMPUTE_REF_INT8 at decode size on the 7B shapes: the streaming kernel's int8 variant (gemv_kernel XV = 3) against the
    oracle's restatement of quantize_fp_u8_colblock + gemv_4bit_u8s8_fp32 on the same blob — exact integer dots, so only the
    fp32 summation order differs (2e-6); the plain entry and, for the square shape, the fused QKV and gate/up entries"""
    import torch
    blob, wt = _device_blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n + 3 * k + m)
    g = torch.Generator(device="cuda").manual_seed(m * 7 + 1)
    dA = torch.randn((m, k), generator=g, device="cuda")
    prev = L.ns_hip_set_compute_mode(1)
    try:
        out = _forward_h(L, pkg, wt, dA, m, k, n, shadow=False)
        ref = nso.gemm_u8s8(dA.cpu().numpy(), blob)
        assert nso.rel_l2(out, ref) < 2e-6, (n, k, m, nso.rel_l2(out, ref))
        if n == k:  # fused entries: three / two weights, one activation quantization, one launch
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            blob2, wt2 = _device_blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n + 5 * k + m)
            qkv = torch.full((3, m, n), 7.0, dtype=torch.float32, device="cuda")
            pkg.check(L.ns_hip_fusion_qkv_forward_h(dA.data_ptr(), None, wt.h, wt2.h, wt.h, qkv.data_ptr(), None, m, k, n, st))
            t1 = torch.empty((m, n), dtype=torch.float32, device="cuda")
            t2 = torch.full((m, n), 7.0, dtype=torch.float32, device="cuda")
            pkg.check(L.ns_hip_fusion_ffn3_gateup_h(dA.data_ptr(), None, wt.h, wt2.h, t1.data_ptr(), t2.data_ptr(), None, m, pkg.EPI_SILU, st))
            torch.cuda.synchronize()
            ref2 = nso.gemm_u8s8(dA.cpu().numpy(), blob2)
            q = qkv.cpu().numpy()
            assert nso.rel_l2(q[0], ref) < 2e-6 and nso.rel_l2(q[1], ref2) < 2e-6 and nso.rel_l2(q[2], ref) < 2e-6
            silu = ref.astype(np.float64) / (1.0 + np.exp(-ref.astype(np.float64)))
            assert nso.rel_l2(t2.cpu().numpy(), (silu * ref2).astype(np.float32)) < 5e-6
            wt2.free()
    finally:
        L.ns_hip_set_compute_mode(prev)
    wt.free()


@pytest.mark.parametrize("bm", [128, 256])
@pytest.mark.parametrize("ns,k,m", [((4096, 4096, 4096), 4096, 2048), ((4096, 1024, 1024), 4096, 2048), ((1000, 200, 136), 1024, 300)],
                         ids=["mha-4096", "gqa-4096-1024", "ragged-n"])
def test_fused_qkv_at_gemm_size_is_one_launch_of_the_tiled_kernel(L, pkg, nso, ns, k, m, bm):
    """bestla_fusion_QKV_f32f32_forward's GEMM-sized form (ip_fusion_qkv.cpp:84-86): the three weights side by side along the
    column blocks of ONE gemm3_kernel launch — equal, GQA-sized and ragged widths (matrices that end inside a column block);
    every output of sampled rows against the oracle's fp64 GEMM per weight, and the three separate launches to fp32 rounding"""
    import torch
    blobs, wts = [], []
    for i, n in enumerate(ns):
        b, w = _device_blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n * 3 + k + i)
        blobs.append(b), wts.append(w)
    g = torch.Generator(device="cuda").manual_seed(13)
    dA = torch.randn((m, k), generator=g, device="cuda")
    dA16 = dA.half()
    ldc = max(ns)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fused = torch.full((3, m, ldc), 7.0, dtype=torch.float32, device="cuda")
    fused16 = torch.zeros((3, m, ldc), dtype=torch.float16, device="cuda")
    assert L.ns_hip_set_tuning(b"g3_bm", bm) == 0
    try:
        pkg.check(L.ns_hip_fusion_qkv_forward_h(dA.data_ptr(), dA16.data_ptr(), wts[0].h, wts[1].h, wts[2].h, fused.data_ptr(),
                                                fused16.data_ptr(), m, k, ldc, st))
        sep = torch.full((3, m, ldc), 7.0, dtype=torch.float32, device="cuda")
        for i, n in enumerate(ns):
            pkg.check(L.ns_hip_f32f32_forward_h(dA.data_ptr(), dA16.data_ptr(), wts[i].h, sep[i].data_ptr(), None, m, k, ldc, 0, None, 0, st))
        torch.cuda.synchronize()
    finally:
        L.ns_hip_set_tuning(b"g3_bm", 0)
    rng = np.random.default_rng(k + m + bm)
    rows = np.unique(np.concatenate([[0, m - 1, min(255, m - 1), min(256, m - 1)], rng.integers(0, m, 10)]))
    a = dA[torch.from_numpy(rows).cuda()].cpu().numpy()
    f, s16, sp = fused.cpu().numpy(), fused16.cpu().numpy(), sep.cpu().numpy()
    for i, n in enumerate(ns):
        ref = nso.gemm_f64(a, blobs[i])
        assert nso.rel_l2(f[i][rows][:, :n], ref) < TOL, (i, n)
        # the separate launches K-split a matrix with few tiles (another fp32 summation order), the fused launch never does
        assert nso.rel_l2(f[i][:, :n], sp[i][:, :n]) < 1e-6, (i, n)
        assert np.array_equal(s16[i][:, :n], f[i][:, :n].astype(np.float16)), (i, n)
        assert (f[i][:, n:] == 7.0).all(), (i, n)  # nothing written beyond a matrix's own columns
    for w in wts:
        w.free()
