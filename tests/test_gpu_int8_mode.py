"""NS_COMPUTE_REF_INT8: the reference's int8-compute numerics on the GPU (csrc/ns_i8ref.hip) against the oracle's
restatement of gemv_4bit_u8s8_fp32 (kernel_ref.h:2371-2429; the oracle function is pinned to the real kernel in
tests/test_oracle_vs_ref.py).  Activation quantization is bit-exact (tests/test_gpu_parity.py covers the prologue), the
integer dots are exact, so the only difference is fp32 summation order: the bar is 2e-6 relative L2 — three orders of
magnitude tighter than the distance between the int8 path and the fp32 truth, which the last test shows."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def int8_mode(L):
    prev = L.ns_hip_set_compute_mode(1)
    assert prev in (0, 1)
    yield
    L.ns_hip_set_compute_mode(prev)


CASES = [  # qtype, scale dtype, asym, group, n, k, reference core the blob is packed for
    ("S4", "BF16", False, 32, 512, 1024, "CORE_AVX512_VNNI_KB"),
    ("S4", "F32", True, 128, 200, 1024, "CORE_AVX512_VNNI_KB"),   # asymmetric, ragged N
    ("S4", "F16", False, 64, 256, 832, "CORE_AVX512_VNNI_KB"),    # K not a multiple of the 128-deep k-step
    ("S8", "BF16", False, 32, 256, 512, "CORE_AVX512_VNNI_KB"),
    ("S8", "F32", True, 64, 96, 704, "CORE_AVX512F"),
    ("S3", "BF16", False, 32, 128, 512, "CORE_AVX512_VNNI_KB"),   # bit-plane type widened to the nibble container
    ("S4", "F32", False, -1, 128, 1024, "CORE_AVX512F"),          # per-channel
]


@pytest.mark.parametrize("qt,st,asym,bs,n,k,core", CASES)
@pytest.mark.parametrize("m", [1, 3, 9])
def test_int8_mode_matches_reference_int8_semantics(L, pkg, nso, int8_mode, qt, st, asym, bs, n, k, core, m):
    rng = np.random.default_rng(n * 7 + k * 3 + m)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    blob = nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), asym, getattr(nso, core))
    out = np.zeros((m, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    ref = nso.gemm_u8s8(a, blob)
    assert nso.rel_l2(out, ref) < 2e-6, nso.rel_l2(out, ref)


@pytest.mark.parametrize("qt,st,asym,bs,n,k,core", CASES)
@pytest.mark.parametrize("m", [16, 77, 200])
def test_int8_mode_gemm_sized_calls_run_on_the_matrix_cores(L, pkg, nso, int8_mode, qt, st, asym, bs, n, k, core, m):
    """16 rows and up take the matrix-core kernels (nibble containers: i8mfma2_kernel, one v_mfma_i32_16x16x64_i8 per 32-deep
    slice on zero-point-folded operands; byte containers: i8mfma_kernel, v_mfma_i32_16x16x32_i8 + exact integer corrections): the same
    numbers as the decode-sized kernel — ragged M (row tiles of 16 inside workgroups of 64), ragged N, K tails, asymmetric
    weights, byte containers, per-channel blocks"""
    rng = np.random.default_rng(n * 7 + k * 3 + m)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    a[m // 2] *= 30.0   # one row with a very different activation scale
    blob = nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), asym, getattr(nso, core))
    out = np.full((m, n), 7.0, np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    ref = nso.gemm_u8s8(a, blob)
    assert nso.rel_l2(out, ref) < 2e-6, nso.rel_l2(out, ref)
    # row by row as well: a wrong row tile must not hide behind the large row
    per_row = np.sqrt(((out - ref) ** 2).sum(-1) / np.maximum((ref.astype(np.float64) ** 2).sum(-1), 1e-30))
    assert per_row.max() < 1e-5, (int(per_row.argmax()), per_row.max())


@pytest.mark.parametrize("qt,st,asym,bs,n,k,core", CASES)
@pytest.mark.parametrize("m", [16, 77, 200])
def test_int8_mode_both_matrix_core_kernels_return_the_same_bits(L, pkg, nso, int8_mode, qt, st, asym, bs, n, k, core, m):
    """i8mfma2_kernel (one fp16 MFMA per slice on operands with both zero points folded in: fp16(a - za) against fp16(u - zbb),
    every product and partial sum an integer below 2^24, so the MFMA returns float(isum) exactly) and i8mfma_kernel (integer MFMA
    + integer corrections per accumulator) form the same exact integer per slice and the same fp32 expression in the same
    order: equal bit for bit, in every workgroup tile of the second kernel — nibble and byte containers alike (bytes:
    fp16(q - zb) in [-255, 255], a slice's sum below 2^21).  Rows that drive the folding to its corners included:
    all-positive / all-negative rows (zero point 0 / 255, a - za = +-255) and a constant row."""
    rng = np.random.default_rng(n * 5 + k + m)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    a[0] = np.abs(a[0]) + 0.5
    a[1] = -np.abs(a[1]) - 0.5
    a[2] = 3.0
    a[3, ::2] = 0.0
    blob = nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), asym, getattr(nso, core))
    outs = {}
    try:
        for gen, tile in ((1, 0), (2, 1), (2, 4)):  # the second kernel in each of its workgroup tiles
            assert L.ns_hip_set_tuning(b"i8_mfma", gen) == 0
            assert L.ns_hip_set_tuning(b"i8_tile", tile) == 0
            o = np.full((m, n), 7.0, np.float32)
            L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(o), m, n, k, k, n, None)
            outs[(gen, tile)] = o
    finally:
        L.ns_hip_set_tuning(b"i8_mfma", 2)
        L.ns_hip_set_tuning(b"i8_tile", 0)
    # one scale per 32-deep slice (group 32): the same expression in the same order, bit for bit.  Wider groups: the second kernel
    # chains the group's slices through the MFMA accumulator and scales the k-block's exact integer sum ONCE (round 4: the
    # reference's own order, bestla_wrapper.h:768-831) while the first kernel scales slice by slice — equal to fp32 rounding;
    # the second kernel's tiles still agree bit for bit with each other.
    one_scale_per_slice = bs == 32
    for tile in (1, 4):
        if one_scale_per_slice:
            assert np.array_equal(outs[(1, 0)].view(np.uint32), outs[(2, tile)].view(np.uint32)), (tile, np.abs(outs[(1, 0)] - outs[(2, tile)]).max())
        else:
            assert nso.rel_l2(outs[(2, tile)], outs[(1, 0)]) < 1e-6
    assert np.array_equal(outs[(2, 1)].view(np.uint32), outs[(2, 4)].view(np.uint32))
    ref = nso.gemm_u8s8(a, blob)
    assert nso.rel_l2(outs[(2, 4)], ref) < 2e-6 and nso.rel_l2(outs[(1, 0)], ref) < 2e-6


def test_int8_mode_fused_qkv_at_gemm_size_prepares_the_activations_once(L, pkg, nso, int8_mode):
    """the fused QKV entry at GEMM size: one activation quantization and one operand preparation (the fp16 A', by the quantizer itself or i8prep_kernel)
    for the three weights, as the reference quantizes A once (ip_fusion_qkv.cpp:84-86)"""
    import torch
    rng = np.random.default_rng(17)
    n, k, m, bs = 192, 640, 70, 32
    mk = lambda: nso.quant_pack((rng.standard_normal((n, k)) * 0.05).astype(np.float32), bs, nso.S4, nso.BF16, False,
                                nso.CORE_AVX512_VNNI_KB)
    blobs = [mk(), mk(), mk()]
    ws = [pkg.Weight.from_host_blob(nso.ptr(b)) for b in blobs]
    a = rng.standard_normal((m, k)).astype(np.float32)
    dA = torch.from_numpy(a).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dC = torch.zeros((3, m, n), device="cuda")
    pkg.check(L.ns_hip_fusion_qkv_forward(dA.data_ptr(), ws[0].h, ws[1].h, ws[2].h, dC.data_ptr(), m, k, n, st))
    torch.cuda.synchronize()
    for i, b in enumerate(blobs):
        assert nso.rel_l2(dC[i].cpu().numpy(), nso.gemm_u8s8(a, b)) < 2e-6


def test_int8_mode_device_entries_and_epilogues(L, pkg, nso, int8_mode):
    import torch
    rng = np.random.default_rng(5)
    n, k, m, bs = 256, 512, 2, 32
    mk = lambda: nso.quant_pack((rng.standard_normal((n, k)) * 0.05).astype(np.float32), bs, nso.S4, nso.BF16, False,
                                nso.CORE_AVX512_VNNI_KB)
    bq, bk, bv = mk(), mk(), mk()
    wq, wk, wv = (pkg.Weight.from_host_blob(nso.ptr(b)) for b in (bq, bk, bv))
    a = rng.standard_normal((m, k)).astype(np.float32)
    dA = torch.from_numpy(a).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # fused QKV entry: three int8-reference forwards
    dC = torch.zeros((3, m, n), device="cuda")
    pkg.check(L.ns_hip_fusion_qkv_forward(dA.data_ptr(), wq.h, wk.h, wv.h, dC.data_ptr(), m, k, n, st))
    torch.cuda.synchronize()
    for i, b in enumerate((bq, bk, bv)):
        assert nso.rel_l2(dC[i].cpu().numpy(), nso.gemm_u8s8(a, b)) < 2e-6
    # epilogue: C = gelu(A W + bias), bias broadcast (ldd = 0)
    bias = torch.randn((1, n), device="cuda")
    dG = torch.zeros((m, n), device="cuda")
    pkg.check(L.ns_hip_f32f32_forward(dA.data_ptr(), wq.h, dG.data_ptr(), m, k, n, pkg.EPI_ADD_GELU, bias.data_ptr(), 0, st))
    torch.cuda.synchronize()
    x = nso.gemm_u8s8(a, bq).astype(np.float64) + bias.cpu().numpy().astype(np.float64)
    gelu = 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))
    assert nso.rel_l2(dG.cpu().numpy(), gelu) < 1e-5
    # gate/up entry without a tmp1 buffer: silu(A W1) * (A W3), unfused in this mode
    dT2 = torch.zeros((m, n), device="cuda")
    pkg.check(L.ns_hip_fusion_ffn3_gateup_h(dA.data_ptr(), None, wq.h, wk.h, None, dT2.data_ptr(), None, m, pkg.EPI_SILU, st))
    torch.cuda.synchronize()
    g = nso.gemm_u8s8(a, bq).astype(np.float64)
    ref = g / (1 + np.exp(-g)) * nso.gemm_u8s8(a, bk).astype(np.float64)
    assert nso.rel_l2(dT2.cpu().numpy(), ref) < 1e-5


def test_modes_differ_as_the_reference_paths_do(L, pkg, nso):
    """fp16-activation default vs int8 mode on the same blob: the int8 path sits ~1e-2 from the fp64 truth (u8
    activations), the default within 1e-3 — and switching modes really switches kernels."""
    rng = np.random.default_rng(11)
    n, k, bs = 256, 2048, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((1, k)).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    truth = nso.gemm_f64(a, blob)
    outs = {}
    prev = L.ns_hip_get_compute_mode()
    try:
        for mode in (0, 1):
            L.ns_hip_set_compute_mode(mode)
            o = np.zeros((1, n), np.float32)
            L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(o), 1, n, k, k, n, None)
            outs[mode] = o
    finally:
        L.ns_hip_set_compute_mode(prev)
    assert nso.rel_l2(outs[0], truth) < 1e-3
    e1 = nso.rel_l2(outs[1], truth)
    assert 1e-3 < e1 < 5e-2, e1
    assert L.ns_hip_set_compute_mode(7) == -1   # bad argument
    L.ns_hip_reset_error()
