"""Seeded random sweep over shapes, row counts and formats: the forward through the reference entry point against the
oracle's fp64 GEMM on the same blob.  Exercises every dispatch boundary (M = 1 / 16 / 17 / 32 / 33 / 64 / 65 and beyond:
weight-streaming kernel variants, tiled GEMM with and without split-K), ragged N and K, all scale types."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3

FORMATS = [  # qtype, scale dtype, asym, core, group sizes that are legal for it
    ("S4", "BF16", False, "CORE_AVX512_VNNI_KB", (32, 64, 128)),
    ("S4", "F32", True, "CORE_AVX512F", (32, 128, -1)),
    ("S4", "F16", False, "CORE_AVX512F", (64, 256)),
    ("S8", "BF16", False, "CORE_AVX512F", (32, 64)),
    ("S8", "F32", True, "CORE_AVX512F", (64, 128)),
    ("F4_NF4", "BF16", False, "CORE_AVX512F", (32, 128)),
    ("F4_E2M1", "F32", False, "CORE_AVX512F", (64,)),
    ("S3", "BF16", False, "CORE_AVX512_VNNI_KB", (32,)),
    ("S6", "F32", True, "CORE_AVX512F", (32,)),
]
FORMATS_F8 = [  # fp8 weights (WeightKBlockNFloat with F8_E4M3 / F8_E5M2, quant_utils.cpp:307-341)
    ("F8_E4M3", "F8_E8M0", False, "CORE_AVX512F", (32, 64, 128)),
    ("F8_E4M3", "F32", False, "CORE_AVX512F", (32, -1)),
    ("F8_E5M2", "F8_E8M0", False, "CORE_AMX_BF16", (32, 128)),
    ("F8_E5M2", "F32", False, "CORE_AVX512F", (64,)),
]
MS = [1, 2, 4, 5, 16, 17, 31, 32, 33, 63, 64, 65, 100, 129, 200, 300]


def _cases(count, seed, formats=FORMATS):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        qt, st, asym, core, groups = formats[int(rng.integers(len(formats)))]
        bs = int(groups[int(rng.integers(len(groups)))])
        n = int(rng.integers(1, 40)) * 16 + int(rng.choice([0, 0, 1, 7, 15]))
        k = int(rng.integers(1, 17)) * 128 + int(rng.choice([0, 0, 0, 32, 64, 96]))
        if bs > 0 and k % bs:
            k = (k // bs + 1) * bs
        m = int(MS[int(rng.integers(len(MS)))])
        out.append((i, qt, st, asym, core, bs, n, k, m))
    return out


@pytest.mark.parametrize("i,qt,st,asym,core,bs,n,k,m", _cases(64, 20260925) + [(100 + c[0],) + c[1:] for c in _cases(24, 7, FORMATS_F8)])
def test_random_forward(L, pkg, nso, i, qt, st, asym, core, bs, n, k, m):
    rng = np.random.default_rng(1000 + i)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    a = rng.standard_normal((m, k + 3)).astype(np.float32)  # lda > k
    qtype = getattr(nso, qt) if hasattr(nso, qt) else nso.INT_TYPES[int(qt[1:])]
    blob = nso.quant_pack(w, bs, qtype, getattr(nso, st), asym, getattr(nso, core))
    out = np.full((m, n + 1), -3.0, np.float32)  # ldo > n
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k + 3, n + 1, None)
    assert np.all(out[:, n] == -3.0)
    ref = nso.gemm_f64(np.ascontiguousarray(a[:, :k]), blob)
    e = nso.rel_l2(np.ascontiguousarray(out[:, :n]), ref)
    assert e < TOL, (qt, st, asym, bs, n, k, m, e)


INT_FORMATS = [f for f in FORMATS if f[0].startswith("S")]


def _i8_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        qt, st, asym, core, groups = INT_FORMATS[int(rng.integers(len(INT_FORMATS)))]
        bs = int(groups[int(rng.integers(len(groups)))])
        n = int(rng.integers(1, 60)) * 16 + int(rng.choice([0, 0, 1, 7, 15]))
        k = int(rng.integers(1, 40)) * 128 + int(rng.choice([0, 0, 0, 32, 64, 96]))
        if bs > 0 and k % bs:
            k = (k // bs + 1) * bs
        out.append((i, qt, st, asym, core, bs, n, k, int(rng.integers(1, 6))))
    return out


@pytest.mark.parametrize("i,qt,st,asym,core,bs,n,k,m", _i8_cases(40, 31))
def test_random_int8_reference_numerics_decode(L, pkg, nso, i, qt, st, asym, core, bs, n, k, m):
    """NS_COMPUTE_REF_INT8 at 1..5 rows over random integer formats, group sizes (32 / 64 / 128 / 256 / per-channel), ragged N
    and K: rows <= 4 on the streaming kernel's int8 variant (or, outside its envelope, the general int8 kernel), 5 rows on the
    general kernel — the reference's u8 x s8 arithmetic (oracle: quantize_fp_u8_colblock + gemv_4bit_u8s8_fp32) to 2e-6"""
    rng = np.random.default_rng(3000 + i)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    a[0] *= 17.0  # rows with very different activation scales
    qtype = getattr(nso, qt) if hasattr(nso, qt) else nso.INT_TYPES[int(qt[1:])]
    blob = nso.quant_pack(w, bs, qtype, getattr(nso, st), asym, getattr(nso, core))
    out = np.full((m, n), -3.0, np.float32)
    prev = L.ns_hip_set_compute_mode(1)
    try:
        L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    finally:
        L.ns_hip_set_compute_mode(prev)
    ref = nso.gemm_u8s8(a, blob)
    assert nso.rel_l2(out, ref) < 2e-6, (qt, st, asym, bs, n, k, m, nso.rel_l2(out, ref))


def _qkv_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        qt, st, asym, core, groups = [f for f in FORMATS if f[0] in ("S4", "S8", "F4_NF4")][int(rng.integers(5))]
        bs = int(groups[int(rng.integers(len(groups)))])
        k = int(rng.integers(2, 12)) * 128
        if bs > 0 and k % bs:
            k = (k // bs + 1) * bs
        ns = tuple(int(rng.integers(1, 30)) * 16 + int(rng.choice([0, 0, 3, 9])) for _ in range(3))
        out.append((i, qt, st, asym, core, bs, ns, k, int(rng.choice([65, 100, 150, 191, 192, 256, 300, 513]))))
    return out


@pytest.mark.parametrize("i,qt,st,asym,core,bs,ns,k,m", _qkv_cases(16, 57))
def test_random_fused_qkv_at_gemm_size(L, pkg, nso, i, qt, st, asym, core, bs, ns, k, m):
    """bestla_fusion_QKV_f32f32_forward with 65+ rows: three weights of a random ragged width (the host entry has ONE n) in one launch of the
    tiled kernel, host pointers; each output block against the oracle's fp64 GEMM of its weight"""
    rng = np.random.default_rng(5000 + i)
    n = max(ns)
    qtype = getattr(nso, qt) if hasattr(nso, qt) else nso.INT_TYPES[int(qt[1:])]
    a = rng.standard_normal((m, k)).astype(np.float32)
    blobs = [nso.quant_pack((rng.standard_normal((n, k)) * 0.05).astype(np.float32), bs, qtype, getattr(nso, st), asym, getattr(nso, core))
             for _ in range(3)]
    if not L.bestla_fusion_QKV_f32f32_support(nso.ptr(blobs[0]), nso.ptr(blobs[1]), nso.ptr(blobs[2]), m, n, k):
        pytest.skip("format not offered by the fused entry")
    out = np.full((3, m, n), -3.0, np.float32)
    L.bestla_fusion_QKV_f32f32_forward(nso.ptr(a), nso.ptr(blobs[0]), nso.ptr(blobs[1]), nso.ptr(blobs[2]), nso.ptr(out), m, n, k, k, n, None)
    for j in range(3):
        assert nso.rel_l2(out[j], nso.gemm_f64(a, blobs[j])) < TOL, (j, qt, bs, n, k, m)
