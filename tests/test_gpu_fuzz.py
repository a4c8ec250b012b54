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
