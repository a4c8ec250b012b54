"""F4 blobs written on an AVX512 host carry SIGNED block scales (the reference's AVX512 quantizer keeps the sign of the
block's largest element, kernel_avx512f.h:1189-1196; tests/test_oracle_vs_avx.py): roughly half of the scales of a real
NF4 model file are negative.  The dequantizer is LUT[code] * scale either way, so every path of the library must carry
the sign through its scale conversions — decode GEMV, small-M kernel, tiled GEMM (which pre-scales by a power of two
chosen from the largest |scale|), unpack."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("qt,st", [("F4_NF4", "BF16"), ("F4_NF4", "F32"), ("F4_E2M1", "BF16"), ("F4_BNB", "F16")])
@pytest.mark.parametrize("m", [1, 8, 200])
def test_f4_blob_with_negative_scales(L, pkg, nso, qt, st, m):
    rng = np.random.default_rng(17 + m)
    n, k, bs = 256, 1024, 128 if qt == "F4_NF4" else 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), False, nso.CORE_AVX512F)
    bi = nso.parse(blob)
    # flip the sign of every second scale: the weights of those blocks change sign, nothing else
    if st == "F32":
        sc = blob[bi.scale_off:bi.scale_off + bi.scale_bytes].view(np.uint32)
        sc[::2] ^= np.uint32(0x80000000)
    else:
        sc = blob[bi.scale_off:bi.scale_off + bi.scale_bytes].view(np.uint16)
        sc[::2] ^= np.uint16(0x8000)
    wd = nso.unpack_fp32(blob)
    assert (np.sign(wd) != np.sign(nso.unpack_fp32(nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), False,
                                                                 nso.CORE_AVX512F)))).mean() > 0.3
    a = rng.standard_normal((m, k)).astype(np.float32)
    out = np.zeros((m, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    e = nso.rel_l2(out, nso.gemm_f64(a, blob))
    assert e < 1e-3, e
    # and the library's own unpack (BTLAGemmUnPackB)
    back = np.zeros((k, n), np.float32)
    assert L.ns_BTLAGemmUnPackB(nso.ptr(back), nso.ptr(blob), n, k, n, None)
    assert np.array_equal(back, wd)
