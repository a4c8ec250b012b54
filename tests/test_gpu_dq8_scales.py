"""DQ8_BNB double-quantised scales (bestla_storage.h:750-759, kernel_ref.h:1930-1992; round 4): blobs written by the oracle's packer — itself
byte-equal to the reference's real packer on these formats, tests/test_oracle_vs_packer.py — load into the library (the u8 scale codes are
expanded to the fp32 scales the reference dequantises with), unpack bit for bit and run the forward within north_star's 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.mark.parametrize("qt,bs,core,n,k", [
    ("S4", 32, "CORE_AVX512F", 272, 1024),          # every dq block full
    ("S4", 128, "CORE_AVX512_VNNI_KB", 100, 768),   # ragged N: the scale count is not a multiple of the dq block (the reference's tail indexing)
    ("F4_NF4", 32, "CORE_AVX512F", 144, 1024),
    ("F4_NF4", 128, "CORE_AMX_BF16", 100, 512),
])
@pytest.mark.parametrize("m", [1, 4, 70])
def test_dq8_scaled_blobs_load_unpack_and_forward(L, pkg, nso, qt, bs, core, n, k, m):
    rng = np.random.default_rng(n + k + m)
    w = (rng.standard_normal((n, k)) * 0.03).astype(np.float32)
    w[::7] *= 6.0  # a spread of scales, so that the code map is exercised beyond its middle
    blob = nso.quant_pack(w, bs, getattr(nso, qt), nso.DQ8_BNB, False, getattr(nso, core))
    bi = nso.parse(blob)
    assert bi.dq_bytes > 0 and bi.dq_blocksize == bs
    deq = np.zeros((k, n), np.float32)
    L.bestla_unpackweight_fp32(nso.ptr(blob), n, k, nso.ptr(deq), n)
    assert np.array_equal(deq.view(np.uint32), nso.unpack_fp32(blob).view(np.uint32))
    a = rng.standard_normal((m, k)).astype(np.float32)
    out = np.full((m, n), 7.0, np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    assert nso.rel_l2(out, nso.gemm_f64(a, blob)) < TOL


def test_dq8_scales_on_other_weight_types_are_refused(L, pkg, nso):
    """the reference writes such blobs but cannot read them back (bestla_prologue_b.h:742-751): refused loudly, not guessed at"""
    # an S4 blob relabelled S8 would not parse; build the header the library sees instead: an S4 DQ8 blob whose dtype word says S3
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((64, 256)) * 0.03).astype(np.float32)
    blob = nso.quant_pack(w, 32, nso.S4, nso.DQ8_BNB, False, nso.CORE_AVX512F)
    bad = blob.copy()
    bad[36:40] = np.frombuffer(np.uint32(nso.S3).tobytes(), np.uint8)  # u64 size, u32 prologue, u64 core, 4 x i32, then the dtype word
    assert nso.parse(bad).dtype == nso.S3
    out = np.zeros((1, 64), np.float32)
    a = np.ones((1, 256), np.float32)
    L.ns_hip_reset_error()
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(bad), nso.ptr(out), 1, 64, 256, 256, 64, None)
    assert b"DQ8_BNB" in L.ns_hip_last_error() or b"blob" in L.ns_hip_last_error()
