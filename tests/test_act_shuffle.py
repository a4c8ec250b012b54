"""GPTQ act-order blobs: the int[K] ShuffleIndices section (bestla_storage.h:704, :761-765), its construction from g_idx
(setShuffleIndices, bestla_prologue_b.h:337-356) and the activation gather in front of the GEMM
(ShuffleActivationKBlockBase, bestla_prologue_a.h:322-330 -> kernel_ref.h:28-37).

CPU: the oracle's shuffled GEMM equals the plain product with the weights in their ORIGINAL row order — the property
the whole mechanism exists for.  GPU: pack, load and forward through the C-ABI against the oracle."""
import ctypes as C

import numpy as np
import pytest


def _gptq_case(nso, rng, n, k, bs, bits=4, asym=True):
    """original-order int weights + per-group scales/zps + a random balanced g_idx, as a GPTQ desc_act checkpoint"""
    full = 1 << (bits - 1)
    groups = k // bs
    g_idx = np.repeat(np.arange(groups, dtype=np.int32), bs)
    rng.shuffle(g_idx)
    q_orig = rng.integers(-full, full, (k, n), dtype=np.int8)
    sc = rng.uniform(0.002, 0.02, (groups, n)).astype(np.float32)
    zp = rng.integers(-full, full, (groups, n), dtype=np.int8) if asym else None
    w_orig = (q_orig.astype(np.float64) - (zp[g_idx].astype(np.float64) if asym else 0.0)) * sc[g_idx].astype(np.float64)
    q_sorted = nso.sort_rows_by_group(q_orig, g_idx, bs)
    return g_idx, q_orig, q_sorted, sc, zp, w_orig


@pytest.mark.parametrize("asym", [False, True])
@pytest.mark.parametrize("core", ["CORE_AVX512F", "CORE_AVX512_VNNI_KB"])
def test_oracle_shuffle_equals_original_order_gemm(nso, asym, core):
    rng = np.random.default_rng(12 + asym)
    n, k, bs = 40, 256, 32
    g_idx, q_orig, q_sorted, sc, zp, w_orig = _gptq_case(nso, rng, n, k, bs, asym=asym)
    blob = nso.pack_q(q_sorted, sc, zp, bs, nso.S4, nso.F32, getattr(nso, core), g_idx=g_idx)
    bi = nso.parse(blob)
    assert bi.has_shuffle and bi.shuf_bytes == 4 * k
    idx = blob[bi.shuf_off: bi.shuf_off + bi.shuf_bytes].view(np.int32)
    assert sorted(idx.tolist()) == list(range(k))                       # a permutation ...
    assert np.array_equal(g_idx[idx], np.repeat(np.arange(k // bs), bs))  # ... that sorts the channels by group
    assert all(np.all(np.diff(idx[g * bs:(g + 1) * bs]) > 0) for g in range(k // bs))  # original order inside a group
    # without the section the blob is byte-identical up to the trailing flag + section
    plain = nso.pack_q(q_sorted, sc, zp, bs, nso.S4, nso.F32, getattr(nso, core))
    pbi = nso.parse(plain)
    assert np.array_equal(blob[bi.q_off: bi.q_off + bi.q_bytes], plain[pbi.q_off: pbi.q_off + pbi.q_bytes])
    a = rng.standard_normal((3, k)).astype(np.float32)
    c = nso.gemm_f64(a, blob)
    want = a.astype(np.float64) @ w_orig
    assert np.max(np.abs(c - want)) <= 1e-6 * np.max(np.abs(want))   # the oracle dequantises in fp32
    # the unpack is the stored (sorted-order) matrix, like BTLAGemmUnPackB
    assert np.array_equal(nso.unpack_fp32(blob), nso.unpack_fp32(plain))


def test_gather_matches_reference_kernel(nso, refk):
    """kernel_ref.h:28-37 shuffle_activation, the real one"""
    rng = np.random.default_rng(3)
    m, k = 5, 96
    a = rng.standard_normal((m, k + 4)).astype(np.float32)
    idx = rng.permutation(k).astype(np.int32)
    out = np.zeros((m, k), np.float32)
    refk.ref_shuffle_activation(nso.ptr(a), nso.ptr(out), m, k, 0, 0, nso.ptr(idx), k + 4, k)
    assert np.array_equal(out, a[:, idx])


@pytest.mark.gpu
@pytest.mark.parametrize("bits,asym,st", [(4, True, "F32"), (4, False, "BF16"), (8, False, "BF16"), (3, True, "F32")])
def test_hip_pack_load_forward_with_g_idx(L, pkg, nso, bits, asym, st):
    rng = np.random.default_rng(100 + bits)
    n, k, bs = 144, 512, 32
    g_idx, q_orig, q_sorted, sc, zp, w_orig = _gptq_case(nso, rng, n, k, bs, bits=bits, asym=asym)
    qt, sdt = nso.INT_TYPES[bits], getattr(nso, st)
    gi = np.ascontiguousarray(g_idx)
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, sdt, asym, pkg.COMP_F32, nso.ptr(gi))
    L.ns_set_pack_core(nso.CORE_AVX512F)
    try:
        ref = nso.pack_q(q_sorted, sc, zp, bs, qt, sdt, nso.CORE_AVX512F, g_idx=g_idx)
        assert size == ref.size
        blob = nso.aligned_bytes(size)
        assert L.ns_BTLAGemmPackB(nso.ptr(blob), nso.ptr(q_sorted), nso.ptr(sc), nso.ptr(zp) if asym else None, n, k, n, bs,
                                  qt, sdt, asym, pkg.COMP_F32, nso.ptr(gi), None), pkg.last_error()
    finally:
        L.ns_set_pack_core(pkg.CORE_AUTO)
    assert np.array_equal(blob, ref)
    for m in (1, 5, 70, 130):
        a = rng.standard_normal((m, k + 2)).astype(np.float32)
        out = np.zeros((m, n), np.float32)
        L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k + 2, n, None)
        a_c = np.ascontiguousarray(a[:, :k])
        assert nso.rel_l2(out, nso.gemm_f64(a_c, blob)) < 1e-3
        if st == "F32":  # exact scales: also against the checkpoint's original-order weights
            assert nso.rel_l2(out, a_c.astype(np.float64) @ w_orig) < 1e-3
    # QKV fusion is declined for such weights, as in the reference (ip_fusion_qkv.cpp:174-176); the FFN entry works
    assert not L.bestla_fusion_QKV_f32f32_support(nso.ptr(blob), nso.ptr(blob), nso.ptr(blob), 1, n, k)


@pytest.mark.gpu
def test_hip_ffn_and_slices_with_g_idx(L, pkg, nso):
    rng = np.random.default_rng(9)
    fin, fmid, bs, m = 256, 320, 32, 3
    blobs, gs = [], []
    for (n, k) in ((fmid, fin), (fin, fmid), (fmid, fin)):   # w1, w2, w3
        g_idx, _, q_sorted, sc, zp, _ = _gptq_case(nso, rng, n, k, bs, asym=False)
        blobs.append(nso.pack_q(q_sorted, sc, None, bs, nso.S4, nso.F32, nso.CORE_AVX512F, g_idx=g_idx))
    w1, w2, w3 = blobs
    a = rng.standard_normal((m, fin)).astype(np.float32)
    assert L.bestla_fusion_FFN_SiLu_f32f32_support(nso.ptr(w1), nso.ptr(w2), nso.ptr(w3), m, fin, fmid, fin)
    t1 = np.zeros((m, fmid), np.float32)
    t2 = np.zeros((m, fmid), np.float32)
    out = np.zeros((m, fin), np.float32)
    L.bestla_fusion_FFN_SiLu_f32f32_forward(nso.ptr(a), nso.ptr(w1), nso.ptr(w2), nso.ptr(w3), nso.ptr(t1), nso.ptr(t2),
                                            nso.ptr(out), m, fin, fmid, fin, None)
    g = nso.gemm_f64(a, w1)
    u = nso.gemm_f64(a, w3)
    h = (g / (1 + np.exp(-g)) * u).astype(np.float32)
    assert nso.rel_l2(out, nso.gemm_f64(h, w2)) < 2e-3
    # TP: column slices keep the shuffle, row (K) slices are refused
    wt = pkg.Weight.from_host_blob(nso.ptr(w1))
    half = wt.slice(0, fmid // 2 // 16 * 16, 0, fin)
    import torch
    da = torch.from_numpy(a).cuda()
    dc = torch.zeros((m, half.n), device="cuda")
    pkg.check(L.ns_hip_f32f32_forward(da.data_ptr(), half.h, dc.data_ptr(), m, fin, half.n, 0, None, 0, None))
    torch.cuda.synchronize()
    assert nso.rel_l2(dc.cpu().numpy(), g[:, :half.n]) < 1e-3
    with pytest.raises(Exception):
        wt.slice(0, fmid, 0, fin // 2)
