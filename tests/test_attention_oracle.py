"""CPU: the oracle's restatement of bestla_fusion_attn_forward_ref (mha_dense_wrapper.h:1371-1517) against (1) an
independent fp64 softmax(QK^T)V, (2) the reference's unfused attention graph run by its ne_layers.c, and (3) — round 3 —
THE FUNCTION ITSELF: mha_dense_wrapper.h as a whole needs the JIT headers, the function does not, so oracle/Makefile
attnref cuts it out of the reference file at build time and compiles it (oracle/_ref/libattn_ref.so).  With the
reference's polynomial exp selected the restatement reproduces it BIT FOR BIT in every mode; golden rows minted from
the function travel to the GPU box (tests/golden/attn_forward_ref.npz)."""
import numpy as np
import pytest


def _np_attention(q, k, v, scale, causal, alibi):
    bs, sl_q, hn, hs = q.shape
    sl_kv, hkv = k.shape[1], k.shape[2]
    g = hn // hkv
    out = np.zeros(q.shape, np.float64)
    lf = 1 << int(np.floor(np.log2(hn)))
    m0, m1 = 2.0 ** (-8.0 / lf), 2.0 ** (-4.0 / lf)
    for b in range(bs):
        for h in range(hn):
            kk = k[b, :, h // g].astype(np.float64)
            vv = v[b, :, h // g].astype(np.float64)
            s = q[b, :, h].astype(np.float64) @ kk.T * scale
            if alibi:
                slope = m0 ** (h + 1) if h < lf else m1 ** (2 * (h - lf) + 1)
                s = s + np.arange(sl_kv)[None, :] * slope
            if causal:
                i = np.arange(sl_q)[:, None]
                j = np.arange(sl_kv)[None, :]
                s = np.where(j <= i + (sl_kv - sl_q), s, -np.inf)
            p = np.exp(s - s.max(axis=1, keepdims=True))
            p /= p.sum(axis=1, keepdims=True)
            out[b, :, h] = p @ vv
    return out


CASES = [  # bs, heads, heads_kv, head_size, sl_q, sl_kv, flags
    (1, 4, 4, 64, 1, 37, 0),
    (2, 8, 2, 128, 5, 5, 1),
    (1, 6, 3, 80, 3, 70, 1),
    (1, 8, 8, 32, 7, 64, 3),
    (1, 2, 1, 256, 2, 9, 0),
]


@pytest.mark.parametrize("bs,hn,hkv,hs,sl_q,sl_kv,flags", CASES)
def test_attn_oracle_matches_fp64_softmax(nso, bs, hn, hkv, hs, sl_q, sl_kv, flags):
    rng = np.random.default_rng(hs + sl_kv)
    q = rng.standard_normal((bs, sl_q, hn, hs)).astype(np.float32)
    k = rng.standard_normal((bs, sl_kv, hkv, hs)).astype(np.float16)
    v = rng.standard_normal((bs, sl_kv, hkv, hs)).astype(np.float16)
    scale = 1.0 / np.sqrt(hs)
    ref = _np_attention(q, k, v, scale, bool(flags & 1), bool(flags & 2))
    out = nso.attn_ref(q, k, v, scale, flags)
    assert nso.rel_l2(out, ref) < 2e-6
    # transposed K (step_k_head_size != 1) is the same tensor
    kt = np.ascontiguousarray(k.transpose(0, 2, 3, 1))
    assert np.array_equal(nso.attn_ref(q, kt, v, scale, flags, k_trans=True), out)
    # the reference's default bf16 rounding of Q, K, P stays within its own test tolerance of the fp32 form
    assert np.max(np.abs(nso.attn_ref(q, k, v, scale, flags, bf16_gemm=True) - out)) < 3e-2


# ---------------------------------------------------------------- pinned against the reference's own (unfused) graph
# bestla_fusion_attn_forward_ref lives behind xbyak-dependent headers and cannot be compiled here, but the SAME operator
# spelled as graph nodes can: mul_mat(K, Q) -> scale -> diag_mask_inf -> soft_max -> mul_mat(V^T, P), executed by the
# reference's ne_layers.c (oracle/_ref/libne_ref.so).  That pins what a restatement can get wrong — mask placement for
# sl_q != sl_kv, the scale, the GQA head mapping, the normalisation — against real reference code.  Tolerance: the
# reference's soft_max goes through an fp16 exp table, i.e. ~2e-4 relative on the result.
import numpy as _np
import pytest as _pytest


@_pytest.mark.parametrize("hn,hkv,hs,slq,slkv,causal", [(4, 4, 64, 5, 5, True), (8, 2, 128, 1, 37, True),
                                                       (4, 2, 64, 6, 20, True), (4, 1, 32, 3, 9, False),
                                                       (32, 32, 128, 2, 70, True)])
def test_attention_oracle_matches_reference_unfused_graph(nso, hn, hkv, hs, slq, slkv, causal):
    if nso.neref() is None:
        _pytest.skip("oracle/_ref/libne_ref.so not built (reference tree absent)")
    rng = _np.random.default_rng(hn * 100 + slkv)
    q = rng.standard_normal((1, slq, hn, hs)).astype(_np.float32)
    k = rng.standard_normal((1, slkv, hkv, hs)).astype(_np.float16)
    v = rng.standard_normal((1, slkv, hkv, hs)).astype(_np.float16)
    scale = hs ** -0.5
    ours = nso.attn_ref(q, k, v, scale, 1 if causal else 0)
    ref = nso.neref_attn_unfused(q, k, v, scale, causal)
    assert nso.rel_l2(ours, ref) < 1e-3
    # a wrong mask offset / head mapping is an O(1) error, not 1e-4: make the test's power explicit
    if causal and slq > 1:
        wrong = nso.attn_ref(q, k, v, scale, 0)
        assert nso.rel_l2(wrong, ref) > 1e-2


# ---- (3) the reference's own function ---------------------------------------------------------------------------------
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("make_attn_golden", os.path.join(os.path.dirname(__file__), "golden", "make_attn_golden.py"))
_gold = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gold)
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "attn_forward_ref.npz")


@pytest.mark.parametrize("idx", range(len(_gold.CASES)))
def test_restatement_equals_the_reference_function_bit_for_bit(nso, idx):
    """every mode of bestla_fusion_attn_forward_ref<float, fp16, fp16, float>: PREFER_FP32 and the default (Q, P rounded to bf16
    to nearest even; K, V through fp16::operator bf16 — truncation, subnormals flushed — when K is not transposed,
    IS_BF16_GEMM :1389-1394), plain and transposed K, causal with sl_q < sl_kv, GQA, alibi"""
    if nso.attnref() is None:
        pytest.skip("oracle/_ref/libattn_ref.so absent (needs the reference tree)")
    bs, hn, hkv, hs, slq, slkv, causal, alibi = _gold.CASES[idx]
    q, k, v = _gold.case_inputs(idx)
    sc = 1.0 / np.sqrt(hs)
    flags = (1 if causal else 0) | (2 if alibi else 0)
    for kt in (False, True):
        kk = np.ascontiguousarray(k.transpose(0, 2, 3, 1)) if kt else k
        for pf in (True, False):
            ref = nso.attn_reference(q, kk, v, sc, causal=causal, alibi8=alibi, prefer_fp32=pf, k_trans=kt)
            mine = nso.attn_ref(q, kk, v, sc, flags=flags, k_trans=kt, bf16_gemm=(not pf) and (not kt), ref_exp=True)
            assert np.array_equal(ref.view(np.int32), mine.view(np.int32)), (idx, kt, pf, float(np.abs(ref - mine).max()))
            # the exact exp (what the product evaluates and is checked against) stays within the polynomial's own error
            exact = nso.attn_ref(q, kk, v, sc, flags=flags, k_trans=kt, bf16_gemm=(not pf) and (not kt), ref_exp=False)
            assert nso.rel_l2(exact, ref) < 4e-3, (idx, kt, pf)


@pytest.mark.parametrize("idx", range(len(_gold.CASES)))
def test_restatement_equals_the_golden_rows_of_the_reference_function(nso, idx):
    """the same pin where the reference tree (and oracle/_ref) is absent: rows minted by tests/golden/make_attn_golden.py"""
    g = np.load(GOLDEN)
    bs, hn, hkv, hs, slq, slkv, causal, alibi = _gold.CASES[idx]
    q, k, v = _gold.case_inputs(idx)
    rows = _gold.sample_rows(slq)
    sc = 1.0 / np.sqrt(hs)
    flags = (1 if causal else 0) | (2 if alibi else 0)
    for kt in (False, True):
        kk = np.ascontiguousarray(k.transpose(0, 2, 3, 1)) if kt else k
        for pf in (True, False):
            mine = nso.attn_ref(q, kk, v, sc, flags=flags, k_trans=kt, bf16_gemm=(not pf) and (not kt), ref_exp=True)
            want = g["c%d_kt%d_fp32%d" % (idx, int(kt), int(pf))]
            assert np.array_equal(want.view(np.int32), mine[:, rows].view(np.int32)), (idx, kt, pf)
