"""CPU: the oracle's restatement of bestla_fusion_attn_forward_ref (mha_dense_wrapper.h:1371-1517) against an
independent fp64 softmax(QK^T)V.  The reference itself cannot be compiled here (xbyak), so this is what pins the
attention oracle — stated as "parity unpinned" in oracle/ns_oracle.h and DESIGN.md."""
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
