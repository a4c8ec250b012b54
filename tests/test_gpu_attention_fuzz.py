"""Seeded shape fuzz of the fused attention entry against the oracle's restatement of bestla_fusion_attn_forward_ref: the launcher picks
among the context-split decode kernel (any head group since round 4), the 64-row and the 128-row matrix-core prefill kernels (exact,
padded, biased, head size 256) and the generic kernel by shape — every draw must agree with the reference whichever it lands on."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _draw(rng):
    hs = int(rng.choice([32, 40, 64, 80, 96, 128, 160, 256]))
    hkv = int(rng.choice([1, 2, 3, 4, 8]))
    g = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 12]))
    kind = rng.integers(0, 3)
    if kind == 0:      # decode rows
        sl_q, sl_kv = int(rng.integers(1, 4)), int(rng.integers(1, 2500))
    elif kind == 1:    # a prompt
        sl_q = int(rng.integers(16, 300))
        sl_kv = sl_q
    else:              # a prompt chunk behind cached positions
        sl_q = int(rng.integers(16, 200))
        sl_kv = sl_q + int(rng.integers(1, 700))
    sl_kv = max(sl_kv, sl_q)
    flags = int(rng.choice([1, 1, 1, 0, 3, 2]))
    bs = int(rng.choice([1, 1, 2]))
    return bs, hkv * g, hkv, hs, sl_q, sl_kv, flags


@pytest.mark.parametrize("seed", range(48))
def test_random_shape_against_the_reference(L, pkg, nso, seed):
    rng = np.random.default_rng(9000 + seed)
    bs, hn, hkv, hs, sl_q, sl_kv, flags = _draw(rng)
    while bs * hn * sl_q * sl_kv * hs > 3.5e8:  # keep the oracle's loop in seconds
        sl_kv = max(sl_q, sl_kv // 2)
        if bs * hn * sl_q * sl_kv * hs > 3.5e8:
            sl_q = max(1, sl_q // 2)
            sl_kv = max(sl_kv, sl_q)
    q = rng.standard_normal((bs, sl_q, hn, hs)).astype(np.float32)
    k = rng.standard_normal((bs, sl_kv, hkv, hs)).astype(np.float16)
    v = rng.standard_normal((bs, sl_kv, hkv, hs)).astype(np.float16)
    scale = float(1.0 / np.sqrt(hs))
    ref = nso.attn_ref(q, k, v, scale, flags)
    out = np.full(q.shape, 7.0, np.float32)
    a = pkg.attn_args(q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, bs, hn, hkv, hs, sl_q, sl_kv, scale, flags, False)
    L.bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(C.byref(a))
    assert np.all(np.isfinite(out)), (bs, hn, hkv, hs, sl_q, sl_kv, flags)
    e = nso.rel_l2(out, ref)
    assert e < TOL, (e, bs, hn, hkv, hs, sl_q, sl_kv, flags)
