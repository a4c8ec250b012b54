#!/usr/bin/env python3
"""Mint tests/golden/attn_forward_ref.npz from the reference's OWN bestla_fusion_attn_forward_ref<float, fp16, fp16, float>
(mha_dense_wrapper.h:1370-1514, built by `make -C oracle attnref` from the reference tree; needs /root/reference).
Cases = the fp32/fp16 shapes of the reference's test suite (mha_dense_tests.cpp:43-62, :92-112) + GQA / decode shapes, in
PREFER_FP32 and default (bf16-rounded) mode, plain and transposed K.  Inputs are NOT stored: they are re-derived from the
seed by case_inputs() below (numpy's PCG64 stream); the file holds the reference's output rows for a sample of query
positions of every head.  Run: python tests/golden/make_attn_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# bs, heads, heads_kv, head_size, sl_q, sl_kv, causal, alibi8
CASES = [
    (1, 1, 1, 32, 128, 64, False, False), (2, 5, 5, 32, 64, 128, False, False), (2, 5, 5, 80, 128, 77, False, False),
    (1, 1, 1, 256, 63, 63, False, False), (3, 4, 4, 256, 1, 384, False, False), (1, 1, 1, 64, 64, 64, True, False),
    (1, 8, 2, 128, 1, 300, True, False), (1, 8, 8, 64, 9, 40, True, True), (2, 6, 3, 80, 5, 70, True, False),
]


def case_inputs(idx):
    bs, hn, hkv, hs, slq, slkv, _, _ = CASES[idx]
    rng = np.random.default_rng(4242 + idx)
    q = rng.standard_normal((bs, slq, hn, hs)).astype(np.float32)
    k = rng.standard_normal((bs, slkv, hkv, hs)).astype(np.float16)
    v = rng.standard_normal((bs, slkv, hkv, hs)).astype(np.float16)
    k.reshape(-1)[::97] = np.float16(2e-6)  # fp16 subnormals: the default mode flushes them (fp16::operator bf16)
    return q, k, v


def sample_rows(slq):
    return sorted({0, slq // 2, slq - 1})


def main():
    from oracle import nso
    assert nso.attnref() is not None, "build oracle/_ref/libattn_ref.so first (make -C oracle attnref)"
    out = {}
    for i, (bs, hn, hkv, hs, slq, slkv, causal, alibi) in enumerate(CASES):
        q, k, v = case_inputs(i)
        sc = 1.0 / np.sqrt(hs)
        rows = sample_rows(slq)
        for kt in (False, True):
            kk = np.ascontiguousarray(k.transpose(0, 2, 3, 1)) if kt else k
            for pf in (True, False):
                d = nso.attn_reference(q, kk, v, sc, causal=causal, alibi8=alibi, prefer_fp32=pf, k_trans=kt)
                out["c%d_kt%d_fp32%d" % (i, int(kt), int(pf))] = d[:, rows].copy()
    path = os.path.join(ROOT, "tests", "golden", "attn_forward_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
