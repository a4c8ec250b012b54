#!/usr/bin/env python3
"""Mints tests/golden/ne_ops_golden.npz — known-answer vectors of the graph operators next to the GEMM path, produced by
the REAL reference: /root/reference/neural_speed/core/ne_layers.c compiled as is into oracle/_ref/libne_ref.so
(oracle/Makefile `neref`) and run through its own graph executor by oracle/ne_ref_harness.c.

  rope/*      ne_compute_forward_rope_f32 in every mode the product implements (plain, NeoX, YaRN, long-rope, GLM)
  attn/*      the reference's unfused attention graph (mul_mat -> scale -> diag_mask_inf -> soft_max -> mul_mat)

Run in the container that has /root/reference:   python tests/golden/make_ne_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import nso  # noqa: E402

ROPE = [  # name, shape (b, s, h, hs), kwargs of nso.neref_rope
    ("plain_llama", (1, 3, 8, 128), dict(n_past=17, n_dims=128, mode=0)),
    ("plain_scaled", (2, 2, 4, 64), dict(n_past=2000, n_dims=64, mode=0, freq_base=1000000.0, freq_scale=0.25)),
    ("neox_half", (1, 4, 4, 64), dict(n_past=5, n_dims=32, mode=2)),
    ("neox_partial", (1, 2, 2, 80), dict(n_past=9, n_dims=32, mode=2)),
    ("yarn", (1, 4, 4, 128), dict(n_past=3000, n_dims=128, mode=0, freq_scale=0.25, n_orig_ctx=4096, ext_factor=1.0,
                                   attn_factor=1.2, beta_fast=32.0, beta_slow=1.0)),
    ("yarn_neox", (1, 2, 4, 64), dict(n_past=5000, n_dims=64, mode=2, freq_scale=0.5, n_orig_ctx=4096, ext_factor=0.5,
                                       attn_factor=1.0, beta_fast=32.0, beta_slow=1.0)),
    ("longrope", (1, 3, 4, 96), dict(n_past=5000, n_dims=96, mode=0x10, freq_scale=0.5, n_orig_ctx=4096, ext_factor=0.0,
                                      attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, scale_factor=1.19, _factors=True)),
    ("glm_prompt", (1, 6, 4, 128), dict(n_past=0, n_dims=64, mode=4, prompt_size=6, n_padding=[0])),
    ("glm_decode", (2, 1, 8, 128), dict(n_past=9, n_dims=64, mode=4, prompt_size=7, n_padding=[0, 3])),
    ("glm_skip", (1, 5, 2, 64), dict(n_past=2, n_dims=32, mode=5, prompt_size=4, n_padding=[1])),
]
ATTN = [  # name, heads, heads_kv, hs, sl_q, sl_kv, causal
    ("mha_prefill", 4, 4, 64, 5, 5, True),
    ("gqa_decode", 8, 2, 128, 1, 37, True),
    ("gqa_chunk", 4, 2, 64, 6, 20, True),
    ("mqa_full", 4, 1, 32, 3, 9, False),
]


def main():
    assert nso.neref() is not None, "needs oracle/_ref/libne_ref.so (the reference tree)"
    out = {}
    for idx, (name, shape, kw) in enumerate(ROPE):
        rng = np.random.default_rng(777 + idx)
        x = rng.standard_normal(shape).astype(np.float32)
        kw = dict(kw)
        if kw.pop("_factors", False):
            kw["factors"] = (1.0 + rng.random(kw["n_dims"] // 2) * 3).astype(np.float32)
            out["rope/%s/factors" % name] = kw["factors"]
        out["rope/%s/x" % name] = x
        out["rope/%s/y" % name] = nso.neref_rope(x, **kw)
    for idx, (name, hn, hkv, hs, slq, slkv, causal) in enumerate(ATTN):
        rng = np.random.default_rng(888 + idx)
        q = rng.standard_normal((1, slq, hn, hs)).astype(np.float32)
        k = rng.standard_normal((1, slkv, hkv, hs)).astype(np.float16)
        v = rng.standard_normal((1, slkv, hkv, hs)).astype(np.float16)
        out["attn/%s/q" % name], out["attn/%s/k" % name], out["attn/%s/v" % name] = q, k, v
        out["attn/%s/dst" % name] = nso.neref_attn_unfused(q, k, v, hs ** -0.5, causal)
    path = os.path.join(HERE, "ne_ops_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
