"""bestla_device_mha_f32's kernel on the reference device backend's cache layout (fp32 K [batch][heads_kv][n_ctx][head_size], V
TRANSPOSED [batch][heads_kv][head_size][n_ctx]; ne_bestla_sycl.cpp:560-700) against an fp64 softmax(QK^T)V: the single-workgroup
form (contexts up to 128 keys, other head sizes), the context-split form of round 4 (head sizes 64 / 128 / 256 from 129 keys on) and, for
prompts of 32 rows and more, the matrix-core prefill kernels on fp16 copies of the call's K / V rows (round 5)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [  # batch, seq, seq_all, heads, heads_kv, head_size, n_ctx, masked
    (1, 1, 2048, 32, 32, 128, 2048, 1),   # Llama-2-7B decode at the full context
    (1, 1, 777, 8, 2, 64, 1024, 1),       # GQA, ragged last range
    (1, 5, 300, 4, 4, 128, 512, 1),       # a few prompt rows: causal extents differ, some ranges are empty for the first rows
    (2, 1, 513, 4, 1, 256, 600, 0),       # batch 2, largest head, unmasked
    (1, 3, 1001, 4, 2, 128, 1001, 1),     # n_ctx not a multiple of 4: element-wise V reads
    (1, 2, 100, 4, 4, 128, 256, 1),       # one range: the single-workgroup kernel
    (1, 1, 400, 6, 3, 80, 512, 1),        # head size outside the split kernel's set
    (1, 700, 900, 32, 32, 128, 1024, 1),  # a prompt: K / V rows converted to fp16 once, matrix-core prefill attention (round 5; 32 rows and up)
    (1, 64, 64, 8, 8, 128, 128, 1),       # a first prompt: the 64-row matrix-core kernel
    (1, 200, 333, 8, 2, 64, 512, 1),      # GQA, chunked prefill (rows < keys), head size 64
    (2, 130, 130, 4, 4, 96, 256, 0),      # batch 2, head size 96 (padded instantiation), unmasked
    (1, 40, 1000, 4, 4, 256, 1024, 1),    # head size 256, 40 rows over a long context
    (1, 31, 400, 4, 4, 128, 512, 1),      # 31 rows
    (1, 1, 1, 32, 32, 128, 2048, 1),      # the first position
]


@pytest.mark.parametrize("kv16", [1, 0])
@pytest.mark.parametrize("batch,seq,seq_all,heads,hkv,hs,n_ctx,masked", CASES)
def test_device_layout_attention_against_fp64(L, pkg, batch, seq, seq_all, heads, hkv, hs, n_ctx, masked, kv16):
    """kv16 = 1 (default, round 6): K / V are read from the fp16 mirror of the cache (csrc/ns_route.h) by this library's attention kernels, for every
    call shape — the fp64 model is computed on fp16-rounded K / V, as with the reference's own default (fp16) caches; kv16 = 0 (NS_DEVICE_KV=f32):
    the fp32 kernels on the fp32 cache, the numerics of the reference's device kernel."""
    import torch
    rng = np.random.default_rng(seq_all + hs)
    q = rng.standard_normal((batch, seq, heads, hs)).astype(np.float32)
    k = np.full((batch, hkv, n_ctx, hs), np.nan, np.float32)   # cells past seq_all hold NaN: they must never reach the result
    v = np.full((batch, hkv, hs, n_ctx), np.nan, np.float32)
    k[:, :, :seq_all] = rng.standard_normal((batch, hkv, seq_all, hs)).astype(np.float32)
    v[:, :, :, :seq_all] = rng.standard_normal((batch, hkv, hs, seq_all)).astype(np.float32)
    scale = float(hs ** -0.5)
    ref = np.zeros(q.shape, np.float64)
    g = heads // hkv
    rnd = (lambda a: a.astype(np.float16).astype(np.float64)) if kv16 else (lambda a: a.astype(np.float64))
    for b in range(batch):
        for h in range(heads):
            kk, vv = rnd(k[b, h // g, :seq_all]), rnd(v[b, h // g, :, :seq_all])
            for i in range(seq):
                vis = min(seq_all, i + (seq_all - seq) + 1) if masked else seq_all
                s = kk[:vis] @ q[b, i, h].astype(np.float64) * scale
                p = np.exp(s - s.max())
                ref[b, i, h] = (vv[:, :vis] @ p) / p.sum()
    dq, dk, dv = torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
    out = torch.full(q.shape, 7.0, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.ns_hip_mha_f32_device_layout.restype = C.c_int
    L.ns_hip_mha_f32_device_layout.argtypes = [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_float, C.c_int, C.c_void_p]
    L.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
    assert L.ns_hip_set_tuning(b"device_kv_f16", kv16) == 0
    try:
        for rep in range(2):  # the second call re-uses the per-stream workspace (and the mirror)
            out.fill_(7.0)
            pkg.check(L.ns_hip_mha_f32_device_layout(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), out.data_ptr(), batch, seq, seq_all, heads, hkv, hs,
                                                     n_ctx, scale, masked, st))
            torch.cuda.synchronize()
            got = out.cpu().numpy().astype(np.float64)
            assert np.all(np.isfinite(got))
            err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            # fp32 kernels: 2e-6.  Mirror: the decode kernels compute in fp32 on the fp16 rows (5e-6 against the model on the same rounded rows); from 16
            # query rows on the matrix-core prefill kernels also round Q and the probabilities to fp16: 1e-3
            tol = 2e-6 if not kv16 else (1e-3 if seq >= 16 else 5e-6)
            assert err < tol, (err, tol)
    finally:
        L.ns_hip_set_tuning(b"device_kv_f16", -1)
