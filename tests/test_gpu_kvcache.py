"""GPU parity of the library-managed kv-cache entry points (mha_dense.h:124-172 —
bestla_reordered_attn_fp32_batch_kv_info / _update_k / _update_v / _shift_rope_k / _forward and
bestla_fusion_attn_fp32_batch_cpy_k / _v), called through the C ABI the way the reference's graph code calls them
(ne_layers.c:10216-10280, :10529-10558; models/llama/llama.cpp:496-571).  The cache layout is the library's own
(the reference's is CPU tile packing): what is compared is the observable behaviour — what attention reads back after the
updates — against the oracle's attention over fp16-rounded K / V, and the shift against the restated fp16 arithmetic
(ne_layers.c:9494-9530).  The same entries driven by the reference's own graph nodes: tests/tools/ref_graph_worker.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kv_info(L, pkg, hkv, hs, n_ctx):
    info = pkg.KvCacheInfo()
    shape = pkg.KvShape(hkv, hs, n_ctx)
    L.bestla_reordered_attn_fp32_batch_kv_info(C.byref(shape), C.byref(info))
    return info


def update(L, pkg, fn, cache, cur, seq_off, n_ctx, graph_layout=True):
    """cur fp32 [bs][seq][heads_kv][hs] (the graph's (hs, heads_kv, seq, bs) tensor) or, graph_layout=False, a
    [bs][heads_kv][seq][hs] buffer: exercises the four element steps"""
    bs, seq, hkv, hs = cur.shape
    a = pkg.KvUpdateArgs()
    if graph_layout:
        buf = np.ascontiguousarray(cur)
        a.step_bs, a.step_seq, a.step_head_num, a.step_head_size = seq * hkv * hs, hkv * hs, hs, 1
    else:
        buf = np.ascontiguousarray(cur.transpose(0, 2, 1, 3))
        a.step_bs, a.step_head_num, a.step_seq, a.step_head_size = seq * hkv * hs, seq * hs, hs, 1
    a.src, a.cache = buf.ctypes.data, cache.ctypes.data
    a.batch_size, a.heads_kv, a.head_size, a.seq_off, a.seq_size, a.seq_max = bs, hkv, hs, seq_off, seq, n_ctx
    a.no_zeroing = False
    getattr(L, fn)(C.byref(a))


def forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, sl_q, sl_kv, scale, flags):
    out = np.full(q.shape, 7.0, np.float32)
    ws = np.zeros(L.bestla_fusion_attn_workspace_size(C.byref(pkg.AttnShape(bs, hn, hkv, hs, sl_q, sl_kv))), np.uint8)
    a = pkg.ReorderedAttnArgs()
    a.Q, a.K, a.V, a.dst, a.tmp = q.ctypes.data, kc.ctypes.data, vc.ctypes.data, out.ctypes.data, ws.ctypes.data
    a.Q_sc = a.K_sc = a.V_sc = a.dst_sc = 1.0
    a.QK_scale, a.attn_flags = scale, flags
    a.batch_size, a.head_num, a.heads_kv, a.head_size, a.sl_q, a.sl_kv = bs, hn, hkv, hs, sl_q, sl_kv
    a.Q_layout = a.dst_layout = 0
    a.K_layout, a.V_layout = info.k_layout, info.v_layout
    a.step_q_bs, a.step_q_head_num, a.step_q_sl = sl_q * hn * hs, hs, hn * hs
    # what ne_compute_forward_flash_attn_reordered passes (ne_layers.c:10264-10272): nb[] of the views, in BYTES
    a.stride_k_bs, a.stride_k_head_num, a.stride_k_sl, a.stride_k_head_size = info.k_bytes, info.stride_k_head_num, info.stride_k_sl, 0
    a.stride_v_bs, a.stride_v_head_num, a.stride_v_sl, a.stride_v_head_size = info.v_bytes, info.stride_v_head_num, 0, info.stride_v_head_size
    a.step_dst_bs, a.step_dst_head_num, a.step_dst_sl = sl_q * hn * hs, hs, hn * hs
    L.bestla_reordered_attn_fp32_forward(C.byref(a))
    return out


def test_kv_info_describes_one_slab_per_head(L, pkg):
    info = kv_info(L, pkg, 8, 128, 2048)
    assert info.k_bytes == info.v_bytes == 8 * 2048 * 128 * 2
    assert (info.stride_k_head_num, info.stride_k_sl, info.stride_k_head_size) == (2048 * 256, 256, 2)
    assert (info.stride_v_head_num, info.stride_v_sl, info.stride_v_head_size) == (2048 * 256, 256, 2)
    assert info.k_layout == info.v_layout == 0


@pytest.mark.parametrize("bs,hn,hkv,hs,n_ctx,chunks,flags", [
    (1, 32, 32, 128, 64, (7, 1, 1, 1), 1),     # prompt, then single-token appends (Llama-2-7B head shape)
    (2, 8, 2, 64, 40, (5, 20, 1), 1),          # batch 2, GQA, chunked prefill then decode
    (1, 4, 4, 80, 33, (33,), 1),               # cache filled to n_ctx in one call, head size not a multiple of 64
    (1, 6, 3, 256, 20, (3, 2), 0),             # unmasked, largest head
])
def test_update_then_forward_matches_attention_over_fp16_rows(L, pkg, nso, bs, hn, hkv, hs, n_ctx, chunks, flags):
    rng = np.random.default_rng(hs + n_ctx)
    info = kv_info(L, pkg, hkv, hs, n_ctx)
    kc = np.full(bs * info.k_bytes, 0x7f, np.uint8)  # garbage: rows past what was appended must never be read
    vc = np.full(bs * info.v_bytes, 0x7f, np.uint8)
    total = sum(chunks)
    kf = rng.standard_normal((bs, total, hkv, hs)).astype(np.float32)
    vf = rng.standard_normal((bs, total, hkv, hs)).astype(np.float32)
    scale = float(hs ** -0.5)
    off = 0
    for i, n in enumerate(chunks):
        update(L, pkg, "bestla_reordered_attn_fp32_update_k", kc, kf[:, off:off + n], off, n_ctx, graph_layout=i % 2 == 0)
        update(L, pkg, "bestla_reordered_attn_fp32_update_v", vc, vf[:, off:off + n], off, n_ctx, graph_layout=i % 2 == 1)
        off += n
        q = rng.standard_normal((bs, n, hn, hs)).astype(np.float32)
        out = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, n, off, scale, flags)
        ref = nso.attn_ref(q, kf[:, :off].astype(np.float16), vf[:, :off].astype(np.float16), scale, flags)
        e = nso.rel_l2(out, ref)
        assert np.all(np.isfinite(out)) and e < 1e-3, (i, e)
    # the cache holds exactly round-to-nearest fp16 of what was appended, at [batch][head][row][:]; later rows untouched
    k16 = kc.view(np.float16).reshape(bs, hkv, n_ctx, hs)
    assert np.array_equal(k16[:, :, :total].view(np.uint16), kf.astype(np.float16).transpose(0, 2, 1, 3).view(np.uint16))
    assert np.all(kc.reshape(bs, hkv, n_ctx, hs * 2)[:, :, total:] == 0x7f)


def test_shift_rope_k_rotates_rows_past_n_keep(L, pkg, nso):
    bs, hkv, hs, n_ctx, n_keep, shift = 2, 3, 128, 24, 4, 3
    rng = np.random.default_rng(11)
    info = kv_info(L, pkg, hkv, hs, n_ctx)
    kc = np.zeros(bs * info.k_bytes, np.uint8)
    kf = rng.standard_normal((bs, n_ctx, hkv, hs)).astype(np.float32)
    update(L, pkg, "bestla_reordered_attn_fp32_update_k", kc, kf, 0, n_ctx)
    want, cossin = nso.rope_shift_f16_ref(kf.astype(np.float16), shift, n_keep)
    L.bestla_reordered_attn_fp32_shift_rope_k(kc.ctypes.data, cossin.ctypes.data, bs, hkv, hs, n_ctx, n_keep)
    got = kc.view(np.float16).reshape(bs, hkv, n_ctx, hs).transpose(0, 2, 1, 3)
    assert np.array_equal(got[:, :n_keep].view(np.uint16), want[:, :n_keep].view(np.uint16))
    # fp32 arithmetic rounded once to fp16; the device contracts a*b+c*d into an fma: at most one fp16 ulp apart
    d = np.abs(got.astype(np.float32) - want.astype(np.float32))
    assert np.all(d <= np.spacing(np.abs(want).astype(np.float16)).astype(np.float32)), d.max()
    assert (got.view(np.uint16) != want.view(np.uint16)).mean() < 0.01


def test_batch_cpy_copies_the_requested_rows_of_every_head(L, pkg):
    hkv, hs, n_ctx = 4, 64, 32
    info = kv_info(L, pkg, hkv, hs, n_ctx)
    rng = np.random.default_rng(5)
    for fn, nbytes in (("bestla_fusion_attn_fp32_batch_cpy_k", info.k_bytes), ("bestla_fusion_attn_fp32_batch_cpy_v", info.v_bytes)):
        src = rng.integers(0, 256, nbytes, dtype=np.uint8)
        dst = rng.integers(0, 256, nbytes, dtype=np.uint8)
        keep = dst.copy()
        a = pkg.KvBatchCpyArgs(src.ctypes.data, dst.ctypes.data, hkv, hs, 5, 9, n_ctx, False)
        getattr(L, fn)(C.byref(a))
        s3, d3, k3 = (x.reshape(hkv, n_ctx, hs * 2) for x in (src, dst, keep))
        assert np.array_equal(d3[:, 5:14], s3[:, 5:14])
        assert np.array_equal(d3[:, :5], k3[:, :5]) and np.array_equal(d3[:, 14:], k3[:, 14:])


def test_forward_refuses_a_foreign_layout(L, pkg):
    bs, hn, hkv, hs, n_ctx = 1, 4, 4, 64, 8
    info = kv_info(L, pkg, hkv, hs, n_ctx)
    kc = np.zeros(info.k_bytes, np.uint8)
    vc = np.zeros(info.v_bytes, np.uint8)
    q = np.ones((bs, 1, hn, hs), np.float32)
    info.k_layout = 2  # ATTN_FWD_LAYOUT_NTILE48_ROWPACK2: a cache some other library packed
    out = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, 4, 0.125, 1)
    assert np.all(out == 7.0)
    assert b"not laid out by this library" in L.ns_hip_last_error()


def test_device_mirror_of_the_cache_serves_the_same_numbers_as_the_upload_path(L, pkg, nso):
    """A library-managed cache is mirrored on the device from its first update on (the attention entry then uploads Q only).
    Same result with the mirror and, after ns_hip_cache_clear() dropped it, through the upload path; a beam copy and a shift
    reach the mirror too; bytes written to the host copy behind the library's back are seen after a clear (the documented
    contract), not before."""
    rng = np.random.default_rng(5)
    bs, hn, hkv, hs, n_ctx = 1, 8, 8, 128, 96
    info = kv_info(L, pkg, hkv, hs, n_ctx)
    kc = np.zeros(info.k_bytes * bs, np.uint8)
    vc = np.zeros(info.v_bytes * bs, np.uint8)
    scale = float(hs) ** -0.5
    n = 0
    for chunk in (40, 1, 1):
        cur_k = rng.standard_normal((bs, chunk, hkv, hs)).astype(np.float32)
        cur_v = rng.standard_normal((bs, chunk, hkv, hs)).astype(np.float32)
        update(L, pkg, "bestla_reordered_attn_fp32_update_k", kc, cur_k, n, n_ctx)
        update(L, pkg, "bestla_reordered_attn_fp32_update_v", vc, cur_v, n, n_ctx)
        n += chunk
    q = rng.standard_normal((bs, 1, hn, hs)).astype(np.float32)
    with_mirror = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, n, scale, 1)
    L.ns_hip_cache_clear()
    uploaded = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, n, scale, 1)
    assert np.array_equal(with_mirror.view(np.int32), uploaded.view(np.int32))
    # re-establish the mirrors (an update does), then write one cached K row on the host behind the library's back
    cur = rng.standard_normal((bs, 1, hkv, hs)).astype(np.float32)
    update(L, pkg, "bestla_reordered_attn_fp32_update_k", kc, cur, n, n_ctx)
    update(L, pkg, "bestla_reordered_attn_fp32_update_v", vc, cur, n, n_ctx)
    n += 1
    before = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, n, scale, 1)
    rows = kc.view(np.float16).reshape(bs, hkv, n_ctx, hs)
    rows[0, :, 3, :] *= np.float16(-2.0)
    stale = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, n, scale, 1)
    assert np.array_equal(before.view(np.int32), stale.view(np.int32))  # the mirror still holds what the library wrote
    L.ns_hip_cache_clear()
    fresh = forward(L, pkg, q, kc, vc, info, bs, hn, hkv, hs, 1, n, scale, 1)
    kh = rows[0].astype(np.float32).transpose(1, 0, 2)[None, :n]
    vh = vc.view(np.float16).reshape(bs, hkv, n_ctx, hs)[0].astype(np.float32).transpose(1, 0, 2)[None, :n]
    assert nso.rel_l2(fresh, nso.attn_ref(q, kh.astype(np.float16), vh.astype(np.float16), scale, 1)) < 1e-3
    assert not np.array_equal(fresh.view(np.int32), stale.view(np.int32))
