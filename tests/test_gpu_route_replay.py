"""csrc/ns_route.cpp: record / verify / replay of the reference's per-token device graph, driven through the C ABI the way the reference's
executor drives it (ne_layers.c:11915-12028) — a token = host-to-device copy of its input, a fixed sequence of launches on the device
queue (rms_norm, mul by the norm weight, three mul_mat of one input, rope x 2, two cache writes, mul_mat + residual add, gate / silu / up /
mul, mul_mat + add), synchronise, device-to-host copy of its output.  Like the reference's device pool (ne_new_device_tensor_impl,
ne_layers.c:904-945) every activation address moves by a constant per token.  The same token stream runs with the layer on and off:
  * tokens 0, 1 are launched one by one, token 2 onwards are replayed from the plan (verified launch by launch),
  * a token that DEVIATES (its position jumps) falls back in the middle of the plan and must still compute the right thing,
  * after it two agreeing tokens make a new plan.
Every token's output and the cache contents must agree between the two runs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, FF, HEADS, HS, NCTX = 512, 1408, 4, 128, 64
DELTA = 4864  # bytes an activation moves per token (what the reference's pool leaks on the small llama test model)


def _api(L):
    vp, sz, i, ll4 = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_longlong)
    L.bestla_create_device.restype = vp
    L.bestla_create_device.argtypes = [C.c_bool]
    L.bestla_get_device_queue.restype = vp
    L.bestla_get_device_queue.argtypes = [vp]
    L.bestla_release_device.argtypes = [vp]
    L.bestla_device_malloc.restype = vp
    L.bestla_device_malloc.argtypes = [sz, vp]
    L.bestla_device_free.argtypes = [vp, vp]
    L.bestla_device_storage_size.restype = sz
    L.bestla_device_load_storage.argtypes = [vp, vp, vp, vp]
    L.bestla_device_f32f32_forward.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp]
    L.bestla_device_memcpy_sync.argtypes = [vp, vp, sz, vp]
    L.bestla_device_sync.argtypes = [vp]
    L.ns_hip_device_storage_release.argtypes = [vp]
    L.ns_hip_lazy_rms_norm.argtypes = [i, i, C.c_float, vp, vp, vp]
    L.ns_hip_lazy_silu.argtypes = [vp, vp, sz, vp]
    L.ns_hip_lazy_mul.argtypes = [vp, vp, vp, ll4, ll4, ll4, ll4, ll4, vp]
    L.ns_hip_lazy_flush.restype = i
    L.ns_hip_binary_nd_f32.argtypes = [i, vp, vp, vp, ll4, ll4, ll4, ll4, ll4, vp]
    L.ns_hip_rope_f32.argtypes = [vp, vp, i, i, i, i, i, i, i, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.ns_hip_dup_f32.argtypes = [vp, vp, ll4, ll4, ll4, C.c_bool, vp]
    L.ns_hip_route_stats.argtypes = [vp]
    L.ns_hip_route_set_enabled.argtypes = [i]


def _ll(*v):
    return (C.c_longlong * 4)(*v)


def _run(L, nso, blobs, gam, xs, positions, replay):
    """the token stream; returns (outputs per token, K cache, V cache, route statistics)"""
    _api(L)
    L.ns_hip_route_set_enabled(1 if replay else 0)
    dev = L.bestla_create_device(False)
    q = L.bestla_get_device_queue(dev)
    # weights through the reference's loader entry
    stors, slices = {}, []
    for name, blob in blobs.items():
        size = int(np.frombuffer(blob[:8].tobytes(), np.uint64)[0])
        dptr = L.bestla_device_malloc((size + 255) // 256 * 256, q)
        stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
        L.bestla_device_load_storage(nso.ptr(blob.copy()), nso.ptr(stor), dptr, q)
        stors[name] = stor
        slices.append(dptr)
    f4 = 4
    pool_bytes = 1 << 20
    pool = L.bestla_device_malloc(pool_bytes, q)
    kc = L.bestla_device_malloc(HEADS * NCTX * HS * f4, q)  # [head][n_ctx][hs]
    vc = L.bestla_device_malloc(HEADS * HS * NCTX * f4, q)  # [head][hs][n_ctx]
    zero = np.zeros(HEADS * NCTX * HS, np.float32)
    L.bestla_device_memcpy_sync(kc, nso.ptr(zero), zero.nbytes, q)
    L.bestla_device_memcpy_sync(vc, nso.ptr(zero), zero.nbytes, q)
    dg = L.bestla_device_malloc(D * f4, q)
    L.bestla_device_memcpy_sync(dg, nso.ptr(gam), gam.nbytes, q)
    outs = []
    vec = lambda n: (_ll(n, 1, 1, 1), _ll(4, 4 * n, 4 * n, 4 * n))
    for t, (x, pos) in enumerate(zip(xs, positions)):
        base = pool + t * DELTA
        off = [0]

        def alloc(nfloat):
            p = base + off[0]
            off[0] += (nfloat * f4 + 255) // 256 * 256
            return p
        px, pn, ph, pk, pv, pq = alloc(D), alloc(D), alloc(D), alloc(D), alloc(D), alloc(D)
        pt, pr, pn2, ph2 = alloc(D), alloc(D), alloc(D), alloc(D)
        pt1, ps, pt3, pp, pt2, po = alloc(FF), alloc(FF), alloc(FF), alloc(FF), alloc(D), alloc(D)
        # (allocation order as the llama graph makes it: q in front of k in front of v is NOT assumed — here k, v, q like the graph's dump)
        L.bestla_device_sync(q)
        L.bestla_device_memcpy_sync(px, nso.ptr(x), x.nbytes, q)
        ne, nb = vec(D)
        assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, px, pn, q) == 0
        assert L.ns_hip_lazy_mul(pn, dg, ph, ne, nb, ne, nb, nb, q) == 0
        L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wk"]), pk, 1, D, D, D, D, None, q)
        assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pk, pk, 1, 1, HEADS, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
        # K -> cache [head][n_ctx][hs] at position pos: dst extents (hs, 1, heads, 1)
        assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pk, kc + pos * HS * f4, _ll(HS, 1, HEADS, 1), _ll(4, HS * HEADS * 4, HS * 4, HS * HEADS * 4),
                                                                  _ll(4, HS * 4, NCTX * HS * 4, HEADS * NCTX * HS * 4), False, q) == 0
        L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wv"]), pv, 1, D, D, D, D, None, q)
        # V -> cache [head][hs][n_ctx] at position pos: dst extents (1, hs, heads, 1)
        assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pv, vc + pos * f4, _ll(1, HS, HEADS, 1), _ll(HS * HEADS * 4, 4, HS * 4, HS * HEADS * 4),
                                                                  _ll(4, NCTX * 4, HS * NCTX * 4, HEADS * HS * NCTX * 4), False, q) == 0
        L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wq"]), pq, 1, D, D, D, D, None, q)
        assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pq, pq, 1, 1, HEADS, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
        # (no attention here: the stand-in for its output is the rotated q — the launches around it are what this test is about)
        L.bestla_device_f32f32_forward(pq, nso.ptr(stors["wo"]), pt, 1, D, D, D, D, None, q)
        assert L.ns_hip_binary_nd_f32(0, pt, px, pr, ne, nb, ne, nb, nb, q) == 0
        assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, pr, pn2, q) == 0
        assert L.ns_hip_lazy_mul(pn2, dg, ph2, ne, nb, ne, nb, nb, q) == 0
        L.bestla_device_f32f32_forward(ph2, nso.ptr(stors["w1"]), pt1, 1, FF, D, D, FF, None, q)
        assert L.ns_hip_lazy_silu(pt1, ps, FF, q) == 0
        L.bestla_device_f32f32_forward(ph2, nso.ptr(stors["w3"]), pt3, 1, FF, D, D, FF, None, q)
        nef, nbf = vec(FF)
        assert L.ns_hip_lazy_mul(ps, pt3, pp, nef, nbf, nef, nbf, nbf, q) == 0
        L.bestla_device_f32f32_forward(pp, nso.ptr(stors["w2"]), pt2, 1, D, FF, FF, D, None, q)
        assert L.ns_hip_binary_nd_f32(0, pt2, pr, po, ne, nb, ne, nb, nb, q) == 0
        L.bestla_device_sync(q)
        out = np.zeros(D, np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(out), po, out.nbytes, q)
        outs.append(out)
    kcache, vcache = np.zeros(HEADS * NCTX * HS, np.float32), np.zeros(HEADS * HS * NCTX, np.float32)
    L.bestla_device_memcpy_sync(nso.ptr(kcache), kc, kcache.nbytes, q)
    L.bestla_device_memcpy_sync(nso.ptr(vcache), vc, vcache.nbytes, q)
    st = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st)
    for s in stors.values():
        L.ns_hip_device_storage_release(nso.ptr(s))
    for p in slices + [pool, kc, vc, dg]:
        L.bestla_device_free(p, q)
    L.bestla_release_device(dev)
    L.ns_hip_route_set_enabled(1)
    return outs, kcache, vcache, list(st)


def test_replayed_tokens_and_a_deviating_token_compute_what_plain_launches_compute(L, pkg, nso):
    rng = np.random.default_rng(21)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    # positions 0..7, then a JUMP (token 8 is at position 20), then on from there
    positions = list(range(8)) + [20, 21, 22, 23, 24, 25]
    xs = [rng.standard_normal(D).astype(np.float32) for _ in positions]
    base = [int(v) for v in (C.c_uint64 * 8)()]
    st0 = (C.c_uint64 * 8)()
    _api(L)
    L.ns_hip_route_stats(st0)
    ref_out, ref_k, ref_v, _ = _run(L, nso, blobs, gam, xs, positions, replay=False)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_k, got_v, st2 = _run(L, nso, blobs, gam, xs, positions, replay=True)
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    # tokens 0, 1 launched one by one -> plan; 2..7 replayed; 8 deviates (fallback).  Tokens 7 and 8 agree in everything but their moving
    # values (position +13), so they make a plan too — which token 9 (position +1) leaves at its first rope: a second fallback; 8 and 9
    # make the plan that 10..13 are replayed from
    assert (replayed, eager, plans, fallbacks) == (10, 4, 3, 2), (replayed, eager, plans, fallbacks)
    assert st2[5] < st2[4], (st2[4], st2[5])  # the plan's graphs hold fewer launches than the token has (fused QKV / gate-up / residual adds / rope + cache writes)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        # fused launches compute the same values (the residual add in the GEMV's epilogue, silu * up in registers): bit-equal or last-bit close
        assert np.allclose(a, b, rtol=2e-5, atol=2e-5), (t, float(np.abs(a - b).max()))
    assert np.array_equal(ref_k, got_k) and np.array_equal(ref_v, got_v)
    assert np.count_nonzero(got_k) > 0 and np.count_nonzero(got_v) > 0


NL = 2  # decoder layers of the second stream


def _run_layers(L, nso, blobs, gam, xs, replay, nctx=NCTX, pos0=0, cache0=None):
    """NL decoder layers WITH the attention node and the model's last norm + output projection: the shape in which the plan carries RMS norms
    across launches (ns_route.cpp link_norms).  Positions pos0, pos0 + 1, ... of caches made for nctx positions (cache0: their initial contents).  Returns (outputs per token, K caches, V caches, route statistics)."""
    _api(L)
    vp, i = C.c_void_p, C.c_int
    L.ns_hip_mha_f32_device_layout.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, C.c_float, i, vp]
    L.ns_hip_route_set_enabled(replay)  # 0 off, 3 replay with carried norms, 5 replay without
    dev = L.bestla_create_device(False)
    q = L.bestla_get_device_queue(dev)
    stors, slices = {}, []
    for name, blob in blobs.items():
        size = int(np.frombuffer(blob[:8].tobytes(), np.uint64)[0])
        dptr = L.bestla_device_malloc((size + 255) // 256 * 256, q)
        stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
        L.bestla_device_load_storage(nso.ptr(blob.copy()), nso.ptr(stor), dptr, q)
        stors[name] = stor
        slices.append(dptr)
    f4 = 4
    pool = L.bestla_device_malloc(1 << 21, q)
    zero = np.zeros(HEADS * nctx * HS, np.float32)
    kcs, vcs = [], []
    for _ in range(NL):
        kc, vc = L.bestla_device_malloc(zero.nbytes, q), L.bestla_device_malloc(zero.nbytes, q)
        L.bestla_device_memcpy_sync(kc, nso.ptr(zero if cache0 is None else cache0[len(kcs)]), zero.nbytes, q)
        L.bestla_device_memcpy_sync(vc, nso.ptr(zero if cache0 is None else cache0[NL + len(vcs)]), zero.nbytes, q)
        kcs.append(kc), vcs.append(vc)
    dg = L.bestla_device_malloc(D * f4, q)
    L.bestla_device_memcpy_sync(dg, nso.ptr(gam), gam.nbytes, q)
    outs = []
    vec = lambda n: (_ll(n, 1, 1, 1), _ll(4, 4 * n, 4 * n, 4 * n))
    ne, nb = vec(D)
    nef, nbf = vec(FF)
    for tok, x in enumerate(xs):
        pos = pos0 + tok
        base = pool + tok * DELTA
        off = [0]

        def alloc(nfloat):
            p = base + off[0]
            off[0] += (nfloat * f4 + 255) // 256 * 256
            return p
        px = alloc(D)
        L.bestla_device_sync(q)
        L.bestla_device_memcpy_sync(px, nso.ptr(x), x.nbytes, q)
        for il in range(NL):
            pn, ph, pk, pv, pq, pa = alloc(D), alloc(D), alloc(D), alloc(D), alloc(D), alloc(D)
            pt, pr, pn2, ph2 = alloc(D), alloc(D), alloc(D), alloc(D)
            pt1, ps, pt3, pp, pt2, po = alloc(FF), alloc(FF), alloc(FF), alloc(FF), alloc(D), alloc(D)
            kc, vc = kcs[il], vcs[il]
            assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, px, pn, q) == 0
            assert L.ns_hip_lazy_mul(pn, dg, ph, ne, nb, ne, nb, nb, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wk"]), pk, 1, D, D, D, D, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pk, pk, 1, 1, HEADS, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pk, kc + pos * HS * f4, _ll(HS, 1, HEADS, 1), _ll(4, HS * HEADS * 4, HS * 4, HS * HEADS * 4),
                                                                      _ll(4, HS * 4, nctx * HS * 4, HEADS * nctx * HS * 4), False, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wv"]), pv, 1, D, D, D, D, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pv, vc + pos * f4, _ll(1, HS, HEADS, 1), _ll(HS * HEADS * 4, 4, HS * 4, HS * HEADS * 4),
                                                                      _ll(4, nctx * 4, HS * nctx * 4, HEADS * HS * nctx * 4), False, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wq"]), pq, 1, D, D, D, D, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pq, pq, 1, 1, HEADS, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
            assert L.ns_hip_mha_f32_device_layout(pq, kc, vc, pa, 1, 1, pos + 1, HEADS, HEADS, HS, nctx, HS ** -0.5, 1, q) == 0
            L.bestla_device_f32f32_forward(pa, nso.ptr(stors["wo"]), pt, 1, D, D, D, D, None, q)
            assert L.ns_hip_binary_nd_f32(0, pt, px, pr, ne, nb, ne, nb, nb, q) == 0
            assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, pr, pn2, q) == 0
            assert L.ns_hip_lazy_mul(pn2, dg, ph2, ne, nb, ne, nb, nb, q) == 0
            L.bestla_device_f32f32_forward(ph2, nso.ptr(stors["w1"]), pt1, 1, FF, D, D, FF, None, q)
            assert L.ns_hip_lazy_silu(pt1, ps, FF, q) == 0
            L.bestla_device_f32f32_forward(ph2, nso.ptr(stors["w3"]), pt3, 1, FF, D, D, FF, None, q)
            assert L.ns_hip_lazy_mul(ps, pt3, pp, nef, nbf, nef, nbf, nbf, q) == 0
            L.bestla_device_f32f32_forward(pp, nso.ptr(stors["w2"]), pt2, 1, D, FF, FF, D, None, q)
            assert L.ns_hip_binary_nd_f32(0, pt2, pr, po, ne, nb, ne, nb, nb, q) == 0
            px = po
        pfn, pfh, plog = alloc(D), alloc(D), alloc(FF)
        assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, px, pfn, q) == 0
        assert L.ns_hip_lazy_mul(pfn, dg, pfh, ne, nb, ne, nb, nb, q) == 0
        L.bestla_device_f32f32_forward(pfh, nso.ptr(stors["w1"]), plog, 1, FF, D, D, FF, None, q)  # (the gate matrix stands in for an output projection)
        L.bestla_device_sync(q)
        out = np.zeros(FF, np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(out), plog, out.nbytes, q)
        outs.append(out)
    caches = []
    for c in kcs + vcs:
        h = np.zeros(HEADS * nctx * HS, np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(h), c, h.nbytes, q)
        caches.append(h)
    st = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st)
    for s2 in stors.values():
        L.ns_hip_device_storage_release(nso.ptr(s2))
    for p2 in slices + [pool, dg] + kcs + vcs:
        L.bestla_device_free(p2, q)
    L.bestla_release_device(dev)
    L.ns_hip_route_set_enabled(1)
    return outs, caches, list(st)


def test_plan_carries_the_rms_norms_across_launches(L, pkg, nso):
    """Two decoder layers with attention + last norm + output projection: the replayed plan launches neither rms_norm nor mul(gamma) where the
    normed tensor came out of a residual add (4 of the 5 norms of the token: the first one norms the input embedding) — the producer writes
    fp16(gamma . x) and the tile sums of squares, the consumer divides by rms(x).  Same tokens as plain launches within fp16 rounding of the
    activations (either way they are rounded to fp16 once: after the norm, or before it)."""
    rng = np.random.default_rng(5)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(10)]
    _api(L)
    ref_out, ref_c, _ = _run_layers(L, nso, blobs, gam, xs, replay=0)
    st0 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st0)
    plain_out, plain_c, stp = _run_layers(L, nso, blobs, gam, xs, replay=5)
    assert stp[5] == 27, stp[5]  # replay without carried norms: every norm is a launch
    for a, b in zip(ref_out, plain_out):
        assert nso.rel_l2(b, a) < 1e-3  # (fused launches + the context-split attention: a different summation order through two layers)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_c, st2 = _run_layers(L, nso, blobs, gam, xs, replay=3)
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    assert (replayed, eager, plans, fallbacks) == (8, 2, 1, 0), (replayed, eager, plans, fallbacks)
    # 43 launches of the reference per token; fused without carried norms 27 (per layer: norm, mul, QKV, rope(k), rope(q) — this stream's q rows do
    # not follow its k rows —, both cache writes, attention, WO + add, norm, mul, gate/up, down + add; then norm, mul, projection); the four
    # carried norms take eight of them away
    assert st2[4] == 43 and st2[5] == 19, (st2[4], st2[5])
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        assert np.all(np.isfinite(b))
        assert nso.rel_l2(b, a) < 2e-3, (t, nso.rel_l2(b, a))
    for a, b in zip(ref_c, got_c):
        assert np.count_nonzero(b) > 0 and nso.rel_l2(b, a) < 2e-3


def test_replayed_attention_over_several_context_ranges(L, pkg, nso):
    """Positions 250 .. 261 of caches made for 512: the replayed attention covers 2 and then 3 live ranges of 128 keys (of the 4 its grid is
    made for) and its last range to finish merges them inside the launch (ns_device.hip, tickets) — no merge launch in the plan."""
    rng = np.random.default_rng(6)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(12)]
    nctx, pos0 = 512, 250
    cache0 = [(0.5 * rng.standard_normal(HEADS * nctx * HS)).astype(np.float32) for _ in range(2 * NL)]
    _api(L)
    ref_out, ref_c, _ = _run_layers(L, nso, blobs, gam, xs, 0, nctx, pos0, cache0)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_c, st2 = _run_layers(L, nso, blobs, gam, xs, 3, nctx, pos0, cache0)
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    assert (replayed, eager, plans, fallbacks) == (10, 2, 1, 0), (replayed, eager, plans, fallbacks)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        assert np.all(np.isfinite(b))
        assert nso.rel_l2(b, a) < 2e-3, (t, nso.rel_l2(b, a))
    for a, b in zip(ref_c, got_c):
        assert nso.rel_l2(b, a) < 2e-3
