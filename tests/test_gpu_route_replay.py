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


class _Stream:
    """one device context (bestla_create_device) with its weights, activation pool and kv cache; token() issues one token's launches the way the
    reference's executor does.  `late_extra`: the token gets one more launch at its end (an add) — it deviates from a plan AFTER every segment of
    the plan has been launched."""

    def __init__(self, L, nso, blobs, gam):
        self.L, self.nso = L, nso
        self.dev = L.bestla_create_device(False)
        q = self.q = L.bestla_get_device_queue(self.dev)
        self.stors, self.slices = {}, []
        for name, blob in blobs.items():
            size = int(np.frombuffer(blob[:8].tobytes(), np.uint64)[0])
            dptr = L.bestla_device_malloc((size + 255) // 256 * 256, q)
            stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
            L.bestla_device_load_storage(nso.ptr(blob.copy()), nso.ptr(stor), dptr, q)
            self.stors[name] = stor
            self.slices.append(dptr)
        self.pool = L.bestla_device_malloc(1 << 20, q)
        self.kc = L.bestla_device_malloc(HEADS * NCTX * HS * 4, q)  # [head][n_ctx][hs]
        self.vc = L.bestla_device_malloc(HEADS * HS * NCTX * 4, q)  # [head][hs][n_ctx]
        zero = np.zeros(HEADS * NCTX * HS, np.float32)
        L.bestla_device_memcpy_sync(self.kc, nso.ptr(zero), zero.nbytes, q)
        L.bestla_device_memcpy_sync(self.vc, nso.ptr(zero), zero.nbytes, q)
        self.dg = L.bestla_device_malloc(D * 4, q)
        L.bestla_device_memcpy_sync(self.dg, nso.ptr(gam), gam.nbytes, q)
        self.t = 0

    def token(self, x, pos, late_extra=False):
        L, nso, q, stors, kc, vc, dg, f4 = self.L, self.nso, self.q, self.stors, self.kc, self.vc, self.dg, 4
        vec = lambda n: (_ll(n, 1, 1, 1), _ll(4, 4 * n, 4 * n, 4 * n))
        base = self.pool + self.t * DELTA
        self.t += 1
        off = [0]

        def alloc(nfloat):
            p = base + off[0]
            off[0] += (nfloat * f4 + 255) // 256 * 256
            return p
        px, pn, ph, pk, pv, pq = alloc(D), alloc(D), alloc(D), alloc(D), alloc(D), alloc(D)
        pt, pr, pn2, ph2 = alloc(D), alloc(D), alloc(D), alloc(D)
        pt1, ps, pt3, pp, pt2, po = alloc(FF), alloc(FF), alloc(FF), alloc(FF), alloc(D), alloc(D)
        po2 = alloc(D)
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
        if late_extra:   # out = (down + residual) + input embedding: reads the token's INPUT again after everything else ran
            assert L.ns_hip_binary_nd_f32(0, po, px, po2, ne, nb, ne, nb, nb, q) == 0
            po = po2
        L.bestla_device_sync(q)
        out = np.zeros(D, np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(out), po, out.nbytes, q)
        return out

    def finish(self):
        L, nso, q = self.L, self.nso, self.q
        kcache, vcache = np.zeros(HEADS * NCTX * HS, np.float32), np.zeros(HEADS * HS * NCTX, np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(kcache), self.kc, kcache.nbytes, q)
        L.bestla_device_memcpy_sync(nso.ptr(vcache), self.vc, vcache.nbytes, q)
        for s in self.stors.values():
            L.ns_hip_device_storage_release(nso.ptr(s))
        for p in self.slices + [self.pool, self.kc, self.vc, self.dg]:
            L.bestla_device_free(p, q)
        L.bestla_release_device(self.dev)
        return kcache, vcache


def _run(L, nso, blobs, gam, xs, positions, replay, late=()):
    """the token stream; returns (outputs per token, K cache, V cache, route statistics)"""
    _api(L)
    L.ns_hip_route_set_enabled(1 if replay else 0)
    sm = _Stream(L, nso, blobs, gam)
    outs = [sm.token(x, pos, late_extra=(t in late)) for t, (x, pos) in enumerate(zip(xs, positions))]
    st = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st)
    kcache, vcache = sm.finish()
    L.ns_hip_route_set_enabled(1)
    return outs, kcache, vcache, list(st)


def test_replayed_tokens_and_a_deviating_token_compute_what_plain_launches_compute(L, pkg, nso):
    rng = np.random.default_rng(21)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    # positions 0..7, then a JUMP (token 8 is at position 20), then on from there
    positions = list(range(8)) + list(range(20, 32))
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
    # tokens 0, 1 go out through the window -> plan; 2..7 replayed; 8 deviates (fallback).  Tokens 7 and 8 agree in everything but their moving
    # values (position +13), so they make a plan too — which token 9 (position +1) leaves at its first rope: a second fallback IN A ROW, after
    # which the next plan waits for four agreeing tokens (round 6: no capture every other token for interleaved sequences): 9 .. 13 go through the
    # window, 12 and 13 make the plan that 14 .. 19 are replayed from
    assert (replayed, eager, plans, fallbacks) == (12, 8, 3, 2), (replayed, eager, plans, fallbacks)
    assert st2[5] < st2[4], (st2[4], st2[5])  # the plan's graphs hold fewer launches than the token has (fused QKV / gate-up / residual adds / rope + cache writes)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        # fused launches compute the same values (the residual add in the GEMV's epilogue, silu * up in registers): bit-equal or last-bit close
        assert np.allclose(a, b, rtol=2e-5, atol=2e-5), (t, float(np.abs(a - b).max()))
    assert np.array_equal(ref_k, got_k) and np.array_equal(ref_v, got_v)
    assert np.count_nonzero(got_k) > 0 and np.count_nonzero(got_v) > 0


def _blobs(nso, rng):
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    return {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}


def test_a_token_that_deviates_after_every_segment_ran_is_issued_again_from_its_kept_input(L, pkg, nso):
    """a replayed token runs on the PLAN's activations, and its own tensors sit only DELTA * k bytes above them — inside memory the
    plan's segments write.  Token 6 is the plan's launches plus one more at its end that reads the token's INPUT again: every segment has been
    launched when it deviates, the input it asked for has been written over, and the token must still compute what plain launches compute (the
    route keeps a device-side copy of every evaluation's input and puts it back before it issues the token again)."""
    rng = np.random.default_rng(33)
    blobs = _blobs(nso, rng)
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    positions = list(range(12))
    xs = [rng.standard_normal(D).astype(np.float32) for _ in positions]
    _api(L)
    ref_out, ref_k, ref_v, _ = _run(L, nso, blobs, gam, xs, positions, replay=False, late=(6,))
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_k, got_v, st2 = _run(L, nso, blobs, gam, xs, positions, replay=True, late=(6,))
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    # 0, 1 -> plan; 2 .. 5 replayed; 6 falls back at its extra launch; 7 deviates from nothing (no plan held); 7, 8 make the next plan; 9 .. 11 replayed
    assert (replayed, eager, plans, fallbacks) == (7, 5, 2, 1), (replayed, eager, plans, fallbacks)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        assert np.allclose(a, b, rtol=2e-5, atol=2e-5), (t, float(np.abs(a - b).max()))
    assert np.array_equal(ref_k, got_k) and np.array_equal(ref_v, got_v)


def test_two_device_contexts_alternating_tokens_keep_separate_plans(L, pkg, nso):
    """round 5 had ONE process-global route bound to the first queue — a second model (a draft + a target model, two servers in one
    process) ran unrecorded or thrashed the first one's plan.  Round 6: one route per device queue.  Two contexts with different weights generate
    alternately, token by token; each must replay from its own plan and compute what it computes alone with the layer off."""
    rng = np.random.default_rng(34)
    blobs_a, blobs_b = _blobs(nso, rng), _blobs(nso, rng)
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    n = 10
    xa = [rng.standard_normal(D).astype(np.float32) for _ in range(n)]
    xb = [rng.standard_normal(D).astype(np.float32) for _ in range(n)]
    _api(L)
    ref_a = _run(L, nso, blobs_a, gam, xa, list(range(n)), replay=False)
    ref_b = _run(L, nso, blobs_b, gam, xb, list(range(5, 5 + n)), replay=False)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    L.ns_hip_route_set_enabled(1)
    sa, sb = _Stream(L, nso, blobs_a, gam), _Stream(L, nso, blobs_b, gam)
    out_a, out_b = [], []
    for t in range(n):
        out_a.append(sa.token(xa[t], t))
        out_b.append(sb.token(xb[t], 5 + t))
    st2 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st2)
    ka, va = sa.finish()
    kb, vb = sb.finish()
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    assert (replayed, eager, plans, fallbacks) == (2 * (n - 2), 4, 2, 0), (replayed, eager, plans, fallbacks)
    for ref, outs, kk, vv in ((ref_a, out_a, ka, va), (ref_b, out_b, kb, vb)):
        for t, (a, b) in enumerate(zip(ref[0], outs)):
            assert np.allclose(a, b, rtol=2e-5, atol=2e-5), (t, float(np.abs(a - b).max()))
        assert np.array_equal(ref[1], kk) and np.array_equal(ref[2], vv)


NL = 2  # decoder layers of the second stream


def _run_layers(L, nso, blobs, gam, xs, replay, nctx=NCTX, pos0=0, cache0=None, hkv=HEADS, poke=None):
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
    zero = np.zeros(hkv * nctx * HS, np.float32)
    DKV = hkv * HS
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
        if poke is not None and tok == poke[0]:   # a copy INTO the kv caches between two tokens (a restored session): (token index, fp32 rows for position poke[1])
            for il in range(NL):
                for h_ in range(hkv):
                    row = np.ascontiguousarray(poke[2][il][h_])
                    L.bestla_device_memcpy_sync(kcs[il] + (h_ * nctx + poke[1]) * HS * f4, nso.ptr(row), row.nbytes, q)
        L.bestla_device_memcpy_sync(px, nso.ptr(x), x.nbytes, q)
        for il in range(NL):
            pn, ph, pk, pv, pq, pa = alloc(D), alloc(D), alloc(D), alloc(D), alloc(D), alloc(D)
            pt, pr, pn2, ph2 = alloc(D), alloc(D), alloc(D), alloc(D)
            pt1, ps, pt3, pp, pt2, po = alloc(FF), alloc(FF), alloc(FF), alloc(FF), alloc(D), alloc(D)
            kc, vc = kcs[il], vcs[il]
            assert L.ns_hip_lazy_rms_norm(1, D, 1e-5, px, pn, q) == 0
            assert L.ns_hip_lazy_mul(pn, dg, ph, ne, nb, ne, nb, nb, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wk"]), pk, 1, DKV, D, D, DKV, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pk, pk, 1, 1, hkv, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pk, kc + pos * HS * f4, _ll(HS, 1, hkv, 1), _ll(4, HS * hkv * 4, HS * 4, HS * hkv * 4),
                                                                      _ll(4, HS * 4, nctx * HS * 4, hkv * nctx * HS * 4), False, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wv"]), pv, 1, DKV, D, D, DKV, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_dup_f32(pv, vc + pos * f4, _ll(1, HS, hkv, 1), _ll(HS * hkv * 4, 4, HS * 4, HS * hkv * 4),
                                                                      _ll(4, nctx * 4, HS * nctx * 4, hkv * HS * nctx * 4), False, q) == 0
            L.bestla_device_f32f32_forward(ph, nso.ptr(stors["wq"]), pq, 1, D, D, D, D, None, q)
            assert L.ns_hip_lazy_flush() == 0 and L.ns_hip_rope_f32(pq, pq, 1, 1, HEADS, HS, pos, HS, 0, 10000.0, 1.0, 0.0, 1.0, q) == 0
            assert L.ns_hip_mha_f32_device_layout(pq, kc, vc, pa, 1, 1, pos + 1, HEADS, hkv, HS, nctx, HS ** -0.5, 1, q) == 0
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
        h = np.zeros(hkv * nctx * HS, np.float32)
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


@pytest.mark.parametrize("nctx,pos0", [(512, 250), (2048, 122), (2048, 700), (2048, 2030)])
def test_replayed_attention_over_several_context_ranges(L, pkg, nso, nctx, pos0):
    """Twelve positions from pos0 on of caches made for nctx: the replayed attention's grid and partials are laid out for nctx, its ranges follow the LIVE
    length (AttnSplitParams::dyn_*: every workgroup applies the host's range rule to the length it reads from the device counter — a few ranges of a
    few dozen keys at 122 .. 133 positions, crossing the one-range threshold at 128; more at 700; the layout's own at 2030 ..), and the last range to
    finish merges them inside the launch (ns_device.hip, tickets) — no merge launch in the plan."""
    rng = np.random.default_rng(6)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(D, D), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(12)]
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


def test_values_beyond_fp16_turn_the_fp16_shortcuts_off_and_the_token_is_evaluated_again(L, pkg, nso, capfd):
    """the route's fp16 shortcuts — the kv mirror its attention reads, a carried norm's shadow — cannot hold |x| > 65504, which
    the fp32 kernels of the reference's device path can.  A key projection scaled so that K is ~1e5: the converting kernel raises the flag, the route
    reads it behind the token's synchronisation, says so once on stderr, turns both shortcuts off for the process and evaluates the token again on
    the fp32 forms BEFORE its result is copied out — every token then equals the run that had the mirror off from the start (NS_DEVICE_KV=f32)."""
    rng = np.random.default_rng(8)
    mk = lambda n, k, s=1.0: nso.quant_pack((rng.standard_normal((n, k)) * s * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D, 1e-4), "wk": mk(D, D, 1e5), "wv": mk(D, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(8)]
    _api(L)
    L.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
    try:
        assert L.ns_hip_set_tuning(b"device_kv_f16", 0) == 0
        ref_out, ref_c, _ = _run_layers(L, nso, blobs, gam, xs, replay=5)   # fp32 kernels, plans without carried norms
        capfd.readouterr()
        assert L.ns_hip_set_tuning(b"device_kv_f16", 1) == 0
        got_out, got_c, _ = _run_layers(L, nso, blobs, gam, xs, replay=3)   # mirror + carried norms: overflows at the first token
        err = capfd.readouterr().err
        assert err.count("beyond the fp16 range") == 1, err[-1500:]
        for t, (a, b) in enumerate(zip(ref_out, got_out)):
            assert np.all(np.isfinite(b)), t
            assert nso.rel_l2(b, a) < 2e-3, (t, nso.rel_l2(b, a))
        for a, b in zip(ref_c, got_c):
            assert np.array_equal(a, b) or nso.rel_l2(b, a) < 2e-3
    finally:
        L.ns_hip_set_tuning(b"device_kv_f16", -1)
        L.ns_hip_route_set_enabled(1)


@pytest.mark.parametrize("nctx,pos0", [(512, 120), (4096, 1500)])
def test_replayed_grouped_query_attention_on_the_kv_mirror(L, pkg, nso, nctx, pos0):
    """Two kv heads under four query heads (the shape of Mistral / Llama-2-70B layers): the three projections have different widths, so the plan keeps them as
    separate launches (the reference's own graph does, llama.cpp:215) with rope(k), rope(q) and the two cache writes as launches of their own — each cache
    write storing into the fp16 mirror as well — and the replayed attention serves two query heads per kv head of a mirror that grows across a context-range boundary (4 -> 5 live
    ranges of 32 keys; second case: 1500 .. 1513 positions of caches made for 4096 — the launch's layout has 64 ranges of 64 keys per kv head, the live
    length's rule a couple of dozen).  Same tokens as plain launches."""
    rng = np.random.default_rng(9)
    hkv = 2
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    blobs = {"wq": mk(D, D), "wk": mk(hkv * HS, D), "wv": mk(hkv * HS, D), "wo": mk(D, D), "w1": mk(FF, D), "w3": mk(FF, D), "w2": mk(D, FF)}
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(14)]
    cache0 = [(0.5 * rng.standard_normal(hkv * nctx * HS)).astype(np.float32) for _ in range(2 * NL)]
    _api(L)
    ref_out, ref_c, _ = _run_layers(L, nso, blobs, gam, xs, 0, nctx, pos0, cache0, hkv=hkv)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_c, st2 = _run_layers(L, nso, blobs, gam, xs, 3, nctx, pos0, cache0, hkv=hkv)
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    assert (replayed, eager, plans, fallbacks) == (12, 2, 1, 0), (replayed, eager, plans, fallbacks)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        assert np.all(np.isfinite(b))
        assert nso.rel_l2(b, a) < 2e-3, (t, nso.rel_l2(b, a))
    for a, b in zip(ref_c, got_c):
        assert nso.rel_l2(b, a) < 2e-3


def test_a_copy_into_the_kv_cache_between_two_replayed_tokens_reaches_the_attention(L, pkg, nso):
    """The route's attention reads an fp16 MIRROR of the fp32 cache, kept current by the plan's own cache writes.  A copy into the cache from outside the graph
    (a restored session, a beam's rows) between two replayed tokens must reach it: the copy drops the plan and empties the mirror, the next token goes through
    the window and its attention converts the cache afresh.  K rows of position 2 are overwritten in front of token 6; every token as with plain launches."""
    rng = np.random.default_rng(10)
    blobs = _blobs(nso, rng)
    gam = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    xs = [rng.standard_normal(D).astype(np.float32) for _ in range(12)]
    poke = (6, 2, [[(3.0 * rng.standard_normal(HS)).astype(np.float32) for _ in range(HEADS)] for _ in range(NL)])
    _api(L)
    ref_out, ref_c, _ = _run_layers(L, nso, blobs, gam, xs, 0, poke=poke)
    st1 = (C.c_uint64 * 8)()
    L.ns_hip_route_stats(st1)
    got_out, got_c, st2 = _run_layers(L, nso, blobs, gam, xs, 3, poke=poke)
    replayed, eager, plans, fallbacks = (st2[i] - st1[i] for i in range(4))
    # 0, 1 -> plan; 2 .. 5 replayed; the copy drops the plan; 6, 7 through the window -> plan; 8 .. 11 replayed
    # (NS_DEVICE_KV=f32: no mirror — the plan's attention reads the fp32 cache the copy wrote, nothing to drop)
    import os
    f32 = os.environ.get("NS_DEVICE_KV", "") in ("f32", "fp32", "0")
    assert (replayed, eager, plans) == ((10, 2, 1) if f32 else (8, 4, 2)), (replayed, eager, plans, fallbacks)
    for t, (a, b) in enumerate(zip(ref_out, got_out)):
        assert np.all(np.isfinite(b))
        assert nso.rel_l2(b, a) < 2e-3, (t, nso.rel_l2(b, a))
    # the poke changes what the tokens behind it compute (the test would pass on a stale mirror otherwise)
    plain_out, _, _ = _run_layers(L, nso, blobs, gam, xs, 3)
    assert nso.rel_l2(plain_out[7], got_out[7]) > 1e-2
    for a, b in zip(ref_c, got_c):
        assert nso.rel_l2(b, a) < 2e-3


def test_a_large_result_copy_behind_a_deferred_sync_into_untouched_host_pages(L, nso):
    """An evaluation ends with bestla_device_sync, copy, sync (ne_layers.c:8345-8346).  The first of the two waits is left to the copy when only launches
    are pending (route_defer_sync), so the copy of a prompt's logits arrives while the queue still works and its destination — pages the caller has not
    touched yet — is faulted in meanwhile (csrc/ns_device.hip).  48 MB of results behind a long chain of launches: every byte must arrive, the pages in
    front of and behind the destination must stay as they were, with the default switches and with NS_ROUTE_LAZY_SYNC / NS_DEVICE_PRETOUCH semantics
    (a copy into pages that already exist takes the same path minus the faults)."""
    import mmap
    _api(L)
    L.ns_hip_route_set_enabled(1)
    dev = L.bestla_create_device(False)
    q = L.bestla_get_device_queue(dev)
    n = 12 << 20  # floats: 48 MB
    src, tmp = L.bestla_device_malloc(n * 4, q), L.bestla_device_malloc(n * 4, q)
    host = np.arange(n, dtype=np.float32) % 1009.0
    L.bestla_device_memcpy_sync(src, nso.ptr(host), host.nbytes, q)
    ne, nb = _ll(n, 1, 1, 1), _ll(4, 4 * n, 4 * n, 4 * n)
    for rep in range(2):   # second pass: the same destination again, its pages present
        if rep == 0:
            mm = mmap.mmap(-1, n * 4 + 3 * 4096)  # fresh anonymous pages, never touched
            dst = np.frombuffer(mm, np.uint8)
            guard_lo, guard_hi = dst[:4096 + 64], dst[4096 + 64 + n * 4:]
            guard_lo[:] = 0xA5
            guard_hi[:] = 0x5A
            out = dst[4096 + 64:4096 + 64 + n * 4].view(np.float32)   # (not page-aligned on purpose)
        # a chain of launches that keeps the queue busy for a while: x = x + x, 40 times over 48 MB, then back to the source's values by x * 2^-40
        L.bestla_device_sync(q)
        assert L.ns_hip_binary_nd_f32(0, src, src, tmp, ne, nb, ne, nb, nb, q) == 0
        for _ in range(39):
            assert L.ns_hip_binary_nd_f32(0, tmp, tmp, tmp, ne, nb, ne, nb, nb, q) == 0
        L.bestla_device_sync(q)                                        # only launches pending: deferred
        L.bestla_device_memcpy_sync(nso.ptr(out), tmp, n * 4, q)       # arrives with the queue at work
        assert np.array_equal(out, host * np.float32(2.0 ** 40))
        assert np.all(guard_lo == 0xA5) and np.all(guard_hi == 0x5A)
    L.bestla_device_free(src, q)
    L.bestla_device_free(tmp, q)
    L.bestla_release_device(dev)
    del out, guard_lo, guard_hi, dst
    mm.close()
