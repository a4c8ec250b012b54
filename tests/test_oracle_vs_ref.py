"""Pins the oracle restatement (oracle/ns_oracle.cpp) to the REAL reference scalar kernels
(bestla/bestla/kernel_ref.h via oracle/_ref/libkernel_ref.so).  Bit-exact everywhere.
Runs wherever oracle/_ref was built (this container builds it from /root/reference; the prebuilt .so
travels to the GPU box)."""
import ctypes as C

import numpy as np
import pytest

BITS = [1, 2, 3, 4, 5, 6, 7, 8]


def _weights(rng, k, n, kind):
    if kind == "normal":
        return (rng.standard_normal((k, n)) * 0.02).astype(np.float32)
    if kind == "uniform":  # the reference UT distribution, ut/bestla_ut.h:129-134
        return rng.uniform(-0.5, 0.5, (k, n)).astype(np.float32)
    w = (rng.standard_normal((k, n)) * 0.02).astype(np.float32)
    # adversarial blocks: all-zero group, dominant-positive group, single outlier, max == -min tie
    w[0:32, 0] = 0
    w[0:32, 1] = np.abs(w[0:32, 1])
    w[0, 1] = 1.0
    w[1, 1] = -0.7
    w[5, 2] = 3.0
    w[0:32, 3] = 0.01
    w[0, 3] = 0.5
    w[1, 3] = -0.5
    w[0:32, 4] = -np.abs(w[0:32, 4])
    return w


@pytest.mark.parametrize("bits", BITS)
@pytest.mark.parametrize("asym", [False, True])
@pytest.mark.parametrize("kind", ["normal", "uniform", "adversarial"])
def test_quantize_int(nso, refk, bits, asym, kind):
    rng = np.random.default_rng(100 + bits)
    k, n, bs = 160, 24, 32
    w = _weights(rng, k, n, kind)
    qt = nso.INT_TYPES[bits]
    q, sc, zp = nso.quantize(w, bs, qt, asym)
    q2 = np.zeros_like(q)
    sc2 = np.zeros_like(sc)
    zp2 = np.zeros_like(sc, dtype=np.int8) if asym else None
    refk.ref_quantize_int(nso.ptr(w), nso.ptr(q2), k, n, n, n, nso.ptr(sc2), nso.ptr(zp2), bs, C.c_uint32(qt))
    assert np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    if asym:
        assert np.array_equal(zp, zp2)
    assert np.array_equal(q, q2)


def test_quantize_int_tail_block(nso, refk):
    rng = np.random.default_rng(7)
    k, n, bs = 100, 16, 32  # last block has 4 rows (kernel_ref.h:1715-1716)
    w = _weights(rng, k, n, "normal")
    for asym in (False, True):
        q, sc, zp = nso.quantize(w, bs, nso.S4, asym)
        q2 = np.zeros_like(q)
        sc2 = np.zeros_like(sc)
        zp2 = np.zeros_like(sc, dtype=np.int8) if asym else None
        refk.ref_quantize_int(nso.ptr(w), nso.ptr(q2), k, n, n, n, nso.ptr(sc2), nso.ptr(zp2), bs, C.c_uint32(nso.S4))
        assert np.array_equal(q, q2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))


@pytest.mark.parametrize("f4", ["F4_NF4", "F4_BNB", "F4_E2M1"])
def test_quantize_f4(nso, refk, f4):
    qt = getattr(nso, f4)
    rng = np.random.default_rng(11)
    k, n, bs = 128, 16, 32
    for kind in ("normal", "uniform", "adversarial"):
        w = _weights(rng, k, n, kind)
        q, sc, _ = nso.quantize(w, bs, qt)
        q2 = np.zeros_like(q)
        sc2 = np.zeros_like(sc)
        assert refk.ref_quantize_f4(nso.ptr(w), nso.ptr(q2), k, n, n, n, nso.ptr(sc2), bs, C.c_uint32(qt)) == 0
        assert np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
        assert np.array_equal(q, q2)
    # scalar tree vs flattened thresholds, incl. NaN / inf / signed zero
    xs = np.concatenate([np.linspace(-1.2, 1.2, 4001, dtype=np.float32),
                         np.array([np.nan, np.inf, -np.inf, 0.0, -0.0], np.float32)])
    for x in xs:
        assert nso.lib().nso_f4_quantize(C.c_uint32(qt), C.c_float(x)) == refk.ref_f4_quantize(C.c_uint32(qt), C.c_float(x))
    for code in range(16):
        a = np.float32(nso.lib().nso_f4_unpack(C.c_uint32(qt), code))
        b = np.float32(refk.ref_f4_unpack(C.c_uint32(qt), code))
        c = np.float32(refk.ref_lut(C.c_uint32(qt), code))
        assert a.view(np.uint32) == b.view(np.uint32)
        assert a == c


def test_f4_thresholds_exact(nso, refk):
    """feed every threshold value and its float neighbours (the trees use strict '>')."""
    import re
    src = open(nso.HERE + "/ns_oracle.cpp").read()
    for name, qt in (("kThrNF4", nso.F4_NF4), ("kThrBNB", nso.F4_BNB)):
        body = re.search(name + r"\[\d+\] = \{([^}]*)\}", src).group(1)
        vals = [np.float32(float(t.strip().rstrip("f"))) for t in body.split(",") if t.strip()]
        assert len(vals) in (7, 15)
        for v in vals:
            for x in (np.nextafter(v, np.float32(-9)), v, np.nextafter(v, np.float32(9))):
                for s in (x, -x):
                    assert nso.lib().nso_f4_quantize(C.c_uint32(qt), C.c_float(s)) == \
                        refk.ref_f4_quantize(C.c_uint32(qt), C.c_float(s))
    for num in (0.03125, 0.53125, 1.25, 1.75, 2.5, 3.5, 5.0):
        v = np.float32(num) / np.float32(6)
        for x in (np.nextafter(v, np.float32(-9)), v, np.nextafter(v, np.float32(9))):
            for s in (x, -x):
                assert nso.lib().nso_f4_quantize(C.c_uint32(nso.F4_E2M1), C.c_float(s)) == \
                    refk.ref_f4_quantize(C.c_uint32(nso.F4_E2M1), C.c_float(s))


@pytest.mark.parametrize("ntile,packrow", [(48, 1), (48, 2), (48, 4), (24, 1), (24, 4)])
def test_padding_interleave(nso, refk, ntile, packrow):
    rng = np.random.default_rng(3)
    k, n = 70, 50
    kpad = (k + packrow - 1) // packrow * packrow
    npad = (n + ntile - 1) // ntile * ntile
    src = rng.integers(-128, 128, (k, n), dtype=np.int8)
    a = np.full(kpad * npad, 77, np.int8)
    b = np.full(kpad * npad, 77, np.int8)
    nso.lib().nso_padding_interleave(nso.ptr(src), nso.ptr(a), k, n, kpad, npad, n, kpad, ntile, packrow)
    refk.ref_padding_interleave(nso.ptr(src), nso.ptr(b), k, n, kpad, npad, n, kpad, ntile, packrow)
    assert np.array_equal(a, b)


def _ref_compress(nso, refk, codes, bits):
    n = codes.size
    out = np.zeros(nso.lib().nso_qbytes(C.c_size_t(n), C.c_uint32(nso.INT_TYPES[bits])), np.uint8)
    p = out.ctypes.data
    vp = C.c_void_p
    sz = C.c_size_t(n)
    # plane offsets: bestla_prologue_b.h:512-547
    if bits == 4:
        refk.ref_compress_s8_s4(nso.ptr(codes), vp(p), sz)
    elif bits == 7:
        refk.ref_compress_7bit(nso.ptr(codes), vp(p), vp(p + n // 2), vp(p + n // 2 + n // 4), sz)
    elif bits == 6:
        refk.ref_compress_6bit(nso.ptr(codes), vp(p), vp(p + n // 2), sz)
    elif bits == 5:
        refk.ref_compress_5bit(nso.ptr(codes), vp(p), vp(p + n // 2), sz)
    elif bits == 3:
        refk.ref_compress_3bit(nso.ptr(codes), vp(p), vp(p + n // 4), sz)
    elif bits == 2:
        refk.ref_compress_2bit(nso.ptr(codes), vp(p), sz)
    elif bits == 1:
        refk.ref_compress_1bit(nso.ptr(codes), vp(p), sz)
    return out


def _ref_decompress(nso, refk, packed, n, bits):
    out = np.zeros(n, np.int8)
    p = packed.ctypes.data
    vp = C.c_void_p
    sz = C.c_size_t(n)
    if bits == 4:
        refk.ref_decompress_s4_s8(vp(p), nso.ptr(out), sz)
    elif bits == 7:
        refk.ref_decompress_s7_s8(vp(p), vp(p + n // 2), vp(p + n // 2 + n // 4), nso.ptr(out), sz)
    elif bits == 6:
        refk.ref_decompress_s6_s8(vp(p), vp(p + n // 2), nso.ptr(out), sz)
    elif bits == 5:
        refk.ref_decompress_s5_s8(vp(p), vp(p + n // 2), nso.ptr(out), sz)
    elif bits == 3:
        refk.ref_decompress_s3_s8(vp(p), vp(p + n // 4), nso.ptr(out), sz)
    elif bits == 2:
        refk.ref_decompress_s2_s8(vp(p), nso.ptr(out), sz)
    elif bits == 1:
        refk.ref_decompress_s1_s8(vp(p), nso.ptr(out), sz)
    return out


@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 6, 7])
def test_compress_decompress(nso, refk, bits):
    rng = np.random.default_rng(bits)
    n = 48 * 64
    full = 1 << (bits - 1)
    codes = rng.integers(-full, full, n, dtype=np.int8)
    mine = nso.compress(codes, nso.INT_TYPES[bits])
    theirs = _ref_compress(nso, refk, codes, bits)
    assert np.array_equal(mine, theirs)
    d_mine = nso.decompress(mine, n, nso.INT_TYPES[bits])
    d_theirs = _ref_decompress(nso, refk, theirs, n, bits)
    assert np.array_equal(d_mine, d_theirs)
    if bits == 1:
        # compress_1bit reads src[j + FullRange] = src[j+1] for the 5th element of every 8 (kernel_ref.h:355)
        exp = codes.copy().reshape(-1, 8)
        exp[:, 4] = exp[:, 1]
        assert np.array_equal(d_mine, exp.ravel())
    else:
        assert np.array_equal(d_mine, codes)


def test_compress_f4(nso, refk):
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 16, 48 * 32, dtype=np.int8)
    mine = nso.compress(codes, nso.F4_NF4)
    theirs = np.zeros_like(mine)
    refk.ref_compress_f4(nso.ptr(codes), nso.ptr(theirs), C.c_size_t(codes.size))
    assert np.array_equal(mine, theirs)
    assert np.array_equal(nso.decompress(mine, codes.size, nso.F4_NF4), codes)


def test_scalar_casts(nso, refk):
    rng = np.random.default_rng(9)
    mant = rng.standard_normal(4000).astype(np.float32)
    expo = np.float32(10) ** rng.integers(-8, 6, 4000).astype(np.float32)
    vals = np.concatenate([mant * expo,
                           np.array([0, -0.0, 1, -1, 65504, 65520, 1e-8, 6e-8, 6.1e-5, 3e38, 0.5, 1.5, 2.5, -2.5], np.float32)])
    L = nso.lib()
    for v in vals:
        v = float(v)
        assert L.nso_f32_to_bf16(v) == refk.ref_f32_to_bf16(v)
        assert L.nso_f32_to_f16(v) == refk.ref_f32_to_f16(v)
    for h in list(range(0, 65536, 7)) + [0x7c00, 0x7bff, 0x0001, 0x03ff, 0x8001]:
        a = np.float32(L.nso_bf16_to_f32(h))
        b = np.float32(refk.ref_bf16_to_f32(h))
        assert a.view(np.uint32) == b.view(np.uint32)
        a = np.float32(L.nso_f16_to_f32(h))
        b = np.float32(refk.ref_f16_to_f32(h))
        assert a.view(np.uint32) == b.view(np.uint32)
    for x in np.linspace(-6, 6, 241, dtype=np.float32):
        assert np.float32(L.nso_gelu(float(x))) == np.float32(refk.ref_postop(float(x), 0))
        assert np.float32(L.nso_silu(float(x))) == np.float32(refk.ref_postop(float(x), 1))


@pytest.mark.parametrize("core,packrow", [("CORE_AVX512F", 1), ("CORE_AMX_BF16", 2), ("CORE_AVX512_VNNI_KB", 4)])
@pytest.mark.parametrize("stype", ["F32", "BF16"])
@pytest.mark.parametrize("asym", [False, True])
def test_unpack_matches_reference_tile_dequant(nso, refk, core, packrow, stype, asym):
    """oracle unpack (whole blob) == the reference's tile dequant decompress_kblock_s4_fp run over the blob's
    own packed image, scales and zero points."""
    rng = np.random.default_rng(21)
    n, k, bs = 96, 128, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, getattr(nso, stype), asym, getattr(nso, core))
    bi = nso.parse(blob)
    mine = nso.unpack_fp32(blob)  # [K][N]
    for t in range(bi.npad // 48):
        src = blob[bi.q_off + t * 48 * bi.kpad // 2:]
        dst = np.zeros(bi.kpad * 48, np.float32)
        sc = blob[bi.scale_off:]
        zp = blob[bi.zp_off:] if asym else None
        rc = refk.ref_decompress_kblock_s4_fp(packrow, nso.ptr(src), nso.ptr(dst), bi.kpad, nso.ptr(sc),
                                              C.c_uint32(bi.scale_dtype), nso.ptr(zp), 0, t * 48, bs, bi.cstep)
        assert rc == 0
        # dst is the tile in interleaved order [K/PR][48][PR]
        tile = dst.reshape(bi.kpad // packrow, 48, packrow).transpose(0, 2, 1).reshape(bi.kpad, 48)
        assert np.array_equal(tile[:k, :].view(np.uint32), mine[:, t * 48:(t + 1) * 48].view(np.uint32))


def test_unpack_s8_and_f4_match_reference(nso, refk):
    rng = np.random.default_rng(22)
    n, k, bs = 48, 64, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S8, nso.BF16, False, nso.CORE_AVX512F)
    bi = nso.parse(blob)
    dst = np.zeros(bi.kpad * 48, np.float32)
    refk.ref_decompress_kblock_s8_fp(1, nso.ptr(blob[bi.q_off:]), nso.ptr(dst), bi.kpad, nso.ptr(blob[bi.scale_off:]),
                                     C.c_uint32(bi.scale_dtype), None, 0, 0, bs, bi.cstep)
    assert np.array_equal(dst.reshape(bi.kpad, 48)[:k].view(np.uint32), nso.unpack_fp32(blob).view(np.uint32))
    for f4 in (nso.F4_NF4, nso.F4_BNB, nso.F4_E2M1):
        blob = nso.quant_pack(w, bs, f4, nso.F32, False, nso.CORE_AVX512F)
        bi = nso.parse(blob)
        dst = np.zeros(bi.kpad * 48, np.float32)
        sc = blob[bi.scale_off:bi.scale_off + bi.scale_bytes].view(np.float32).copy()
        rc = refk.ref_decompress_kblock_f4_fp(C.c_uint32(f4), 1, nso.ptr(blob[bi.q_off:]), nso.ptr(dst), bi.kpad, 48,
                                              nso.ptr(sc), 0, bs, bi.cstep)
        assert rc == 0
        assert np.array_equal(dst.reshape(bi.kpad, 48)[:k].view(np.uint32), nso.unpack_fp32(blob).view(np.uint32))


def test_reduce_matches_reference(nso, refk):
    rng = np.random.default_rng(23)
    n, k, bs = 48, 96, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    for asym in (False, True):
        blob = nso.quant_pack(w, bs, nso.S4, nso.BF16, asym, nso.CORE_AVX512_VNNI_KB)
        bi = nso.parse(blob)
        assert bi.has_reduce
        deq = nso.unpack_fp32(blob)
        red = blob[bi.red_off:bi.red_off + bi.red_bytes].view(np.uint16).reshape(-1, bi.cstep)
        for kb in range(k // bs):
            out = np.zeros(n, np.uint16)
            blk = np.ascontiguousarray(deq[kb * bs:(kb + 1) * bs])
            refk.ref_row_reduce_sum_bf16(nso.ptr(blk), n, bs, n, nso.ptr(out))
            assert np.array_equal(out, red[kb, :n])


@pytest.mark.parametrize("scale_bf16", [False, True])
@pytest.mark.parametrize("m", [1, 2, 4])
@pytest.mark.parametrize("asym", [False, True])
def test_gemv_fp32_matches_reference(nso, refk, scale_bf16, m, asym):
    """oracle sequential-fp32 GEMV == gemv_4bit_fp32_fp32 (kernel_ref.h:2489-2531), bit for bit."""
    rng = np.random.default_rng(31)
    n, k, bs = 96, 256, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, nso.BF16 if scale_bf16 else nso.F32, asym, nso.CORE_AVX512F)
    bi = nso.parse(blob)
    mine = nso.gemv_f32(a, blob, nthreads=1)
    for t in range(2):
        cref = np.zeros((m, 48), np.float32)
        b4 = blob[bi.q_off + t * 48 * bi.kpad // 2:]
        es = 2 if scale_bf16 else 4
        sc = blob[bi.scale_off + t * 48 * es:]
        zp = blob[bi.zp_off + t * 48:] if asym else None
        rc = refk.ref_gemv_4bit_fp32_fp32(nso.ptr(a), k, nso.ptr(b4), nso.ptr(sc), int(scale_bf16), nso.ptr(zp),
                                          bi.cstep, nso.ptr(cref), 48, k, bs, m)
        assert rc == 0
        assert np.array_equal(cref.view(np.uint32), mine[:, t * 48:(t + 1) * 48].view(np.uint32))
    # and the fp64 GEMM oracle agrees with it to fp32 round-off
    assert nso.rel_l2(mine, nso.gemm_f64(a, blob)) < 2e-6


def test_act_quant_and_u8s8_gemv(nso, refk):
    rng = np.random.default_rng(41)
    n, k, bs, m = 48, 256, 32, 1
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((m, k)).astype(np.float32)
    nb = k // bs
    aq = np.zeros((m, k), np.uint8)
    asc = np.zeros((m, nb), np.float32)
    azp = np.zeros((m, nb), np.uint8)
    red = np.zeros((m, nb), np.float32)
    aq2, asc2, azp2, red2 = [np.zeros_like(x) for x in (aq, asc, azp, red)]
    nso.lib().nso_quantize_fp_u8_colblock(m, k, nso.ptr(a), k, nso.ptr(aq), k, nso.ptr(asc), nb, nso.ptr(azp), bs, nso.ptr(red))
    refk.ref_quantize_fp_u8_colblock(m, k, nso.ptr(a), k, nso.ptr(aq2), k, nso.ptr(asc2), nb, nso.ptr(azp2), bs, nso.ptr(red2))
    assert np.array_equal(aq, aq2) and np.array_equal(azp, azp2)
    assert np.array_equal(asc.view(np.uint32), asc2.view(np.uint32)) and np.array_equal(red.view(np.uint32), red2.view(np.uint32))
    for asym in (False, True):
        blob = nso.quant_pack(w, bs, nso.S4, nso.F32, asym, nso.CORE_AVX512_VNNI_KB)
        bi = nso.parse(blob)
        mine = nso.gemm_u8s8(a, blob)
        cref = np.zeros((1, 48), np.float32)
        sc = blob[bi.scale_off:bi.scale_off + bi.scale_bytes].view(np.float32).copy()
        zp = blob[bi.zp_off:] if asym else None
        rc = refk.ref_gemv_4bit_u8s8_fp32(nso.ptr(aq), nso.ptr(asc), nso.ptr(azp), k, nb, nso.ptr(blob[bi.q_off:]),
                                          nso.ptr(sc), nso.ptr(zp), bi.cstep, nso.ptr(cref), 48, k, bs)
        assert rc == 0
        assert np.array_equal(cref.view(np.uint32), mine.view(np.uint32))


# ---------------------------------------------------------------------------------------------- fp8 weights
@pytest.mark.parametrize("f8", ["F8_E4M3", "F8_E5M2"])
def test_f8_decode_all_codes(nso, refk, f8):
    """f8_to_fp32 (kernel_ref.h:984-1002) for all 256 codes: no zero, no subnormals, no inf/nan in this encoding."""
    t = getattr(nso, f8)
    got = np.array([nso.lib().nso_f8_to_f32(t, c) for c in range(256)], np.float32)
    want = np.array([refk.ref_f8_to_f32(t, c) for c in range(256)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(np.isfinite(got)) and np.all(got != 0)


@pytest.mark.parametrize("f8", ["F8_E4M3", "F8_E5M2"])
@pytest.mark.parametrize("st", ["F8_E8M0", "F32"])
@pytest.mark.parametrize("kind", ["normal", "uniform", "adversarial"])
def test_quantize_f8(nso, refk, f8, st, kind):
    """quantize_f32_f8_rowblock_mxscale (kernel_ref.h:1763-1799): codes and scales bit for bit, incl. a tail block."""
    rng = np.random.default_rng(300 + len(kind))
    k, n, bs = 176, 24, 32   # 176 = 5 * 32 + 16: tail block
    w = _weights(rng, k, n, kind)
    if kind == "adversarial":
        w[40:48, 5] = np.float32(2.0) ** np.arange(-20, -12)      # tiny values next to ...
        w[48, 5] = 3.0                                            # ... a large one (clamped private exponent)
        w[64:96, 6] = np.nextafter(np.float32(0.25), np.float32(0))  # absmax just below a power of two
        w[96:128, 7] = 448.0 * 2.0 ** -9                          # exact max_norm multiples
    t, s_t = getattr(nso, f8), getattr(nso, st)
    q, sc, _ = nso.quantize(w, bs, t, stype=s_t)
    q_r = np.zeros_like(q)
    sc_r = np.zeros_like(sc)
    rc = refk.ref_quantize_f8(nso.ptr(w), nso.ptr(q_r), k, n, n, n, nso.ptr(sc_r), bs, C.c_uint32(t), C.c_uint32(s_t))
    assert rc == 0
    assert np.array_equal(sc.view(np.uint32), sc_r.view(np.uint32))
    assert np.array_equal(q, q_r)


@pytest.mark.parametrize("f8", ["F8_E4M3", "F8_E5M2"])
@pytest.mark.parametrize("st", ["F8_E8M0", "F32"])
@pytest.mark.parametrize("core,packrow", [("CORE_AVX512F", 1), ("CORE_AMX_BF16", 2)])
def test_unpack_f8_matches_reference_tile_dequant(nso, refk, f8, st, core, packrow):
    """blob -> fp32 through the oracle == the reference's decompress_kblock_f8_fp applied to the blob's own tiles."""
    rng = np.random.default_rng(77)
    n, k, bs = 96, 128, 32
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    blob = nso.quant_pack(w, bs, getattr(nso, f8), getattr(nso, st), False, getattr(nso, core))
    bi = nso.parse(blob)
    assert bi.prologue_id == 2 and bi.q_bytes == bi.npad * bi.kpad
    mine = nso.unpack_fp32(blob)
    nt = bi.ntile
    sbytes = 1 if st == "F8_E8M0" else 4
    for t in range(bi.npad // nt):
        tile = np.ascontiguousarray(blob[bi.q_off + t * nt * bi.kpad: bi.q_off + (t + 1) * nt * bi.kpad]).view(np.int8)
        sc = np.ascontiguousarray(
            blob[bi.scale_off: bi.scale_off + bi.scale_bytes].reshape(-1, bi.cstep * sbytes)[:, t * nt * sbytes:(t + 1) * nt * sbytes])
        # One call per k-block with the scale pointer advanced to that block's row: the scalar reference reads fp32
        # scales as scales[j / PACK_ROW] without the k-block offset (kernel_ref.h:1017-1018, a typo the AVX2/AVX512
        # product kernels do not have: kernel_avx512f.h:668-694 use sptr = scales + kpos * NPad for both scale types).
        dst = np.zeros((bi.kpad // packrow, nt * packrow), np.float32)
        rows_blk = bs // packrow
        for kb in range(bi.kpad // bs):
            src_blk = tile[kb * rows_blk * nt * packrow:]
            dst_blk = dst[kb * rows_blk:]
            sc_blk = np.ascontiguousarray(sc[kb])
            rc = refk.ref_decompress_kblock_f8_fp(C.c_uint32(getattr(nso, f8)), packrow, nso.ptr(src_blk),
                                                  nso.ptr(dst_blk), rows_blk, nt * packrow, nso.ptr(sc_blk),
                                                  int(st == "F8_E8M0"), 0, rows_blk, nt)
            assert rc == 0
        # dst[k / P][j * P + k % P] -> [k][j]
        deq = dst.reshape(bi.kpad // packrow, nt, packrow).transpose(0, 2, 1).reshape(bi.kpad, nt)
        cols = min(nt, bi.n - t * nt)
        assert np.array_equal(mine[:, t * nt:t * nt + cols].view(np.uint32), deq[:bi.k, :cols].view(np.uint32))


def test_streamed_u8s8_gemv_equals_canonical_form(nso):
    """nso_gemv_u8s8_f32 (reads the packed tiles, threaded: bench.py's CPU-baseline port of the default int8-compute decode
    path) adds exactly what nso_gemm_u8s8_f32 (unpacked, pinned above to gemv_4bit_u8s8_fp32) adds, in the same order"""
    rng = np.random.default_rng(31)
    for (n, k, bs, q, asym, core) in [(100, 256, 32, nso.S4, False, nso.CORE_AVX512_VNNI_KB), (64, 192, 64, nso.S8, True, nso.CORE_AMX_INT8_KB),
                                      (50, 160, 32, nso.S4, True, nso.CORE_AVX2_VNNI_KB), (48, 128, 128, nso.S4, False, nso.CORE_AVX512F)]:
        w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
        blob = nso.quant_pack(w, bs, q, nso.BF16, asym, core)
        a = rng.standard_normal((3, k)).astype(np.float32)
        assert np.array_equal(nso.gemm_u8s8(a, blob), nso.gemv_u8s8(a, blob, 4))


def test_dq8_code_map_equals_the_reference_table(nso, refk):
    """DQ8_BNB: the oracle BUILDS the 256-entry code map (bitsandbytes' dynamic map printed with five decimals) instead of holding a
    copy of the reference's table — every entry must be that table's (bestla_utils.h:794-...)"""
    import ctypes as C
    refk.ref_dq8_lut.restype = C.POINTER(C.c_float)
    nso.lib().nso_dq8_lut.restype = C.POINTER(C.c_float)
    r, o = refk.ref_dq8_lut(), nso.lib().nso_dq8_lut()
    ref = np.array([r[i] for i in range(256)], np.float32)
    mine = np.array([o[i] for i in range(256)], np.float32)
    assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), np.argwhere(ref != mine)[:5]
    assert np.all(np.diff(mine) >= 0) and mine[0] < -0.99 and mine[-1] == 1.0  # (the smallest magnitudes print as 0.00000 / 0.00001: ties)
