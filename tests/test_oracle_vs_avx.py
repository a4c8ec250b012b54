"""The oracle (= the reference's SCALAR kernels, bit for bit: tests/test_oracle_vs_ref.py) against the reference's VECTOR
kernels — kernel_avx512f.h / kernel_avx2.h compiled from where they lie into oracle/_ref/libkernel_avx_ref.so
(oracle/Makefile avxref) — for the two quantizers that kernel_wrapper.h:546-603 sends to the vector ISA on an AVX512 / AVX2
host: the F4 weight quantizer and the u8 activation quantizer of the int8-compute path.  (The integer weight quantizer
always runs the scalar kernel, :540-543.)  The product follows the scalar kernels; this file records how far the
vector kernels are from them, so that "bit-exact with the reference" is a statement about a named kernel."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def avx(nso):
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if not all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512_vnni")):
        pytest.skip("no AVX512 (F / BW / VL / DQ / VNNI: the code generation the library is compiled with) on this host")
    so = os.path.join(HERE, "..", "oracle", "_ref", "libkernel_avx_ref.so")
    if not os.path.exists(so):
        if not os.path.exists("/root/reference/bestla/bestla/kernel_avx512f.h"):
            pytest.skip("oracle/_ref/libkernel_avx_ref.so not built (reference tree absent)")
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(HERE, "..", "oracle"), "avxref"], stdout=subprocess.DEVNULL)
    return C.CDLL(so)


@pytest.mark.parametrize("f4", ["F4_NF4", "F4_BNB", "F4_E2M1"])
def test_f4_weight_quantizer_avx512_is_a_different_encoder(nso, avx, f4):
    """What an AVX512 host writes for an F4 weight is NOT the scalar kernel's bits, by construction
    (kernel_avx512f.h:1189-1196, :1082-1148 vs kernel_ref.h:1802-1822):
      * the block scale keeps the SIGN of the largest-magnitude element (vrangeps imm 7: absolute max, sign of the selected
        operand, the later element on a tie) where the scalar kernel stores |max| — same magnitude bit for bit;
      * x / scale is x * rcp14(scale), a 14-bit table reciprocal, where the scalar kernel multiplies by the exact 1.f / absmax;
      * codes come from sixteen (eight) sequential `<=` threshold tests instead of the scalar decision tree.
    Both are valid encodings of the same format — the dequantizer is LUT[code] * scale either way — and the product (like the
    oracle) reproduces the scalar one, the only one defined without an ISA-specific approximation table.  What must hold for
    a drop-in: blobs from EITHER encoder load and compute correctly (negative scales: tests/test_gpu_parity.py), and the two
    encodings reconstruct the weights equally well — checked here."""
    qt = getattr(nso, f4)
    rng = np.random.default_rng(3)
    k, n, bs = 256, 64, 32
    nf = nso.lib().nso_f4_unpack
    nf.restype = C.c_float
    lut = np.array([nf(C.c_uint32(qt), c) for c in range(16)], np.float32)
    for w in ((rng.standard_normal((k, n)) * 0.02).astype(np.float32), rng.uniform(-0.5, 0.5, (k, n)).astype(np.float32)):
        q, sc, _ = nso.quantize(w, bs, qt)
        q2, sc2 = np.zeros_like(q), np.zeros_like(sc)
        assert avx.avx512_quantize_f4(nso.ptr(w), nso.ptr(q2), k, n, n, n, nso.ptr(sc2), bs, C.c_uint32(qt)) == 0
        # same magnitude, sign of the block's largest element
        assert np.array_equal(np.abs(sc2).view(np.uint32), sc.view(np.uint32))
        wb = w.reshape(k // bs, bs, n)
        big = np.take_along_axis(wb, np.abs(wb).argmax(1)[:, None, :], 1)[:, 0, :]
        assert np.array_equal(np.signbit(sc2), np.signbit(big))
        assert np.signbit(sc2).mean() > 0.3          # i.e. about half of the blocks
        # reconstruction: both encodings are equally close to the weights (the signed one a little closer: its largest element
        # always lands on the exact code 1.0)
        deq = lambda qq, ss: lut[qq.astype(np.int64) & 15] * np.repeat(ss, bs, axis=0)
        e_scalar = np.linalg.norm(deq(q, sc) - w) / np.linalg.norm(w)
        e_avx = np.linalg.norm(deq(q2, sc2) - w) / np.linalg.norm(w)
        print("%s: rel. reconstruction error scalar %.4f, avx512 %.4f; codes equal on %.1f %% of the positive-scale blocks" % (
            f4, e_scalar, e_avx, 100 * (q == q2)[np.repeat(~np.signbit(sc2), bs, axis=0)].mean()))
        assert e_avx < e_scalar * 1.02 and e_scalar < e_avx * 1.10
        # where the scale came out positive the two encoders see the same normalised values up to rcp14's 2^-14: codes agree
        # except next to a threshold
        pos = np.repeat(~np.signbit(sc2), bs, axis=0)
        assert (q == q2)[pos].mean() > 0.995


@pytest.mark.parametrize("isa", ["avx512", "avx2"])
def test_u8_activation_quantizer_vector_kernels_vs_scalar(nso, avx, isa):
    """quantize_fp_u8_colblock: scales and zero points of the vector kernels == scalar (same min / max reductions); the
    CODES may differ by one where (x - min) / scale lands within an ulp of .5 (the vector kernels multiply by a reciprocal
    and round to nearest-even in the cvt instruction, the scalar kernel divides and uses roundf) — measured and bounded."""
    rng = np.random.default_rng(8)
    m, k, bs = 64, 4096, 32
    a = rng.standard_normal((m, k)).astype(np.float32)
    nb = k // bs
    f = getattr(avx, "%s_quantize_fp_u8_colblock" % isa)
    outs = []
    for fn in (nso.lib().nso_quantize_fp_u8_colblock, f):
        aq, asc, azp, red = np.zeros((m, k), np.uint8), np.zeros((m, nb), np.float32), np.zeros((m, nb), np.uint8), np.zeros((m, nb), np.float32)
        fn(m, k, nso.ptr(a), k, nso.ptr(aq), k, nso.ptr(asc), nb, nso.ptr(azp), bs, nso.ptr(red))
        outs.append((aq, asc, azp, red))
    (aq, asc, azp, red), (aq2, asc2, azp2, red2) = outs
    assert np.array_equal(asc.view(np.uint32), asc2.view(np.uint32)), "scales differ"
    assert np.array_equal(azp, azp2), "zero points differ"
    d = np.abs(aq.astype(np.int32) - aq2.astype(np.int32))
    frac = float((d != 0).mean())
    print("%s u8 activation codes differing from the scalar kernel: %.4f %% (max |diff| %d)" % (isa, 100 * frac, d.max()))
    assert d.max() <= 1 and frac < 5e-3


@pytest.mark.parametrize("n,k,bs,st,asym", [(4096, 4096, 32, "BF16", False), (200, 1024, 128, "F32", True), (688, 2048, 64, "BF16", True)])
def test_reference_avx512_vnni_decode_gemv_equals_the_oracle(nso, avx, n, k, bs, st, asym):
    """the reference's real decode hot loop (avx512f::vnni::gemv_4bit_u8s8_fp32 per 48-column tile, fed by its AVX512 activation
    quantizer) on the oracle's blobs == the oracle's restatement of the scalar gemv_4bit_u8s8_fp32: the blob layout the
    product packs is the one the reference's vector kernels read, and the int8-compute numbers the oracle defines are theirs
    (1e-6: fp32 summation order + the activation codes that differ by one, see above)"""
    if nso.avxref() is None:
        pytest.skip("no AVX512-VNNI on this host")
    rng = np.random.default_rng(n + k)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    a = rng.standard_normal((1, k)).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, getattr(nso, st), asym, nso.CORE_AVX512_VNNI_KB)
    got = nso.gemv_u8s8_avx512vnni(a, blob, 4).copy()
    assert nso.rel_l2(got, nso.gemv_u8s8(a, blob, 4)) < 5e-6


@pytest.mark.parametrize("f8", ["F8_E4M3", "F8_E5M2"])
@pytest.mark.parametrize("st", ["F8_E8M0", "F32"])
@pytest.mark.parametrize("core,packrow", [("CORE_AVX512F", 1), ("CORE_AMX_BF16", 2)])
def test_f8_unpack_equals_the_avx512_tile_dequant_in_one_call(nso, avx, f8, st, core, packrow):
    """the oracle's (and the product's) fp8 unpack == avx512f::decompress_kblock_f8_fp over a WHOLE tile in one call: the
    vector kernel advances the scale row per k-block itself (sptr = scales + kpos * NPad) for both scale types — the form the
    scalar kernel only reproduces when it is called one k-block at a time (its fp32-scale branch drops the k-block offset,
    kernel_ref.h:1017-1018; tests/test_oracle_vs_ref.py)"""
    rng = np.random.default_rng(77)
    n, k, bs = 96, 256, 32
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    blob = nso.quant_pack(w, bs, getattr(nso, f8), getattr(nso, st), False, getattr(nso, core))
    bi = nso.parse(blob)
    mine = nso.unpack_fp32(blob)
    nt = bi.ntile
    sbytes = 1 if st == "F8_E8M0" else 4
    for t in range(bi.npad // nt):
        tile = np.ascontiguousarray(blob[bi.q_off + t * nt * bi.kpad: bi.q_off + (t + 1) * nt * bi.kpad]).view(np.int8)
        # this tile's scale columns with the blob's own row stride: the kernel is told NPad = cstep
        sc = np.ascontiguousarray(blob[bi.scale_off + t * nt * sbytes: bi.scale_off + bi.scale_bytes])
        dst = np.zeros((bi.kpad // packrow, nt * packrow), np.float32)
        rc = avx.avx512_decompress_kblock_f8_fp(C.c_uint32(getattr(nso, f8)), packrow, nso.ptr(tile), nso.ptr(dst), bi.kpad // packrow,
                                                nt * packrow, nso.ptr(sc), int(st == "F8_E8M0"), 0, bs // packrow, bi.cstep)
        assert rc == 0
        deq = dst.reshape(bi.kpad // packrow, nt, packrow).transpose(0, 2, 1).reshape(bi.kpad, nt)
        cols = min(nt, bi.n - t * nt)
        assert np.array_equal(mine[:, t * nt:t * nt + cols].view(np.uint32), deq[:bi.k, :cols].view(np.uint32))
