"""The operand algebra of the int8-reference GEMM kernel (csrc/ns_i8g2.hip i8mfma2_kernel, csrc/ns_i8ref.hip i8prep_kernel,
csrc/ns_quant.hip aquant_u8_vec_kernel), restated bit for bit in numpy: both zero points folded into fp16 operands that hold
small integers, so that an fp16 MFMA with fp32 accumulation returns the reference's integer dot

    sum_k (a_k - za) (q_k - zb)        (gemv_4bit_u8s8_fp32, kernel_ref.h:2371-2429; ref_kblock_int8, bestla/ut/bestla_gemm.cpp:159-190)

exactly.  Checked here: every bit trick the kernels use to build the operands (0x6400 | code = 1024 + code as fp16, the odd
nibble pairs taken in place as 16 (u - zbb) against A' / 16, the byte permutes of the byte containers) yields exactly those
integers, every product and every partial sum of a 32-deep slice stays below 2^24 (so fp32 accumulation in any order is
exact), and the fp32 result equals the integer dot.  No GPU: this is the host-side statement of what the GPU tests
(tests/test_gpu_int8_mode.py) then find bit-identical between the two matrix-core kernels."""
import numpy as np
import pytest

MAGIC = np.uint32(0x64006400)


def _half2(word):
    """the two fp16 values of a dword (low half first)"""
    return np.array([word & 0xffff, word >> 16], dtype=np.uint16).view(np.float16)


def _perm(s0, s1, sel):
    """v_perm_b32 D = perm(S0, S1, sel): selector bytes 0-3 pick bytes of S1, 4-7 bytes of S0"""
    pool = [(int(s1) >> (8 * i)) & 0xff for i in range(4)] + [(int(s0) >> (8 * i)) & 0xff for i in range(4)]
    out = 0
    for i in range(4):
        out |= pool[(sel >> (8 * i)) & 0xff] << (8 * i)
    return out


def _a_prime(a, za, scale16):
    """i8prep_kernel / aquant_u8_vec_kernel: eight u8 codes -> eight fp16 (pairs 1 and 3 divided by 16 for nibble containers)"""
    lo = int(a[0]) | int(a[1]) << 8 | int(a[2]) << 16 | int(a[3]) << 24
    hi = int(a[4]) | int(a[5]) << 8 | int(a[6]) << 16 | int(a[7]) << 24
    zc = _half2((0x6400 + za) * 0x00010001)
    words = [_perm(0x64646464, lo, 0x04010400), _perm(0x64646464, lo, 0x04030402),
             _perm(0x64646464, hi, 0x04010400), _perm(0x64646464, hi, 0x04030402)]
    out = []
    for i, w in enumerate(words):
        v = _half2(w) - zc                      # fp16 arithmetic, as v_pk_add_f16
        if (i & 1) and scale16:
            v = v * np.float16(0.0625)
        out.append(v.astype(np.float16))
    return np.concatenate(out)


def _b_prime_nibbles(u, zbb):
    """i8mfma2_kernel, nibble containers: the record dword holds codes (0,4,1,5) in its low nibbles and (2,6,3,7) in its high ones"""
    x = 0
    for byte, (lo_code, hi_code) in enumerate([(0, 2), (4, 6), (1, 3), (5, 7)]):
        x |= (int(u[lo_code]) | int(u[hi_code]) << 4) << (8 * byte)
    y = x >> 8
    z1 = _half2((0x6400 + zbb) * 0x00010001)
    z16 = _half2((0x6400 + (zbb << 4)) * 0x00010001)
    bw = [_half2((x & 0x000f000f) | int(MAGIC)) - z1, _half2((x & 0x00f000f0) | int(MAGIC)) - z16,
          _half2((y & 0x000f000f) | int(MAGIC)) - z1, _half2((y & 0x00f000f0) | int(MAGIC)) - z16]
    return np.concatenate(bw).astype(np.float16)


def _b_prime_bytes(q, zb):
    """i8mfma2_kernel, byte containers: eight s8 codes in k order as two dwords"""
    b = [int(v) & 0xff for v in q]
    x0 = (b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24) ^ 0x80808080
    x1 = (b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24) ^ 0x80808080
    zc = _half2((0x6480 + zb) * 0x00010001)
    bw = [_half2(_perm(0x64646464, x0, 0x04010400)) - zc, _half2(_perm(0x64646464, x0, 0x04030402)) - zc,
          _half2(_perm(0x64646464, x1, 0x04010400)) - zc, _half2(_perm(0x64646464, x1, 0x04030402)) - zc]
    return np.concatenate(bw).astype(np.float16)


def _corner(rng, lo, hi, n):
    return rng.choice([lo, hi, lo + 1, hi - 1], n)


@pytest.mark.parametrize("seed", range(4))
def test_nibble_operands_are_the_exact_integers(seed):
    rng = np.random.default_rng(seed)
    for trial in range(600):
        a = rng.integers(0, 256, 8) if trial % 5 else _corner(rng, 0, 255, 8)
        za = int(rng.integers(0, 256)) if trial % 3 else int(_corner(rng, 0, 255, 1)[0])
        u = rng.integers(0, 16, 8) if trial % 7 else _corner(rng, 0, 15, 8)
        zbb = int(rng.integers(0, 16)) if trial % 2 else 8   # zb + 8: symmetric weights have zb = 0
        ap, bp = _a_prime(a, za, True), _b_prime_nibbles(u, zbb)
        scale = np.array([1, 1, 16, 16, 1, 1, 16, 16], np.float64)
        assert np.array_equal(ap.astype(np.float64) * scale, (a - za).astype(np.float64))      # A' (pairs 1, 3: / 16, exactly)
        assert np.array_equal(bp.astype(np.float64) / scale, (u - zbb).astype(np.float64))     # B' (pairs 1, 3: x 16, exactly)
        prod = ap.astype(np.float32) * bp.astype(np.float32)                                   # what the MFMA multiplies
        assert np.array_equal(prod.astype(np.int64), (a - za) * (u - zbb))
        assert float(np.add.reduce(prod, dtype=np.float32)) == float(((a - za) * (u - zbb)).sum())


@pytest.mark.parametrize("seed", range(4))
def test_byte_operands_are_the_exact_integers(seed):
    rng = np.random.default_rng(100 + seed)
    for trial in range(600):
        a = rng.integers(0, 256, 8) if trial % 5 else _corner(rng, 0, 255, 8)
        za = int(rng.integers(0, 256)) if trial % 3 else int(_corner(rng, 0, 255, 1)[0])
        q = rng.integers(-128, 128, 8) if trial % 7 else _corner(rng, -128, 127, 8)
        zb = int(rng.integers(-128, 128)) if trial % 2 else 0
        ap, bp = _a_prime(a, za, False), _b_prime_bytes(q, zb)
        assert np.array_equal(ap.astype(np.int64), a - za)
        assert np.array_equal(bp.astype(np.int64), q - zb)
        prod = ap.astype(np.float32) * bp.astype(np.float32)
        assert float(np.add.reduce(prod, dtype=np.float32)) == float(((a - za) * (q - zb)).sum())


def test_slice_sums_stay_below_2_pow_24_in_any_order():
    """32-deep slice, worst case: |a - za| = 255 everywhere; nibbles |u - zbb| <= 15, bytes |q - zb| <= 255.  Every partial sum
    is an integer of magnitude below 2^24, so fp32 accumulation (whatever order the matrix core uses) never rounds."""
    assert 32 * 255 * 15 < 2 ** 24 and 32 * 255 * 255 < 2 ** 24
    rng = np.random.default_rng(7)
    for width in (15, 255):
        terms = (rng.choice([-255, 255], 32) * rng.choice([-width, width], 32)).astype(np.float32)
        for _ in range(50):
            order = rng.permutation(32)
            acc = np.float32(0)
            for t in terms[order]:
                acc = np.float32(acc + t)
            assert float(acc) == float(terms.astype(np.float64).sum())
