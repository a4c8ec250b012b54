"""The oracle's restatement of the BTLA blob container pinned to the reference's OWN container classes.

`oracle/_ref/libstor_ref.so` (make -C oracle storref) is bestla/bestla/bestla_storage.h compiled from where it lies
(StorageWeightKBlockNInteger :697-834, StorageWeightKBlockNFloat :836-859, PackedWeightParser :861-894).  For every
reference core x weight dtype x scale dtype x sym/asym x shuffle and every base alignment mod 64 the test compares:
  * the blob size,
  * every non-payload byte (sizes, ids, dtypes, flags, offsets, alignment pads) the reference's assign() writes,
  * the section offsets and sizes (the reference aligns sections by ABSOLUTE address, bestla_storage.h:85-109),
  * what the reference's PackedWeightParser reads back from a blob the ORACLE wrote.
The product writes its blobs with the same code path it is compared with bit for bit on the GPU (test_gpu_parity.py)."""
import ctypes as C
import itertools

import numpy as np
import pytest

import nso  # (tests/conftest.py puts oracle/ on the path)

stor = nso.storref()
pytestmark = pytest.mark.skipif(stor is None, reason="oracle/_ref/libstor_ref.so not built (reference tree absent)")

INT_Q = [nso.S1, nso.S2, nso.S3, nso.S4, nso.S5, nso.S6, nso.S7, nso.S8]
F4_Q = [nso.F4_NF4, nso.F4_BNB, nso.F4_E2M1]
F8_Q = [nso.F8_E4M3, nso.F8_E5M2]
SHAPES = [(48, 128, 32), (50, 192, 64), (96, 256, 128), (33, 160, 32), (64, 128, -1)]
ALIGN_FEW = [0, 1, 8, 17, 32, 63]


def _buf(nbytes, off):
    """zeroed buffer whose data pointer is == off (mod 64)"""
    raw = np.zeros(nbytes + 192, dtype=np.uint8)
    base = (-raw.ctypes.data) % 64 + off
    return raw[base:base + nbytes]


def _ref_layout(bi, buf, shuffle):
    out = (C.c_uint64 * 24)()
    size = stor.stor_assign(int(bi.prologue_id == 2), bi.core_id, bi.npad, bi.kpad, bi.blocksize, bi.n, bi.k, bi.dtype,
                            bi.scale_dtype, bi.red_dtype, int(bi.is_asym), int(shuffle),
                            buf.ctypes.data if buf is not None else None, out)
    return size, list(out)


def _check(n, k, bs, qtype, stype, asym, core, off, g_idx=False):
    rng = np.random.default_rng(1)
    blocksize = k if bs < 0 else bs
    is_int = nso.is_int_type(qtype)
    q = rng.integers(-1, 2, size=(k, n)).astype(np.int8)
    nb = nso.nblk(k, blocksize)
    sc = (rng.random((nb, n)).astype(np.float32) + 0.5)
    zp = rng.integers(-2, 3, size=(nb, n)).astype(np.int8) if asym else None
    if g_idx:
        size = nso.lib().nso_pack_size_gidx(n, k, bs, C.c_uint32(qtype), C.c_uint32(stype), int(asym), core)
    else:
        size = nso.pack_size(n, k, bs, qtype, stype, asym, core)
    if size == 0:
        return False  # combination the reference does not offer
    blob = _buf(size, off)
    if is_int:
        if g_idx:
            gi = (np.arange(k) // blocksize).astype(np.int32)
            rc = nso.lib().nso_pack_q_gidx(nso.ptr(blob), nso.ptr(q), n, nso.ptr(sc), nso.ptr(zp), n, k, bs, C.c_uint32(qtype),
                                           C.c_uint32(stype), int(asym), core, nso.ptr(gi))
        else:
            rc = nso.lib().nso_pack_q(nso.ptr(blob), nso.ptr(q), n, nso.ptr(sc), nso.ptr(zp), n, k, bs, C.c_uint32(qtype),
                                      C.c_uint32(stype), int(asym), core)
    else:
        w = rng.standard_normal((n, k)).astype(np.float32)
        rc = nso.lib().nso_quant_pack(nso.ptr(blob), nso.ptr(w), n, k, k, bs, C.c_uint32(qtype), C.c_uint32(stype), 0, core, 1)
    assert rc == 0
    bi = nso.parse(blob)
    # 1. the reference's container, same parameters, same base alignment
    rbuf = _buf(size, off)
    rsize, lay = _ref_layout(bi, rbuf, g_idx)
    assert rsize == size == bi.size, (rsize, size, bi.size)
    sections = [(bi.q_off, bi.q_bytes), (bi.scale_off, bi.scale_bytes), (bi.zp_off, bi.zp_bytes), (bi.red_off, bi.red_bytes),
                (bi.shuf_off, bi.shuf_bytes)]
    for i, (o, b) in enumerate(sections):
        assert (lay[2 * i], lay[2 * i + 1]) == (o, b), ("section", i, lay[2 * i], lay[2 * i + 1], o, b)
    assert lay[10] == bi.cstep and lay[11] == bi.csize
    # 2. every non-payload byte
    masked = blob.copy()
    for o, b in sections:
        masked[o:o + b] = 0
    assert np.array_equal(masked, rbuf), "header / pad bytes differ at base alignment %d" % off
    # 3. the reference parser on the oracle's blob
    out = (C.c_uint64 * 24)()
    assert stor.stor_deserialize(blob.ctypes.data, out) == 0
    got = list(out)
    for i, (o, b) in enumerate(sections):
        assert (got[2 * i], got[2 * i + 1]) == (o, b), ("parsed section", i)
    assert got[12] == bi.size and got[13] == bi.prologue_id and got[14] == bi.core_id
    assert got[15] == (bi.npad << 32 | bi.kpad) and got[16] == (bi.n << 32 | bi.k)
    assert got[17] == bi.dtype and got[18] == (bi.blocksize << 32 | (bi.dq_blocksize & 0xffffffff))
    assert got[19] == bi.scale_dtype
    if bi.prologue_id == 1:
        assert got[20] == bi.zp_dtype and got[21] == bi.red_dtype
    return True


@pytest.mark.parametrize("core", range(9))
def test_integer_blobs_every_core(core):
    done = 0
    for (n, k, bs), qtype, stype, asym in itertools.product(SHAPES, INT_Q, [nso.F32, nso.BF16, nso.F16], [False, True]):
        for off in ALIGN_FEW:
            done += bool(_check(n, k, bs, qtype, stype, asym, core, off))
    assert done > 0


@pytest.mark.parametrize("core", range(9))
def test_float_blobs_every_core(core):
    done = 0
    for (n, k, bs), qtype, stype in itertools.product(SHAPES, F4_Q, [nso.F32, nso.BF16]):
        for off in ALIGN_FEW:
            done += bool(_check(n, k, bs, qtype, stype, False, core, off))
    for (n, k, bs), qtype, stype in itertools.product(SHAPES[:3], F8_Q, [nso.F32, nso.F8_E8M0]):
        for off in ALIGN_FEW:
            done += bool(_check(n, k, bs, qtype, stype, False, core, off))
    assert done > 0


def test_every_base_alignment_mod_64():
    """the full sweep 0..63 on a representative set (sections move with the absolute address of the blob)"""
    cases = [(48, 128, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB), (50, 192, 64, nso.S4, nso.F32, True, nso.CORE_AMX_INT8_KB),
             (33, 160, 32, nso.S3, nso.BF16, False, nso.CORE_AVX512F), (96, 256, 128, nso.S8, nso.F16, True, nso.CORE_AVX2),
             (64, 128, -1, nso.S5, nso.BF16, False, nso.CORE_AVX_VNNI_KB), (48, 128, 32, nso.F4_NF4, nso.BF16, False, nso.CORE_AVX512F),
             (50, 192, 64, nso.F8_E4M3, nso.F8_E8M0, False, nso.CORE_AVX512F), (96, 256, 32, nso.S2, nso.F32, True, nso.CORE_AMX_BF16)]
    done = 0
    for (n, k, bs, qtype, stype, asym, core) in cases:
        for off in range(64):
            done += bool(_check(n, k, bs, qtype, stype, asym, core, off))
    assert done >= 64 * 6


def test_shuffle_section_every_alignment():
    """GPTQ act-order blobs: the optional ShuffleIndices section (bestla_storage.h:704, :772-776)"""
    for core in (nso.CORE_AVX512_VNNI_KB, nso.CORE_AVX512F, nso.CORE_AMX_INT8_KB):
        for off in range(0, 64, 3):
            assert _check(48, 128, 32, nso.S4, nso.BF16, False, core, off, g_idx=True)
            assert _check(50, 192, 64, nso.S4, nso.F32, True, core, off, g_idx=True)
