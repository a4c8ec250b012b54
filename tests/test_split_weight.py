"""ns_bestla_split_weight: tensor-parallel shards cut out of reference-format blobs on the host (the reference's per-rank
bestla_split_weight, model_files.h:1538-1563 / :1593-1640, without its dequantise -> re-quantise round trip).  Quantisation is
per (k-block, column), so a cut on block boundaries must give BYTE FOR BYTE the blob the oracle packs from the cut matrix —
checked here for the split rules of model_files.h:145-190 (ROW = columns / rank, COLUMN = rows of K / rank) on every container
the packer produces."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def S(L):
    L.ns_bestla_split_weight_size.restype = C.c_ulonglong
    L.ns_bestla_split_weight_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ns_bestla_split_weight.restype = C.c_int
    L.ns_bestla_split_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int]
    return L


def _cut(S, nso, blob, n0, n1, k0, k1):
    need = S.ns_bestla_split_weight_size(nso.ptr(blob), n1 - n0, k1 - k0)
    assert need > 0
    out = nso.aligned_bytes(int(need))
    out[:] = 0xAB
    rc = S.ns_bestla_split_weight(nso.ptr(blob), nso.ptr(out), need, n0, n1, k0, k1)
    return rc, out


FORMATS = [("int4_g32_vnni", "S4", 32, "BF16", False, "CORE_AVX512_VNNI_KB"),
           ("int4_g128_asym_f32", "S4", 128, "F32", True, "CORE_AVX512_VNNI_KB"),
           ("int8_g64_amx", "S8", 64, "BF16", False, "CORE_AMX_INT8_KB"),
           ("int3_g32", "S3", 32, "BF16", False, "CORE_AVX512_VNNI_KB"),
           ("int5_g32_asym", "S5", 32, "F16", True, "CORE_AVX512_VNNI_KB"),
           ("nf4_g64", "F4_NF4", 64, "BF16", False, "CORE_AVX512F"),
           ("fp8_e4m3_g32", "F8_E4M3", 32, "F32", False, "CORE_AVX512F"),
           ("int4_per_channel", "S4", -1, "BF16", False, "CORE_AVX512F")]


@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
@pytest.mark.parametrize("world", [2, 4])
def test_row_and_column_shards_equal_the_packed_cut_matrix(S, nso, fmt, world):
    name, qt, bs, st, asym, core = fmt
    rng = np.random.default_rng(len(name) + world)
    n, k = 96 * world, 256 * world
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    pack = lambda m: nso.quant_pack(m, bs, getattr(nso, qt), getattr(nso, st), asym, getattr(nso, core))
    full = pack(w)
    for rank in range(world):
        # TP_1D_ROW (wq/wk/wv/w1/w3): the rank's columns of the [N][K] weight
        n0, n1 = rank * n // world, (rank + 1) * n // world
        rc, got = _cut(S, nso, full, n0, n1, 0, k)
        want = pack(w[n0:n1])
        assert rc == 0 and np.array_equal(got[:want.size], want), (name, "row", rank)
        if bs > 0:  # TP_1D_COLUMN (wo/w2): the rank's slice of K (per-channel scales span all of K: refused below)
            k0, k1 = rank * k // world, (rank + 1) * k // world
            rc, got = _cut(S, nso, full, 0, n, k0, k1)
            want = pack(np.ascontiguousarray(w[:, k0:k1]))
            assert rc == 0 and np.array_equal(got[:want.size], want), (name, "column", rank)


def test_ragged_cuts_and_refusals(S, nso, L):
    rng = np.random.default_rng(1)
    n, k = 200, 512
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    pack = lambda m, bs=32: nso.quant_pack(m, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    full = pack(w)
    # columns that start inside a 48-wide tile, a K range that ends at K (ragged last block of the shard is fine)
    rc, got = _cut(S, nso, full, 50, 171, 64, 512)
    want = pack(np.ascontiguousarray(w[50:171, 64:512]))
    assert rc == 0 and np.array_equal(got[:want.size], want)
    rc, _ = _cut(S, nso, full, 0, n, 16, 512)        # K cut inside a 32-deep block: the re-quantising route's job
    assert rc == -2
    rc, _ = _cut(S, nso, full, 0, n, 0, 100)         # K end inside a block
    assert rc == -2
    assert S.ns_bestla_split_weight(nso.ptr(full), nso.ptr(nso.aligned_bytes(64)), 64, 0, n, 0, k) == -1   # destination too small
    assert S.ns_bestla_split_weight_size(nso.ptr(full), n + 1, k) == 0
    L.ns_hip_reset_error()


def test_dq8_scaled_blob_is_parsed_on_the_host_and_not_cut(S, nso, L):
    """DQ8_BNB scales (round 4): the host-side header parser accepts such a blob (shape query) — it was refused outright before — and
    the exact cut declines it: dq blocks run across rows and columns of the scale array, so the rank's part is not a byte range"""
    rng = np.random.default_rng(8)
    w = (rng.standard_normal((96, 512)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, 32, nso.S4, nso.DQ8_BNB, False, nso.CORE_AVX512F)
    n, k = C.c_int(0), C.c_int(0)
    L.ns_blob_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    assert L.ns_blob_shape(nso.ptr(blob), C.byref(n), C.byref(k)) == 0 and (n.value, k.value) == (96, 512)
    out = nso.aligned_bytes(len(blob))
    assert S.ns_bestla_split_weight(nso.ptr(blob), nso.ptr(out), len(blob), 0, 48, 0, 512) == -2
    assert b"DQ8_BNB" in L.ns_hip_last_error()
    # a blob whose scale dtype says DQ8 but that carries no double-quantisation section is not a blob
    bi = nso.parse(blob)
    plain = nso.quant_pack(w, 32, nso.S4, nso.F32, False, nso.CORE_AVX512F)
    pbi = nso.parse(plain)
    assert pbi.dq_bytes == 0 and bi.dq_bytes == (96 * 16 // 32 + 1) * 4
