"""The blob loader does not trust the header (a truncated or corrupt model file must be refused before any
kernel walks its sections).  Every golden blob — minted by the real reference storage classes — validates; the same
blobs with one header field damaged do not.  Header-only: runs without a GPU."""
import os
import struct

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "btla_golden.npz"))
NAMES = [str(x) for x in G["names"]]

# serialized header (bestla_storage.h:52-92 ObjectAlignedBuffer / :690-760 StorageWeightKBlockNInteger): u64 size,
# u32 prologue, u64 core id, i32 npad kpad n k, u32 dtype, i32 blocksize dq_blocksize, then {u64 bytes, u64 pad} of
# the code section
OFF_SIZE, OFF_NPAD, OFF_KPAD, OFF_N, OFF_K, OFF_DTYPE, OFF_BLOCK, OFF_QBYTES = 0, 20, 24, 28, 32, 36, 40, 48


def _blob(nso, name):
    b = nso.aligned_bytes(G[name + "/blob"].size)
    b[:] = G[name + "/blob"]
    return b


@pytest.mark.parametrize("name", NAMES)
def test_golden_blobs_validate(L, nso, name):
    b = _blob(nso, name)
    assert struct.unpack_from("<Q", b, OFF_SIZE)[0] == b.size
    assert L.ns_hip_blob_validate(nso.ptr(b), b.size) == 0, L.ns_hip_last_error()
    assert L.ns_hip_blob_validate(nso.ptr(b), 0) == 0
    # the file ends before the blob does
    assert L.ns_hip_blob_validate(nso.ptr(b), b.size - 1) != 0
    assert b"exceeds" in L.ns_hip_last_error()


DAMAGE = [
    ("size halved", OFF_SIZE, "<Q", lambda v: v // 2),
    ("size = header only", OFF_SIZE, "<Q", lambda v: 60),
    ("npad x 4", OFF_NPAD, "<i", lambda v: v * 4),
    ("kpad x 4", OFF_KPAD, "<i", lambda v: v * 4),
    ("n beyond npad", OFF_N, "<i", lambda v: v * 8),
    ("k = 0", OFF_K, "<i", lambda v: 0),
    ("blocksize = 0", OFF_BLOCK, "<i", lambda v: 0),
    ("code bytes halved", OFF_QBYTES, "<Q", lambda v: v // 2),
    ("code bytes huge", OFF_QBYTES, "<Q", lambda v: 1 << 60),
]


@pytest.mark.parametrize("what,off,fmt,fn", DAMAGE, ids=[d[0] for d in DAMAGE])
@pytest.mark.parametrize("name", NAMES[:6])
def test_damaged_headers_are_refused(L, nso, name, what, off, fmt, fn):
    b = _blob(nso, name)
    old = struct.unpack_from(fmt, b, off)[0]
    struct.pack_into(fmt, b, off, fn(old))
    assert L.ns_hip_blob_validate(nso.ptr(b), 0) != 0, what
    assert L.ns_hip_last_error().startswith(b"blob:")


def test_small_blocksize_needs_more_scales(L, nso):
    """blocksize / 2 with everything else unchanged: the scale section is now half of what the geometry needs"""
    for name in NAMES:
        b = _blob(nso, name)
        bs = struct.unpack_from("<i", b, OFF_BLOCK)[0]
        kpad = struct.unpack_from("<i", b, OFF_KPAD)[0]
        if bs < 2 or kpad // bs < 1:
            continue
        struct.pack_into("<i", b, OFF_BLOCK, bs // 2)
        assert L.ns_hip_blob_validate(nso.ptr(b), 0) != 0, name


def test_null_blob(L):
    assert L.ns_hip_blob_validate(None, 0) != 0
