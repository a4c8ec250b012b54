"""Known-answer tests against tests/golden/btla_golden.npz (minted from the real reference kernel_ref.h by
tests/golden/make_golden.py).  CPU part: the oracle reproduces every golden byte.  GPU part: the HIP product does."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "btla_golden.npz"))
NAMES = [str(x) for x in G["names"]]
# formats the MI355X forward kernels cover today (the quantizer/packer covers all of them)
FWD_OK = lambda qt: qt in (4 | (1 << 8), 8 | (1 << 8), 4, 4 | (1 << 16), 4 | (2 << 16), 8, 8 | (1 << 16))


def _meta(name):
    qt, st, asym, core, bs, n, k = [int(x) for x in G[name + "/meta"]]
    return qt, st, bool(asym), core, bs, n, k


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(nso, name):
    qt, st, asym, core, bs, n, k = _meta(name)
    w, a = G[name + "/w"], G[name + "/a"]
    blob = nso.quant_pack(w, bs, qt, st, asym, core)
    assert np.array_equal(blob, G[name + "/blob"])
    q, sc, zp = nso.unpack_canonical(blob)
    assert np.array_equal(q, G[name + "/codes"])
    if asym:
        assert np.array_equal(zp, G[name + "/zps"])
    assert np.array_equal(nso.unpack_fp32(blob).view(np.uint32), G[name + "/dequant"].view(np.uint32))
    assert np.array_equal(nso.gemm_f64(a, blob), G[name + "/c_f64"])
    # serialized size is a pure function of the header
    assert nso.pack_size(n, k, bs, qt, st, asym, core) == blob.size


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(L, pkg, nso, name):
    qt, st, asym, core, bs, n, k = _meta(name)
    w, a = G[name + "/w"], G[name + "/a"]
    gold = G[name + "/blob"]
    L.ns_set_pack_core(core)
    try:
        comp = {0: pkg.COMP_F32, 1: pkg.COMP_F32, 2: pkg.COMP_BF16, 3: pkg.COMP_F16}.get(core, pkg.COMP_INT8)
        size = L.ns_BTLAGemmPackBSize(n, k, bs & 0xFFFFFFFFFFFFFFFF, qt, st, asym, comp, None)
        assert size == gold.size
        blob = nso.aligned_bytes(size)
        assert L.ns_BTLAGemmQuantPackB(nso.ptr(blob), nso.ptr(w), n, k, k, bs & 0xFFFFFFFFFFFFFFFF, qt, st, asym, comp, True, None)
    finally:
        L.ns_set_pack_core(pkg.CORE_AUTO)
    assert np.array_equal(blob, gold), "GPU quantize+pack differs from the golden blob"
    if not FWD_OK(qt):
        return
    gb = nso.aligned_bytes(gold.size)
    gb[:] = gold
    deq = np.zeros((k, n), np.float32)
    L.bestla_unpackweight_fp32(nso.ptr(gb), n, k, nso.ptr(deq), n)
    assert np.array_equal(deq.view(np.uint32), G[name + "/dequant"].view(np.uint32))
    out = np.zeros((a.shape[0], n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(gb), nso.ptr(out), a.shape[0], n, k, k, n, None)
    assert nso.rel_l2(out, G[name + "/c_f64"]) < 1e-3
