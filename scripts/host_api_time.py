#!/usr/bin/env python3
"""Wall time of the reference-surface (host pointer) entry points per call, PCIe included."""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()


class _Buf:  # pointer / aligned-buffer helpers (no dependency on the test oracle)
    @staticmethod
    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def aligned_bytes(nbytes, align=64):
        raw = np.zeros(nbytes + align, np.uint8)
        off = (-raw.ctypes.data) % align
        return raw[off:off + nbytes]


nso = _Buf
rng = np.random.default_rng(1)
def mk(n, k):
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = nso.aligned_bytes(size)
    assert L.ns_BTLAGemmQuantPackB(nso.ptr(blob), nso.ptr(w), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, None)
    return blob
d, ff = 4096, 11008
w1, w2, w3 = mk(ff, d), mk(d, ff), mk(ff, d)
a = rng.standard_normal((1, d)).astype(np.float32)
t1 = np.zeros((1, ff), np.float32); t2 = np.zeros((1, ff), np.float32); o = np.zeros((1, d), np.float32)
def tm(f, n=200):
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return round((time.perf_counter() - t0) * 1e6 / n, 1)
# pointers are taken once: building a ctypes pointer from a numpy array costs tens of microseconds in Python
pa, p1, p2, p3, pt1, pt2, po = (nso.ptr(x) for x in (a, w1, w2, w3, t1, t2, o))
print("host-pointer API, us per call (M = 1, Llama-2-7B FFN shapes, includes PCIe and the synchronisation):")
print("  bestla_fusion_FFN_SiLu_f32f32_forward", tm(lambda: L.bestla_fusion_FFN_SiLu_f32f32_forward(pa, p1, p2, p3, pt1, pt2, po, 1, d, ff, d, None)))
print("  bestla_f32f32_forward 4096->11008     ", tm(lambda: L.bestla_f32f32_forward(pa, p1, pt1, 1, ff, d, d, ff, None)))
wq, wk, wv = mk(d, d), mk(d, d), mk(d, d)
qkv = np.zeros((3, d), np.float32)
pq, pk, pv, pqkv = (nso.ptr(x) for x in (wq, wk, wv, qkv))
print("  bestla_fusion_QKV_f32f32_forward      ", tm(lambda: L.bestla_fusion_QKV_f32f32_forward(pa, pq, pk, pv, pqkv, 1, d, d, d, d, None)))
