import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); nso = ge.load_oracle(); L = pkg.lib()

def run(qt, asym, core, comp, n, k, bs, st, kind, seed):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32) if kind == "normal" else rng.uniform(-0.5, 0.5, (n, k)).astype(np.float32)
    L.ns_set_pack_core(core)
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st, asym, comp, None)
    blob = nso.aligned_bytes(size)
    assert L.ns_BTLAGemmQuantPackB(nso.ptr(blob), nso.ptr(w), n, k, k, bs, qt, st, asym, comp, True, None), pkg.last_error()
    L.ns_set_pack_core(-1)
    ref = nso.quant_pack(w, bs, qt, st, asym, core)
    if np.array_equal(blob, ref):
        print("OK", qt, asym, core, n, k, bs, kind); return
    q1, s1, z1 = nso.unpack_canonical(blob)
    q2, s2, z2 = nso.unpack_canonical(ref)
    dq = np.argwhere(q1 != q2)
    print("DIFF", qt, asym, core, n, k, bs, kind, "codes differ:", len(dq), "scales differ:", int((s1.view(np.uint32) != s2.view(np.uint32)).sum()), "zp differ", int((z1 != z2).sum()))
    if len(dq):
        print("  first (k,n):", dq[:8].tolist(), "cols:", np.unique(dq[:, 1])[:20], "ks:", np.unique(dq[:, 0])[:20])
        kk, nn = dq[0]
        print("  gpu", q1[kk, nn], "ref", q2[kk, nn], "w", w[nn, kk], "scale", s2[kk // bs, nn])
    bi = nso.parse(ref)
    bad = np.nonzero(blob != ref)[0]
    print("  bytes differ", bad.size, "first", bad[:5], "q", bi.q_off, "s", bi.scale_off, "z", bi.zp_off, "r", bi.red_off)

for kind in ("normal", "uniform"):
    for seed in (1, 2):
        run(pkg.INT_TYPES[5], False, 4, pkg.COMP_INT8, 100, 160, 32, pkg.F32, kind, seed)
        run(pkg.F4_NF4, False, 1, pkg.COMP_F32, 100, 160, 32, pkg.F32, kind, seed)
        run(pkg.INT_TYPES[2], False, 1, pkg.COMP_F32, 100, 160, 32, pkg.F32, kind, seed)
        run(pkg.S8, True, 2, pkg.COMP_BF16, 96, 128, 32, pkg.BF16, kind, seed)
        run(pkg.S4, False, 4, pkg.COMP_INT8, 100, 160, 32, pkg.F32, kind, seed)
