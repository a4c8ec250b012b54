import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); nso = ge.load_oracle(); L = pkg.lib()
rng = np.random.default_rng(1)
n, k, bs = 100, 160, 32
w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
L.ns_set_pack_core(4)
size = L.ns_BTLAGemmPackBSize(n, k, bs, pkg.S4, pkg.F32, False, pkg.COMP_INT8, None)
blob = nso.aligned_bytes(size)
assert L.ns_BTLAGemmQuantPackB(nso.ptr(blob), nso.ptr(w), n, k, k, bs, pkg.S4, pkg.F32, False, pkg.COMP_INT8, True, None)
ref = nso.quant_pack(w, bs, nso.S4, nso.F32, False, 4)
bi = nso.parse(ref)
r1 = blob[bi.red_off:bi.red_off + bi.red_bytes].view(np.uint16).reshape(-1, bi.cstep)
r2 = ref[bi.red_off:bi.red_off + bi.red_bytes].view(np.uint16).reshape(-1, bi.cstep)
q, sc, zp = nso.unpack_canonical(ref)
f = lambda h: np.array([h], np.uint32).__lshift__(16).view(np.float32)[0]
for kb, c in np.argwhere(r1 != r2)[:10]:
    prods = (q[kb*bs:(kb+1)*bs, c].astype(np.float32) * sc[kb, c]).astype(np.float32)
    seq = np.float32(0)
    for p in prods: seq = np.float32(seq + p)
    fma = np.float64(0)
    t = np.float32(0)
    for qq in q[kb*bs:(kb+1)*bs, c]:
        t = np.float32(np.float64(t) + np.float64(np.float32(qq)) * np.float64(sc[kb, c]))  # fused
    print(kb, c, "gpu", hex(r1[kb, c]), f(int(r1[kb, c])), "ref", hex(r2[kb, c]), f(int(r2[kb, c])), "seq32", seq, hex(nso.lib().nso_f32_to_bf16(float(seq))), "fma", t, hex(nso.lib().nso_f32_to_bf16(float(t))), "exact", float(np.sum(q[kb*bs:(kb+1)*bs, c].astype(np.float64)) * sc[kb, c]))
