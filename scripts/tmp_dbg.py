import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nso
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
L.ns_hip_set_compute_mode(1)
rng = np.random.default_rng(0)
m, n, k, bs = 16, 64, 128, 32
w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
a = rng.standard_normal((m, k)).astype(np.float32)
blob = nso.quant_pack(w, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
q, sc, zp = nso.unpack_canonical(blob)
out = np.full((m, n), 7.0, np.float32)
L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
np.set_printoptions(linewidth=200, precision=4, suppress=True)
print("mode", os.environ.get("NS_I8_DBG"), "nan", int(np.isnan(out).sum()))
print(out[:6, :8])
print("expect: su(col) = sum(q[:32]+8):", (q[:32].astype(int) + 8).sum(0)[:8], " sb:", sc[0, :8])
