"""Host-pointer entry points keep one device copy per blob pointer (validated by a content fingerprint).  With
NS_CACHE_MAX_BYTES the least recently used copies are dropped and re-uploaded on their next use; results do not change."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/oracle")
import __graft_entry__ as ge, nso
pkg = ge.load_package(); L = pkg.lib()
rng = np.random.default_rng(3)
n, k = 512, 512
blobs = [nso.quant_pack((rng.standard_normal((n, k)) * 0.05).astype(np.float32), 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
         for _ in range(4)]
a = rng.standard_normal((2, k)).astype(np.float32)
refs = [nso.gemm_f64(a, b) for b in blobs]
for rep in range(3):
    for i in (0, 1, 2, 3, 1, 0):
        out = np.zeros((2, n), np.float32)
        L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blobs[i]), nso.ptr(out), 2, n, k, k, n, None)
        assert nso.rel_l2(out, refs[i]) < 1e-3, (rep, i)
# a blob rewritten in place (same pointer, new content) must not be served from the old device copy
blobs[0][:] = blobs[3]
out = np.zeros((2, n), np.float32)
L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blobs[0]), nso.ptr(out), 2, n, k, k, n, None)
assert nso.rel_l2(out, refs[3]) < 1e-3
print("CACHE_OK")
''' % (ROOT, ROOT)


@pytest.mark.parametrize("cap", ["", "300000", "1"])
def test_weight_cache_cap(cap):
    env = dict(os.environ)
    if cap:
        env["NS_CACHE_MAX_BYTES"] = cap  # one 512 x 512 int4 weight is ~150 KB on the device: 2 fit / none fits
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "CACHE_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
