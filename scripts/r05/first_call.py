#!/usr/bin/env python3
"""What a process's FIRST call of each kernel family costs (code-object load, scratch allocation) against its second: ms."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib(); nso = ge.load_oracle()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
d, m = 1024, 300
w = (rng.standard_normal((d, d)) * d ** -0.5).astype(np.float32)
blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
t0 = time.time(); wt = pkg.Weight.from_host_blob(nso.ptr(blob), st); torch.cuda.synchronize(); t_load = time.time() - t0
a = torch.randn((m, d), device="cuda"); c = torch.zeros((m, d), device="cuda")
a1 = torch.randn((1, d), device="cuda"); c1 = torch.zeros((1, d), device="cuda")
def timed(f):
    torch.cuda.synchronize(); t = time.time(); f(); torch.cuda.synchronize(); return round((time.time() - t) * 1e3, 2)
res = {"weight_load_ms": round(t_load * 1e3, 2)}
if os.environ.get("NS_FIRST_CALL_WARM", "1") != "0":
    L.ns_hip_warm_up.restype = C.c_int
    t0 = time.time(); L.ns_hip_warm_up(); res["warm_up_ms"] = round((time.time() - t0) * 1e3, 2)
gemm = lambda: pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c.data_ptr(), m, d, d, 0, None, 0, st))
gemv = lambda: pkg.check(L.ns_hip_f32f32_forward(a1.data_ptr(), wt.h, c1.data_ptr(), 1, d, d, 0, None, 0, st))
q = torch.randn((1, 200, 8, 128), device="cuda"); kc = torch.randn((1, 200, 8, 128), device="cuda").half(); o = torch.zeros_like(q)
def attn():
    ar = pkg.attn_args(q.data_ptr(), kc.data_ptr(), kc.data_ptr(), o.data_ptr(), 1, 8, 8, 128, 200, 200, 0.088, pkg.ATTN_CAUSAL)
    pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward(C.byref(ar), st))
for name, f in (("tiled_gemm", gemm), ("decode_gemv", gemv), ("prefill_attention", attn)):
    res[name] = {"first_ms": timed(f), "second_ms": timed(f), "third_ms": timed(f)}
print(json.dumps(res))
