#!/usr/bin/env python3
"""One-shape probe for kernel-trace runs: `nrep` different weights of one shape streamed back to back in ONE graph replay
(so a rocprofv3 --kernel-trace --stats run gives the kernel's own average duration, free of the replay's fixed cost).
usage: gvs_probe.py <shape> [m]   shape: c4gu | c4w2 | c4wq | c2gu | c2w2 | c5gu | c5wq ; knobs via NS_GVS* environment"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
shape = sys.argv[1]
F4 = (pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16)
I4 = (pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)
TAB = {"c4gu": (14336, 4096, 8, F4, True), "c4w2": (4096, 14336, 8, F4, False), "c4wq": (4096, 4096, 8, F4, False),
       "c4i4gu": (14336, 4096, 8, I4, True), "c4i4w2": (4096, 14336, 8, I4, False),
       "c2gu": (11008, 4096, 8, I4, True), "c2w2": (4096, 11008, 8, I4, False), "c2wo": (4096, 4096, 8, I4, False),
       "c5gu": (3584, 8192, 1, I4, True), "c5wq": (1024, 8192, 1, I4, False), "c5w2": (8192, 3584, 1, I4, False)}
n, k, m, fmt, fused = TAB[shape]
if len(sys.argv) > 2:
    m = int(sys.argv[2])
nrep = int(os.environ.get("NREP", "8"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = []
for i in range(nrep * (2 if fused else 1)):
    g = torch.Generator(device="cuda").manual_seed(7 + i)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, fmt[2], fmt[0], fmt[1], False, fmt[3], None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, fmt[2], fmt[0], fmt[1], False, fmt[3], True, st))
    ws.append((pkg.Weight.from_device_blob(blob.data_ptr(), size, st), blob))
    del w
torch.cuda.synchronize()
a = torch.randn((m, k), device="cuda"); ah = a.half()
c = torch.empty((m, n), device="cuda"); c2 = torch.empty((m, n), device="cuda")


def fn(s):
    if fused:
        for i in range(nrep):
            pkg.check(L.ns_hip_fusion_ffn3_gateup_h(a.data_ptr(), ah.data_ptr(), ws[2 * i][0].h, ws[2 * i + 1][0].h, c2.data_ptr(), c.data_ptr(),
                                                    None, m, pkg.EPI_SILU, s))
    else:
        for wt, _ in ws:
            pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), ah.data_ptr(), wt.h, c.data_ptr(), None, m, k, n, pkg.EPI_NONE, None, 0, s))


for _ in range(2):
    fn(st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn(C.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    g.replay()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps / nrep
byt = ws[0][0].stream_bytes * (2 if fused else 1)
print("PROBE %s m=%d lib=%s knobs=%s: %.2f us per launch (graph of %d, event time) = %.0f GB/s" % (
    shape, m, os.path.basename(os.environ.get("NS_LIB_PATH", "default")), {k: v for k, v in os.environ.items() if k.startswith("NS_GVS")}, us, nrep,
    byt / us / 1e3), flush=True)
