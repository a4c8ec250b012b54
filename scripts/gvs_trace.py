#!/usr/bin/env python3
"""Intra-kernel timeline of gemvs_kernel (needs a -DNS_TRACE build: FILES="ns_kernels ns_gemvs" scripts/build_variants.sh trace:-DNS_TRACE).
Stamps (100 MHz wall clock) per wave: 0 entry, 1 ring issued, 2 first barrier passed (A landed), 3 A shuffled (f4), 4 first tile
flushed (service wave: first tile's epilogue done), 5 last unit consumed (service: last tile done).  Percentiles over waves, us after
the earliest entry.  usage: NS_LIB_PATH=variants/libns_hip_trace.so gvs_trace.py <shape> [m]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
L.ns_hip_debug_trace_read.argtypes = [C.c_void_p, C.c_size_t]
F4 = (pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16); I4 = (pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)
TAB = {"c4gu": (14336, 4096, 8, F4, True), "c4w2": (4096, 14336, 8, F4, False), "c4wq": (4096, 4096, 8, F4, False),
       "c2gu": (11008, 4096, 8, I4, True), "c2w2": (4096, 11008, 8, I4, False)}
n, k, m, fmt, fused = TAB[sys.argv[1]]
if len(sys.argv) > 2: m = int(sys.argv[2])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = []
for i in range(4):
    g = torch.Generator(device="cuda").manual_seed(7 + i)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, fmt[2], fmt[0], fmt[1], False, fmt[3], None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, fmt[2], fmt[0], fmt[1], False, fmt[3], True, st))
    ws.append((pkg.Weight.from_device_blob(blob.data_ptr(), size, st), blob)); del w
a = torch.randn((m, k), device="cuda"); ah = a.half()
c = torch.empty((m, n), device="cuda"); c2 = torch.empty((m, n), device="cuda")
def run(i):
    if fused:
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(a.data_ptr(), ah.data_ptr(), ws[i][0].h, ws[i + 1][0].h, c2.data_ptr(), c.data_ptr(), None, m, pkg.EPI_SILU, st))
    else:
        pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), ah.data_ptr(), ws[i][0].h, c.data_ptr(), None, m, k, n, pkg.EPI_NONE, None, 0, st))
run(0); torch.cuda.synchronize()
buf = np.zeros(4096 * 16 * 8, np.uint64)
L.ns_hip_debug_trace_read(buf.ctypes.data, buf.nbytes)   # clears
z = torch.empty(1 << 29, dtype=torch.uint8, device="cuda"); z.fill_(1); torch.cuda.synchronize()
if os.environ.get("WARM", "1") == "1":   # steady state: the launch repeated on alternating weights, the last one is read
    for i in range(6):
        run(2 if i % 2 == 0 else 0)
else:
    run(2)
torch.cuda.synchronize()
L.ns_hip_debug_trace_read(buf.ctypes.data, buf.nbytes)
t = buf.reshape(4096, 16, 8).astype(np.int64)
live = t[:, :, 0] > 0
t0 = t[:, :, 0][live].min()
nblk = int(live.any(axis=1).sum()); nwav = int(live[0].sum())
print("gemvs trace %s m=%d: %d workgroups x %d waves (last wave = service)" % (sys.argv[1], m, nblk, nwav))
svc = np.zeros_like(live); svc[:, nwav - 1] = True
for name, mask in (("streaming", live & ~svc), ("service", live & svc)):
    print(" ", name)
    for i, lab in enumerate(["entry", "ring issued", "A landed (barrier 1)", "A shuffled (barrier 2)", "first tile flushed/done", "last unit consumed/tile done"]):
        v = (t[:, :, i][mask & (t[:, :, i] > 0)] - t0) / 100.0
        if v.size:
            print("    %-30s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (lab, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
