#!/usr/bin/env python3
"""Weight-streaming kernel vs M (1..64) on a 14336x4096 weight, NF4 g128 and int4 g32: GB/s of the weight stream,
with and without the fp16 activation shadow."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
n, k = [int(x) for x in os.environ.get("SWEEP_SHAPE", "14336x4096").split("x")]
MS = [int(x) for x in os.environ.get("SWEEP_MS", "1,2,4,8,16,32,64").split(",")]
FMTS = (("nf4_g128", pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16), ("int4_g32", pkg.S4, pkg.BF16, 32, pkg.COMP_INT8))
if os.environ.get("SWEEP_INT4_ONLY"):
    FMTS = FMTS[1:]
for name, qt, sdt, bs, comp in FMTS:
    ws = []
    for i in range(3):
        w = torch.randn((n, k), device="cuda") * 0.02
        size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, sdt, False, comp, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, bs, qt, sdt, False, comp, True, st))
        ws.append(pkg.Weight.from_device_blob(blob.data_ptr(), size, st))
    torch.cuda.synchronize()
    for m in MS:
        a = torch.randn((m, k), device="cuda"); a16 = a.half()
        c = torch.empty((m, n), device="cuda")
        for shadow in (True, False):
            def f(s):
                for wt in ws:
                    pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr() if shadow else None, wt.h, c.data_ptr(), None,
                                                        m, k, n, 0, None, 0, s))
            for _ in range(3): f(st)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                f(C.c_void_p(torch.cuda.current_stream().cuda_stream))
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g.replay()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 60
            res["%s m=%d %s" % (name, m, "a16" if shadow else "f32")] = [round(us, 2), round(ws[0].stream_bytes / us / 1e3)]
print(json.dumps({"shape": [n, k], "smallm_max": os.environ.get("NS_SMALLM_MAX", "default"), "us_GBps": res}))
