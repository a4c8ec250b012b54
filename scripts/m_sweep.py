#!/usr/bin/env python3
"""Forward time vs M on a 4096x4096 and a 11008x4096 int4 g32 weight: where the weight-streaming kernel (M <= NS_SMALLM_MAX)
hands over to the tiled GEMM."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
if os.environ.get("NS_SWEEP_G3MIN"):  # A-B: the third-generation tiled kernel from this many rows (default 192)
    L.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
    L.ns_hip_set_tuning(b"g3_min_m", int(os.environ["NS_SWEEP_G3MIN"]))
res = {}
for n, k in [tuple(int(v) for v in sh.split('x')) for sh in os.environ.get("NS_SWEEP_SHAPES", "4096x4096,11008x4096").split(',')]:
    ws = []
    for i in range(4):
        w = torch.randn((n, k), device="cuda") * 0.02
        size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
        ws.append(pkg.Weight.from_device_blob(blob.data_ptr(), size, st))
    torch.cuda.synchronize()
    for m in [int(x) for x in os.environ.get("NS_SWEEP_MS", "8,16,17,32,33,48,64,65,96,128,160,191,192,256,384,512").split(",")]:
        a = torch.randn((m, k), device="cuda"); a16 = a.half()
        c = torch.empty((m, n), device="cuda")
        def f():
            for wt in ws:
                pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr(), wt.h, c.data_ptr(), None, m, k, n, 0, None, 0, st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res["%dx%d m=%d" % (n, k, m)] = round(e0.elapsed_time(e1) * 1e3 / 40, 2)
print(json.dumps(res))
