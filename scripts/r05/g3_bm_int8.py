#!/usr/bin/env python3
"""gemm3_kernel, int8 and int4 weights at M = 2048: workgroup tile by tuning (g3_bm 0 = automatic, 128, 256 = 2 x 2 waves of 128 x 64, 257 = tall 256 x 32 wave tiles)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
m = 2048
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(run, warm=15, reps=30):
    for _ in range(warm): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for qname, qt in (("int8", pkg.S8), ("int4", pkg.S4)):
    for n, k in ((4096, 4096), (11008, 4096), (4096, 11008)):
        w = torch.randn((n, k), device="cuda") * k ** -0.5
        size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
        wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
        a = torch.randn((m, k), device="cuda"); a16 = a.half()
        c = torch.empty((m, n), device="cuda"); c16 = torch.empty((m, n), device="cuda", dtype=torch.float16)
        row = {"weights": qname, "shape": "%dx%d" % (n, k)}
        for rnd in range(2):
            for bm in (0, 128, 256, 257):
                L.ns_hip_set_tuning(b"g3_bm", bm)
                ms = timed(lambda: pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr(), wt.h, c.data_ptr(), c16.data_ptr(), m, k, n, 0, None, 0, st)))
                row.setdefault("bm%d" % bm, []).append(round(2.0 * m * n * k / ms / 1e9))
        L.ns_hip_set_tuning(b"g3_bm", 0)
        print(json.dumps(row), flush=True)
        del w, blob, a, a16, c, c16, wt
