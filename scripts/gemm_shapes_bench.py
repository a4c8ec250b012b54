#!/usr/bin/env python3
"""Per-shape prefill GEMM throughput: Llama-2-7B shapes at M rows, int4 g32 (and int8 g32) weights, each shape timed
alone.  Usage: scripts/gemm_shapes_bench.py [M] [int8]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
qt = pkg.S8 if (len(sys.argv) > 2 and sys.argv[2] == "int8") else pkg.S4
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
for n, k in [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]:
    w = torch.randn((n, k), device="cuda") * k ** -0.5
    size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    a = torch.randn((m, k), device="cuda"); a16 = a.half()
    c = torch.empty((m, n), device="cuda"); c16 = torch.empty((m, n), device="cuda", dtype=torch.float16)
    run = lambda: pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr(), wt.h, c.data_ptr(), c16.data_ptr(), m, k, n, 0, None, 0, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    # the same ten launches as one HIP graph (no host in the loop)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(10):
            run()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    gms = e0.elapsed_time(e1) / 30
    res["%dx%d" % (n, k)] = {"us": round(ms * 1e3, 1), "tflops": round(2.0 * m * n * k / ms / 1e9, 1),
                              "graph_us": round(gms * 1e3, 1), "graph_tflops": round(2.0 * m * n * k / gms / 1e9, 1)}
    del w, blob, a, a16, c, c16
print(json.dumps({"m": m, "weights": "int8" if qt == pkg.S8 else "int4", "shapes": res}))
