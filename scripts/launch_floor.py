#!/usr/bin/env python3
"""What one dependent launch costs inside a replayed HIP graph: a near-empty kernel (ns_hip_add on 16 floats), the decode kernel on
a weight of ONE tile (16 x 4096: 2 KiB x 32 records) and on growing weights — the fixed part of the per-launch model of DESIGN.md
section 4.2c.  64 launches per graph, each consuming its predecessor's output where shapes allow; HIP-event time per launch."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()


def make(n, k, seed):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt


def time_us(fn, n_launch, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n_launch


res = {}
NL = 64
x = torch.zeros(16, device="cuda"); y = torch.ones(16, device="cuda")
L.ns_hip_add.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]


def empty():
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(NL):
        pkg.check(L.ns_hip_add(1, 16, x.data_ptr(), y.data_ptr(), 16, x.data_ptr(), s))
res["near_empty_kernel_us"] = round(time_us(empty, NL), 3)
print("near-empty kernel: %.2f us per dependent launch" % res["near_empty_kernel_us"], flush=True)
k = 4096
for n in (16, 256, 1024, 4096, 11008, 32000):
    ws = [make(n, k, 3 + i) for i in range(4)]
    a = torch.randn((1, k), device="cuda"); ah = a.half()
    c = torch.empty((1, n), device="cuda")

    def fn():
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i in range(NL):
            pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), ah.data_ptr(), ws[i % 4].h, c.data_ptr(), None, 1, k, n, pkg.EPI_NONE, None, 0, s))
    us = time_us(fn, NL)
    mb = ws[0].stream_bytes / 1e6
    res["gemv_%dx%d" % (n, k)] = {"us": round(us, 2), "MB": round(mb, 2), "tiles": (n + 15) // 16}
    print("decode kernel %6d x %d (%5d tiles, %6.2f MB): %.2f us per launch" % (n, k, (n + 15) // 16, mb, us), flush=True)
    for w in ws:
        w.free()
print(json.dumps(res))
