#!/usr/bin/env python3
"""prefill TFLOPS of the tiled MFMA GEMM on Llama-2-7B shapes (M = 2048): int4 g32 bf16 ("Q4_0") and int8 g32."""
import ctypes as C, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 2048
res = {}
for name, qt in (("int4_g32_bf16", pkg.S4), ("int8_g32_bf16", pkg.S8)):
    tot_t, tot_f = 0.0, 0.0
    for (n, k) in ((4096, 4096), (11008, 4096), (4096, 11008)):
        w = torch.randn((n, k), device="cuda") * 0.02
        comp = pkg.COMP_INT8 if qt == pkg.S4 else pkg.COMP_F32
        size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, comp, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, comp, True, st))
        wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
        a = torch.randn((M, k), device="cuda")
        c = torch.empty((M, n), device="cuda")
        f = lambda: pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c.data_ptr(), M, k, n, 0, None, 0, st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * M * n * k
        # spot check against a torch fp32 matmul on the device-unpacked weights
        res["%s %dx%d" % (name, n, k)] = {"ms": round(ms, 3), "TFLOPS": round(fl / ms / 1e9, 1)}
        tot_t += ms; tot_f += fl
        del wt
    res[name + " weighted"] = round(tot_f / tot_t / 1e9, 1)
print(json.dumps(res, indent=1))
