#!/usr/bin/env python3
"""Round-5 probe for the prefill GEMM work: (1) what the dense fp16 library GEMM (torch.matmul -> hipBLASLt) reaches on the
four Llama-2-7B shapes at M rows — the practical ceiling of an fp16 MFMA GEMM on this box, for orientation only, nothing in
the product calls it; (2) gemm3_kernel per shape, int4 and int8 weights, steady state (many back-to-back launches).
NS_G3_DIAG=1 in the environment skips the output stores (what the epilogue traffic costs).
Usage: scripts/r05/prefill_probe.py [M] [reps]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(4096, 4096), (12288, 4096), (11008, 4096), (4096, 11008)]


def timed(run, warm=20, reps=reps):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"m": m, "diag": os.environ.get("NS_G3_DIAG", "0")}
if "--no-lib" not in sys.argv:
    lib = {}
    for n, k in SHAPES:
        a = torch.randn((m, k), device="cuda", dtype=torch.float16)
        w = torch.randn((n, k), device="cuda", dtype=torch.float16) * k ** -0.5
        c = torch.empty((m, n), device="cuda", dtype=torch.float16)
        ms = timed(lambda: torch.matmul(a, w.t(), out=c))
        lib["%dx%d" % (n, k)] = {"us": round(ms * 1e3, 1), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
        del a, w, c
    out["hipblaslt_fp16_dense"] = lib
for qname, qt in (("int4", pkg.S4), ("int8", pkg.S8)):
    res = {}
    for n, k in SHAPES:
        if n == 12288:
            continue
        w = torch.randn((n, k), device="cuda") * k ** -0.5
        size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
        wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
        torch.cuda.synchronize()
        a = torch.randn((m, k), device="cuda"); a16 = a.half()
        c = torch.empty((m, n), device="cuda"); c16 = torch.empty((m, n), device="cuda", dtype=torch.float16)
        both = timed(lambda: pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr(), wt.h, c.data_ptr(), c16.data_ptr(), m, k, n, 0, None, 0, st)))
        f32only = timed(lambda: pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), a16.data_ptr(), wt.h, c.data_ptr(), None, m, k, n, 0, None, 0, st)))
        fl = 2.0 * m * n * k
        res["%dx%d" % (n, k)] = {"us_f32_f16_out": round(both * 1e3, 1), "tflops": round(fl / both / 1e9, 1),
                                  "us_f32_out_only": round(f32only * 1e3, 1), "tflops_f32_only": round(fl / f32only / 1e9, 1)}
        del w, blob, a, a16, c, c16, wt
    # the FFN through the fused entry: gate / up tile pairs (one launch, fp16-only intermediate) + down projection
    d, ff = 4096, 11008
    ws = []
    for n, k in ((ff, d), (ff, d), (d, ff)):
        w = torch.randn((n, k), device="cuda") * k ** -0.5
        size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, None)
        blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
        ws.append(pkg.Weight.from_device_blob(blob.data_ptr(), size, st))
        torch.cuda.synchronize()
        del w, blob
    a = torch.randn((m, d), device="cuda"); a16 = a.half()
    t216 = torch.empty((m, ff), device="cuda", dtype=torch.float16)
    o = torch.empty((m, d), device="cuda"); o16 = torch.empty((m, d), device="cuda", dtype=torch.float16)
    gu = timed(lambda: pkg.check(L.ns_hip_fusion_ffn3_gateup_h(a.data_ptr(), a16.data_ptr(), ws[0].h, ws[1].h, None, None, t216.data_ptr(), m, pkg.EPI_SILU, st)))
    ffn = timed(lambda: pkg.check(L.ns_hip_fusion_ffn3_forward_h(a.data_ptr(), a16.data_ptr(), ws[0].h, ws[2].h, ws[1].h, None, None, None, o.data_ptr(), o16.data_ptr(), m, pkg.EPI_SILU, st)))
    res["gate_up_pairs_fp16_out"] = {"us": round(gu * 1e3, 1), "tflops": round(4.0 * m * d * ff / gu / 1e9, 1)}
    res["ffn_fused_entry"] = {"us": round(ffn * 1e3, 1), "tflops": round(6.0 * m * d * ff / ffn / 1e9, 1)}
    del ws, a, a16, t216, o, o16
    out[qname] = res
print(json.dumps(out))
