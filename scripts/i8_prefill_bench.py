"""NS_COMPUTE_REF_INT8 at prefill size: M = 2048 rows through a Llama-2-7B weight in the reference's int8-compute
semantics (activation quantizer + the matrix-core kernels of ns_i8ref.hip: "i8_mfma" 1 and 2), event-timed.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
L = pkg.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for name, (n, k, qt, bs) in {"int4_g32_4096x4096": (4096, 4096, pkg.S4, 32), "int4_g128_11008x4096": (11008, 4096, pkg.S4, 128),
                             "int8_g32_4096x4096": (4096, 4096, pkg.S8, 32)}.items():
    m = 2048
    w = torch.randn((n, k), device="cuda") * k ** -0.5
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, bs, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    a = torch.randn((m, k), device="cuda")
    c = torch.zeros((m, n), device="cuda")
    res = {}
    for mode, gen, tile in ((1, 1, 0), (1, 2, 1), (1, 2, 4), (1, 2, 0), (0, 0, 0)):
        L.ns_hip_set_compute_mode(mode)
        if gen:
            L.ns_hip_set_tuning(b"i8_mfma", gen)
            L.ns_hip_set_tuning(b"i8_tile", tile)

        def run():
            pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c.data_ptr(), m, k, n, pkg.EPI_NONE, None, 0, st))
        for _ in range(25):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 25
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        key = ("int8_semantics_kernel%d" % gen + ("_tile%d" % tile if tile else ("_default" if gen == 2 else ""))) if mode else "fp16_default"
        res[key] = {"ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
        if gen:
            res[key]["checksum"] = float(c.double().abs().sum().item())
    L.ns_hip_set_tuning(b"i8_mfma", 2)
    L.ns_hip_set_tuning(b"i8_tile", 0)
    L.ns_hip_set_compute_mode(0)
    out[name] = res
print(json.dumps({"m": 2048, "results": out}))
