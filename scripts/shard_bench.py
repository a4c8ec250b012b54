#!/usr/bin/env python3
"""Per-rank decode GEMV times of the Llama-2-7B tensor-parallel shards (world = 1, 2, 4, 8) on one GPU: which small-M
kernel handles the narrow shards better.  Run twice: NS_DECODE_KERNEL unset / =1."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
d, ff = 4096, 11008


def make(n, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt


def time_us(body, reps=20):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
NL = 4  # distinct layers streamed back to back so nothing is cache resident
for world in (1, 2, 4, 8):
    dl, ffl = d // world, ff // world
    layers = [dict(q=make(dl, d, 1 + 10 * i), k=make(dl, d, 2 + 10 * i), v=make(dl, d, 3 + 10 * i), o=make(d, dl, 4 + 10 * i),
                   w1=make(ffl, d, 5 + 10 * i), w3=make(ffl, d, 6 + 10 * i), w2=make(d, ffl, 7 + 10 * i)) for i in range(NL)]
    x = torch.randn(1, d, device="cuda"); xh = x.half()
    qkv = torch.empty(3, dl, device="cuda"); qkvh = torch.empty(3, dl, device="cuda", dtype=torch.float16)
    att = torch.empty(1, d, device="cuda"); t2 = torch.empty(1, ffl, device="cuda"); t2h = t2.half()
    out = torch.empty(1, d, device="cuda")
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ops = {
        "qkv": lambda lw: pkg.check(L.ns_hip_fusion_qkv_forward_h(x.data_ptr(), xh.data_ptr(), lw["q"].h, lw["k"].h, lw["v"].h, qkv.data_ptr(), qkvh.data_ptr(), 1, d, dl, st())),
        "wo": lambda lw: pkg.check(L.ns_hip_f32f32_forward_h(qkv.data_ptr(), qkvh.data_ptr(), lw["o"].h, att.data_ptr(), None, 1, dl, d, 0, None, 0, st())),
        "gateup": lambda lw: pkg.check(L.ns_hip_fusion_ffn3_gateup_h(x.data_ptr(), xh.data_ptr(), lw["w1"].h, lw["w3"].h, None, t2.data_ptr(), t2h.data_ptr(), 1, pkg.EPI_SILU, st())),
        "down": lambda lw: pkg.check(L.ns_hip_f32f32_forward_h(t2.data_ptr(), t2h.data_ptr(), lw["w2"].h, out.data_ptr(), None, 1, ffl, d, 0, None, 0, st())),
    }
    r = {}
    for name, f in ops.items():
        r[name] = round(time_us(lambda: [f(lw) for lw in layers]) / NL, 2)
    r["layer_sum"] = round(sum(r.values()), 2)
    res["tp%d" % world] = r
    for lw in layers:
        for w in lw.values():
            w.free()
print(json.dumps(res))
