#!/usr/bin/env python3
"""Decode (one row) on the 1-3 / 5-7 bit formats: the widened nibble / byte records (ns_hip_set_tuning "planes" 0) against the
native bit-plane records (1, default) — Llama-2-7B gate/up and down shapes, several different weights back to back in one HIP
graph, HIP-event time per launch.  usage: planes_bench.py [bits ...]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
L.ns_hip_set_tuning(b"planes_load", 1)  # (off by default: the second copy is built at load)


def make(n, k, bits, seed):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    qt = pkg.INT_TYPES[bits]
    size = L.ns_BTLAGemmPackBSize(n, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, 32, qt, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt


def time_us(fn, reps=20):
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
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
for bits in [int(b) for b in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8]:
    for tag, n, k in (("gate_11008x4096", 11008, 4096), ("down_4096x11008", 4096, 11008)):
        nrep = 6
        ws = [make(n, k, bits, 7 + i) for i in range(nrep)]
        a = torch.randn((1, k), device="cuda"); ah = a.half()
        c = torch.empty((1, n), device="cuda")

        def fn():
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for wt in ws:
                pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), ah.data_ptr(), wt.h, c.data_ptr(), None, 1, k, n, pkg.EPI_NONE, None, 0, s))
        row = {"algorithmic_MB": round(ws[0].stream_bytes / 1e6, 2)}
        for on in (0, 1):
            L.ns_hip_set_tuning(b"planes", on)
            us = time_us(fn) / nrep
            row["native" if on else "widened"] = {"us": round(us, 2), "algorithmic_GBps": round(ws[0].stream_bytes / us / 1e3, 0)}
        row["speedup"] = round(row["widened"]["us"] / row["native"]["us"], 3)
        res["S%d_%s" % (bits, tag)] = row
        print("S%d %-16s %s" % (bits, tag, json.dumps(row)), flush=True)
        for wt in ws:
            wt.free()
L.ns_hip_set_tuning(b"planes", 1)
print(json.dumps(res))
