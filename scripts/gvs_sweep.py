#!/usr/bin/env python3
"""Timing sweep of gemvs_kernel's decomposition knobs (ns_hip_set_tuning gvs / gvs_slices / gvs_waves / gvs_grid) on the
shapes of BASELINE configs 4 (Mistral-7B NF4 g128, 8 rows), 2 at 2..16 rows (Llama-2-7B int4 g32) and 5 (70B TP-8 shards,
1 row).  Each shape: `nrep` different weights streamed back to back inside one HIP graph, HIP-event timed.
usage: gvs_sweep.py [c4|c2|c5|all]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def make(n, k, qt, st_dt, bs, comp, seed):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, False, comp, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, bs, qt, st_dt, False, comp, True, st))
    wt = pkg.Weight.from_device_blob(blob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt, blob


def time_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def tune(**kv):
    for k in ("gvs_slices", "gvs_waves", "gvs_grid"):
        L.ns_hip_set_tuning(k.encode(), kv.get(k, 0))
    L.ns_hip_set_tuning(b"gvs", kv.get("gvs", 1))


def shape_sweep(tag, n, k, m, qt, st_dt, bs, comp, fused, settings, nrep=6):
    ws = [make(n, k, qt, st_dt, bs, comp, 7 + i) for i in range(nrep * (2 if fused else 1))]
    a = torch.randn((m, k), device="cuda"); ah = a.half()
    c = torch.empty((m, n), device="cuda"); c2 = torch.empty((m, n), device="cuda")
    byt = ws[0][0].stream_bytes * (2 if fused else 1)

    def fn(s=None):
        s = s or C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if fused:
            for i in range(nrep):
                pkg.check(L.ns_hip_fusion_ffn3_gateup_h(a.data_ptr(), ah.data_ptr(), ws[2 * i][0].h, ws[2 * i + 1][0].h, c2.data_ptr(),
                                                        c.data_ptr(), None, m, pkg.EPI_SILU, s))
        else:
            for wt, _ in ws:
                pkg.check(L.ns_hip_f32f32_forward_h(a.data_ptr(), ah.data_ptr(), wt.h, c.data_ptr(), None, m, k, n, pkg.EPI_NONE, None, 0, s))
    row = {}
    for name, kv in settings:
        tune(**kv)
        us = time_us(fn) / nrep
        row[name] = round(us, 2)
    tune()
    best = min(row, key=row.get)
    print("%-34s m=%-2d %s  | best %s %.2f us = %.0f GB/s" % (tag, m, " ".join("%s=%.2f" % kv for kv in row.items()), best, row[best],
                                                            byt / row[best] / 1e3), flush=True)
    for wt, _ in ws:
        wt.free()
    return row


res = {}
OFF = ("off", {"gvs": 0})
AUTO = ("auto", {})
if which in ("c4", "all"):
    F4 = (pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16)
    wv = [("w8", {"gvs_waves": 8}), ("w11", {"gvs_waves": 11}), ("w15", {"gvs_waves": 15})]
    res["c4_wq"] = shape_sweep("c4 wq 4096x4096 nf4", 4096, 4096, 8, *F4, False, [OFF, AUTO] + wv + [("s2", {"gvs_slices": 2}), ("s4", {"gvs_slices": 4})])
    res["c4_wk"] = shape_sweep("c4 wk 1024x4096 nf4", 1024, 4096, 8, *F4, False, [OFF, AUTO, ("s2", {"gvs_slices": 2}), ("s4", {"gvs_slices": 4})])
    res["c4_gu"] = shape_sweep("c4 gate+up 14336x4096 nf4", 14336, 4096, 8, *F4, True, [OFF, AUTO] + wv + [("g512", {"gvs_grid": 512})], nrep=3)
    res["c4_w2"] = shape_sweep("c4 w2 4096x14336 nf4", 4096, 14336, 8, *F4, False,
                               [OFF, AUTO, ("s2", {"gvs_slices": 2}), ("s4", {"gvs_slices": 4}), ("s8", {"gvs_slices": 8}), ("s4w7", {"gvs_slices": 4, "gvs_waves": 7}),
                                ("s8w7", {"gvs_slices": 8, "gvs_waves": 7}), ("s16", {"gvs_slices": 16})], nrep=3)
    res["c4_head"] = shape_sweep("c4 lm_head 32000x4096 nf4", 32000, 4096, 8, *F4, False, [OFF, AUTO] + wv, nrep=2)
if which in ("c2", "all"):
    I4 = (pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)
    for m in (2, 4, 8, 16):
        res["c2_wo_m%d" % m] = shape_sweep("c2 wo 4096x4096 int4", 4096, 4096, m, *I4, False, [OFF, AUTO, ("w8", {"gvs_waves": 8}), ("w15", {"gvs_waves": 15})])
        res["c2_gu_m%d" % m] = shape_sweep("c2 gate+up 11008x4096 int4", 11008, 4096, m, *I4, True, [OFF, AUTO, ("w8", {"gvs_waves": 8}), ("w15", {"gvs_waves": 15})], nrep=3)
        res["c2_w2_m%d" % m] = shape_sweep("c2 w2 4096x11008 int4", 4096, 11008, m, *I4, False,
                                           [OFF, AUTO, ("s1", {"gvs_slices": 1}), ("s2", {"gvs_slices": 2}), ("s4", {"gvs_slices": 4})], nrep=3)
    res["c2_gu_m1"] = shape_sweep("c2 gate+up 11008x4096 int4", 11008, 4096, 1, *I4, True,
                                  [OFF, ("on", {"gvs": 2}), ("w8", {"gvs": 2, "gvs_waves": 8}), ("w15", {"gvs": 2, "gvs_waves": 15}), ("g512", {"gvs": 2, "gvs_grid": 512})], nrep=3)
    res["c2_wo_m1"] = shape_sweep("c2 wo 4096x4096 int4", 4096, 4096, 1, *I4, False, [OFF, ("on", {"gvs": 2}), ("w15", {"gvs": 2, "gvs_waves": 15})])
    res["c2_w2_m1"] = shape_sweep("c2 w2 4096x11008 int4", 4096, 11008, 1, *I4, False, [OFF, ("on", {"gvs": 2}), ("w15", {"gvs": 2, "gvs_waves": 15})], nrep=3)
if which in ("c5", "all"):
    I4 = (pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)
    on = lambda **kv: dict(gvs=2, **kv)
    sl = [("s1", on(gvs_slices=1)), ("s2", on(gvs_slices=2)), ("s4", on(gvs_slices=4)), ("s8", on(gvs_slices=8))]
    res["c5_wq"] = shape_sweep("c5 wq/8 1024x8192", 1024, 8192, 1, *I4, False, [OFF, ("on", on())] + sl)
    res["c5_wk"] = shape_sweep("c5 wk/8 128x8192", 128, 8192, 1, *I4, False, [OFF, ("on", on())] + sl + [("s16", on(gvs_slices=16))])
    res["c5_wo"] = shape_sweep("c5 wo/8 8192x1024", 8192, 1024, 1, *I4, False, [OFF, ("on", on()), ("w8", on(gvs_waves=8)), ("s2", on(gvs_slices=2))])
    res["c5_gu"] = shape_sweep("c5 gate+up/8 3584x8192", 3584, 8192, 1, *I4, True, [OFF, ("on", on()), ("w8", on(gvs_waves=8)), ("w13", on(gvs_waves=13)), ("s2", on(gvs_slices=2))], nrep=3)
    res["c5_w2"] = shape_sweep("c5 w2/8 8192x3584", 8192, 3584, 1, *I4, False, [OFF, ("on", on()), ("w7", on(gvs_waves=7)), ("w14", on(gvs_waves=14)), ("s2", on(gvs_slices=2))], nrep=3)
print(json.dumps(res))
