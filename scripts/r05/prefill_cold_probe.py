#!/usr/bin/env python3
"""VERDICT r04 #1 asked where the int8-weight prefill leg's cold reading (740 TFLOPS after two warm-up passes) against its steady
898 comes from: clocks or address translation?  The same leg (bench.py prefill_tflops_int8w's seven GEMMs) timed 4 passes at a time
  (a) right after the weights were made and the GPU sat idle for 0.5 s            -> what bench.py calls "cold"
  (b) again after 0.5 s idle, but with 150 ms of unrelated dense fp16 matmuls (no byte of the leg's weights or buffers) in front
  (c) pass group after pass group, 4 passes each, until 100 groups have run        -> the climb to the steady state
  (b2) like (b) with 150 ms of 1 GiB device-to-device copies (HBM-bound) in front
If (b) is at the steady level, nothing about the leg's own memory (TLB, caches) is cold in (a): the chip's clock is."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.cuda.set_device(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
m, d, ff, grp = 2048, 4096, 11008, 32
ws = []
for i, (n, k) in enumerate([(d, d)] * 4 + [(ff, d)] * 2 + [(d, ff)]):
    g = torch.Generator(device="cuda").manual_seed(4242 + i)
    w = torch.randn((n, k), generator=g, device="cuda") * (k ** -0.5)
    size = L.ns_BTLAGemmPackBSize(n, k, grp, pkg.S8, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(blob.data_ptr(), w.data_ptr(), n, k, k, grp, pkg.S8, pkg.BF16, False, pkg.COMP_INT8, True, st))
    ws.append(pkg.Weight.from_device_blob(blob.data_ptr(), size, st))
    torch.cuda.synchronize()
    del w, blob
a_d = torch.randn((m, d), device="cuda"); a_d16 = a_d.half()
out_big = torch.empty((m, ff), device="cuda"); out_big16 = torch.empty((m, ff), device="cuda", dtype=torch.float16)
qkv = torch.empty((3, m, d), device="cuda"); qkv16 = torch.empty((3, m, d), device="cuda", dtype=torch.float16)
run = bench._prefill_layer_run(pkg, L, st, m, d, d, a_d, a_d16, ws[0], ws[1], ws[2], ws[3], ws[4], ws[5], ws[6], qkv, qkv16, out_big, out_big16)
flops = sum(2.0 * m * w.n * w.k for w in ws)


def group(passes=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(passes):
        run()
    e1.record(); torch.cuda.synchronize()
    return round(flops / (e0.elapsed_time(e1) / passes) / 1e9, 1)


def clock():
    try:
        return torch.cuda.clock_rate()
    except Exception:  # noqa: BLE001
        return None


out = {}
torch.cuda.synchronize(); time.sleep(0.5)
run(); run()
out["a_cold_after_idle_2_warmup_then_4_passes"] = group()
torch.cuda.synchronize(); time.sleep(0.5)
x = torch.randn((4096, 4096), device="cuda", dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < 0.15:
    for _ in range(20):
        torch.mm(x, x)
    torch.cuda.synchronize()
out["clock_MHz_after_unrelated_load"] = clock()
run(); run()
out["b_after_150ms_of_unrelated_matmuls_2_warmup_then_4_passes"] = group()
torch.cuda.synchronize(); time.sleep(0.5)
# (b2) the same with an HBM-bound unrelated load: 150 ms of 1 GiB device-to-device copies
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
t0 = time.time()
while time.time() - t0 < 0.15:
    for _ in range(4):
        dst.copy_(src)
    torch.cuda.synchronize()
out["clock_MHz_after_unrelated_copies"] = clock()
run(); run()
out["b2_after_150ms_of_unrelated_HBM_copies_2_warmup_then_4_passes"] = group()
del src, dst
torch.cuda.synchronize(); time.sleep(0.5)
out["clock_MHz_after_idle"] = clock()
climb = [group() for _ in range(100)]
out["c_consecutive_groups_of_4_passes"] = climb[:12] + ["..."] + climb[-4:]
out["clock_MHz_steady"] = clock()
print(json.dumps(out))
