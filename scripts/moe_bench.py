#!/usr/bin/env python3
"""Mixture-of-experts FFN at Mixtral-8x7B shapes (8 experts of 14336 x 4096 gate / up and 4096 x 14336 down, 2 experts per token,
int4 g32 bf16): the three expert-indexed matmuls of ffn_id_silu (ne_layers.c:8053-8170: gate with SiLU, up with Mul, down) per
selected expert, ids on the device, one HIP graph; parity of one token's rows against the oracle's fp64 product.  Prints us per
MoE FFN layer, the HBM rate over the expert weights a token really touches, and tokens/s of 32 such layers.
usage: moe_bench.py [tokens=1]"""
import ctypes as C, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib(); nso = ge.load_oracle()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_as, d, ff, topk = 8, 4096, 14336, 2
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make(n, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * k ** -0.5
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    b = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(b.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    wt = pkg.Weight.from_device_blob(b.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt, b


def group(n, k, seed0):
    ws = [make(n, k, seed0 + i) for i in range(n_as)]
    arr = (C.c_void_p * n_as)(*[w[0].h for w in ws])
    g = L.ns_hip_expert_group_create(arr, n_as)
    assert g, pkg.last_error()
    return ws, g


nlay = 2   # two different layers in the graph: 2 x 3 x 8 experts x 33 MB = 1.6 GB resident, a token touches 2 of 8
layers = [(group(ff, d, 100 * i), group(ff, d, 100 * i + 20), group(d, ff, 100 * i + 40)) for i in range(nlay)]
rng = np.random.default_rng(1)
a = torch.randn((m, d), device="cuda")
ids_np = np.stack([rng.permutation(n_as)[:topk] for _ in range(m)]).astype(np.int32)
ids = torch.from_numpy(ids_np).cuda()
gate, act, y = torch.empty((m, ff), device="cuda"), torch.empty((m, ff), device="cuda"), torch.empty((topk, m, d), device="cuda")


def ffn(layer, s):
    (wg, gg), (wu, gu), (wd, gd) = layer
    for sel in range(topk):
        pkg.check(L.ns_hip_mul_mat_id(a.data_ptr(), ids.data_ptr(), topk, sel, gg, gate.data_ptr(), m, d, ff, pkg.EPI_SILU, None, 0, s))
        pkg.check(L.ns_hip_mul_mat_id(a.data_ptr(), ids.data_ptr(), topk, sel, gu, act.data_ptr(), m, d, ff, pkg.EPI_MUL, gate.data_ptr(), ff, s))
        pkg.check(L.ns_hip_mul_mat_id(act.data_ptr(), ids.data_ptr(), topk, sel, gd, y[sel].data_ptr(), m, ff, d, pkg.EPI_NONE, None, 0, s))


def chain(s):
    for layer in layers:
        ffn(layer, s)


for _ in range(3):
    chain(st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if m >= 32:
    # prefill size (round 5): the grouped form reads the ids on the host, so it is timed as plain calls (NS_MOE_GROUPED_ROWS=0: the per-row kernels)
    reps = 3
    e0.record()
    for _ in range(reps):
        chain(st)
    e1.record(); torch.cuda.synchronize()
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    reps = 30
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps / nlay
# parity of the last layer's last selection (token 0): silu(a Wg) * (a Wu) then Wd, fp64 over the oracle's blobs
(wg, _), (wu, _), (wd, _) = layers[-1]
e = int(ids_np[0, topk - 1])
host = lambda t: (lambda b: (b.__setitem__(slice(None), t.cpu().numpy()), b)[1])(nso.aligned_bytes(t.numel()))
a0 = a[:1].cpu().numpy()
hg = nso.gemm_f64(a0, host(wg[e][1])).astype(np.float64)
hu = nso.gemm_f64(a0, host(wu[e][1])).astype(np.float64)
t = (hg / (1.0 + np.exp(-hg)) * hu).astype(np.float32)
ref = nso.gemm_f64(t, host(wd[e][1]))
par = nso.rel_l2(y[topk - 1][:1].cpu().numpy(), ref)
touched = sum(w[0][e2][0].stream_bytes for w in layers[0] for e2 in set(ids_np.reshape(-1).tolist())) if m > 1 else \
    sum(layers[0][j][0][int(ex)][0].stream_bytes for j in range(3) for ex in ids_np[0])
print(json.dumps({"what": "Mixtral-8x7B-shaped MoE FFN (8 x {14336x4096 gate, up; 4096x14336 down}, top-2, int4 g32 bf16), %d token(s)" % m,
                  "us_per_moe_ffn_layer": round(us, 2), "expert_weight_bytes_touched_per_layer": int(touched),
                  "hbm_GBps_over_touched_weights": round(touched / us / 1e3, 1), "tokens_per_s_of_32_moe_ffn_layers": round(m * 1e6 / (32 * us), 1),
                  "parity_rel_l2_vs_fp64_oracle_token0": float("%.3g" % par), "launches_per_layer": 3 * topk,
                  "grouped_by_expert": m >= 32 and os.environ.get("NS_MOE_GROUPED_ROWS", "32") != "0",
                  "tflops": round(m * topk * 3 * 2.0 * d * ff / us / 1e6, 1)}))
