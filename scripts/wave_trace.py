#!/usr/bin/env python
"""Intra-kernel timeline of the decode kernel (diagnostics; needs a -DNS_TRACE build, see scripts/build_variants.sh).

Every wave stamps the 100 MHz wall clock at: 0 entry, 1 ring issued, 2 A staged (barrier), 3 first item arrived,
4 last item consumed, 5 reduction barrier, 6 exit.  Prints percentiles relative to the earliest entry stamp.
Run: NS_LIB_PATH=variants/libns_hip_trace.so python scripts/wave_trace.py [shape ...]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
L = pkg.lib()
L.ns_hip_debug_trace_read.argtypes = [C.c_void_p, C.c_size_t]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream

SHAPES = {  # name: (n, k, kind)
    "gateup": (11008, 4096, "dual"),
    "qkv": (4096, 4096, "qkv"),
    "down": (4096, 11008, "plain"),
    "wo": (4096, 4096, "plain"),
    "head": (32000, 4096, "plain"),
}


def make_weight(n, k, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = (torch.randn(n, k, generator=g) * 0.02).to(dev)
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    blob = torch.zeros(size, dtype=torch.uint8, device=dev)
    ptr = blob.data_ptr()
    pkg.check(L.ns_hip_quant_pack_device(ptr, w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    torch.cuda.synchronize()
    return pkg.Weight.from_device_blob(ptr, size, st), blob


def flush():
    z = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    z.fill_(1)
    torch.cuda.synchronize()
    del z


# grid, waves of the decode launch: smallm_kernel (default) or decode_kernel (NS_DECODE_KERNEL=1)
if os.environ.get("NS_DECODE_KERNEL", "0") == "1":
    DIMS = {"gateup": (256, 12), "qkv": (256, 12), "down": (256, 12), "wo": (256, 8)}
else:
    DIMS = {"gateup": (688, 4), "qkv": (768, 4), "down": (256, 16), "wo": (256, 16)}
if os.environ.get("NS_GV_NW"):  # gemv_kernel with a forced wave count
    DIMS = {k: (v[0], int(os.environ["NS_GV_NW"])) for k, v in DIMS.items()}
KARG = os.environ.get("NS_DECODE_KERNEL", "0") == "1"  # slot 7 = "kernargs arrived" (decode_kernel) instead of HW id
LAYERS = 3
ORDER = ["qkv", "wo", "gateup", "down"]


def build_ops():
    """LAYERS x 4 operators with their own weights, inputs hot from the previous launch like in the decode chain."""
    ops = []
    seed = 100
    for layer in range(LAYERS):
        for name in ORDER:
            n, k, kind = SHAPES[name]
            nw = 3 if kind == "qkv" else (2 if kind == "dual" else 1)
            ws = [make_weight(n, k, seed + i) for i in range(nw)]
            seed += nw
            ops.append((name, ws))
    return ops


def cur():
    return torch.cuda.current_stream().cuda_stream


def launcher(name, ws, bufs):
    n, k, kind = SHAPES[name]
    x, xh = bufs[k]
    out, outh = bufs["out"]
    if kind == "dual":
        return lambda: pkg.check(L.ns_hip_fusion_ffn3_gateup_h(x.data_ptr(), xh.data_ptr(), ws[0][0].h, ws[1][0].h, None,
                                                               out.data_ptr(), outh.data_ptr(), 1, pkg.EPI_SILU, cur()))
    if kind == "qkv":
        return lambda: pkg.check(L.ns_hip_fusion_qkv_forward_h(x.data_ptr(), xh.data_ptr(), ws[0][0].h, ws[1][0].h,
                                                               ws[2][0].h, out.data_ptr(), outh.data_ptr(), 1, k, 3 * n, cur()))
    return lambda: pkg.check(L.ns_hip_f32f32_forward_h(x.data_ptr(), xh.data_ptr(), ws[0][0].h, out.data_ptr(),
                                                       outh.data_ptr(), 1, k, n, pkg.EPI_NONE, None, 0, cur()))


def hw_fields(hw):
    xcc = (hw >> 32) & 0xf
    hwid = hw & 0xffffffff
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    return xcc, se, sh, cu


def main(targets):
    ops = build_ops()
    bufs = {}
    for k in (4096, 11008):
        x = torch.randn(1, k, device=dev) * 0.5
        bufs[k] = (x, x.half())
    bufs["out"] = (torch.empty(1, 3 * 11008, device=dev), torch.empty(1, 3 * 11008, device=dev, dtype=torch.float16))
    calls = [(nm, launcher(nm, ws, bufs)) for nm, ws in ops]
    buf = np.zeros((4096, 16, 8), dtype=np.uint64)
    for target in targets:
        # rotate so that the LAST launch of the graph is `target`
        idx = max(i for i, (nm, _) in enumerate(calls) if nm == target)
        seq = calls[idx + 1:] + calls[:idx + 1]
        for _, f in seq:  # warm (code, descriptors)
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _, f in seq:
                f()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        L.ns_hip_debug_trace_read(buf.ctypes.data, buf.nbytes)  # clears
        g.replay()
        torch.cuda.synchronize()
        L.ns_hip_debug_trace_read(buf.ctypes.data, buf.nbytes)
        t = buf.astype(np.int64)
        # stamps of the LAST launch only: earlier launches of the graph leave entries outside its grid
        nb, nwv = DIMS[target]
        live = np.zeros(t.shape[:2], dtype=bool)
        live[:nb, :nwv] = True
        live &= t[:, :, 6] > 0
        report(target, t, live)


def report(name, t, live):
    t0 = t[:, :, 0][live].min()
    rel = (t - t0) * 0.01  # us
    print(f"== {name}: blocks {int(live.any(axis=1).sum())} waves/block {int(live.sum(axis=1).max())} "
          f"kernel span {rel[:, :, 6][live].max():.2f} us")
    names = ["entry", "ring issued", "A staged", "first item", "last item", "reduced", "exit"]
    if KARG:
        v = rel[:, :, 7][live]
        q = np.percentile(v, [0, 10, 50, 90, 100])
        print(f"  {'kernargs in':12s} min {q[0]:6.2f}  p10 {q[1]:6.2f}  p50 {q[2]:6.2f}  p90 {q[3]:6.2f}  max {q[4]:6.2f}")
    for i, nm in enumerate(names):
        v = rel[:, :, i][live]
        q = np.percentile(v, [0, 10, 50, 90, 100])
        print(f"  {nm:12s} min {q[0]:6.2f}  p10 {q[1]:6.2f}  p50 {q[2]:6.2f}  p90 {q[3]:6.2f}  max {q[4]:6.2f}")
    d = {
        "entry->ring issued": rel[:, :, 1] - rel[:, :, 0],
        "ring issued->A staged": rel[:, :, 2] - rel[:, :, 1],
        "A staged->first item": rel[:, :, 3] - rel[:, :, 2],
        "first->last item": rel[:, :, 4] - rel[:, :, 3],
        "last item->exit": rel[:, :, 6] - rel[:, :, 4],
    }
    for nm, v in d.items():
        v = v[live]
        q = np.percentile(v, [10, 50, 90])
        print(f"  dur {nm:22s} p10 {q[0]:6.2f}  p50 {q[1]:6.2f}  p90 {q[2]:6.2f}")
    # per-workgroup spread of the waves' finish times
    fin = np.where(live, rel[:, :, 4], np.nan)
    spread = np.nanmax(fin, axis=1) - np.nanmin(fin, axis=1)
    spread = spread[~np.isnan(spread)]
    print(f"  per-workgroup spread of last-item times: p50 {np.median(spread):.2f} max {spread.max():.2f}")
    if KARG:
        return
    # balance: where did waves run, and when did each CU finish?
    hw = t[:, :, 7].astype(np.uint64)
    xcc, se, sh, cu = hw_fields(hw)
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ids = np.unique(cuid[live])
    waves_per_cu = np.array([int((live & (cuid == c)).sum()) for c in ids])
    done_per_cu = np.array([rel[:, :, 4][live & (cuid == c)].max() for c in ids])
    print(f"  CUs used {ids.size}; waves/CU min {waves_per_cu.min()} p50 {int(np.median(waves_per_cu))} max {waves_per_cu.max()}")
    for wv in np.unique(waves_per_cu):
        sel = waves_per_cu == wv
        print(f"    CUs with {wv:3d} waves: {int(sel.sum()):4d}   last-item time mean {done_per_cu[sel].mean():6.2f} max {done_per_cu[sel].max():6.2f}")
    for x in np.unique(xcc[live]):
        sel = live & (xcc == x)
        print(f"    XCC {int(x)}: waves {int(sel.sum()):5d}  last item p50 {np.median(rel[:, :, 4][sel]):6.2f} max {rel[:, :, 4][sel].max():6.2f}")
    bi = np.arange(t.shape[0])[:, None] * np.ones((1, t.shape[1]), dtype=np.int64)
    for lo in range(0, int(bi[live].max()) + 1, 256):
        sel = live & (bi >= lo) & (bi < lo + 256)
        print(f"    blocks {lo:4d}..{lo + 255:4d}: entry p50 {np.median(rel[:, :, 0][sel]):5.2f}  last item p50 {np.median(rel[:, :, 4][sel]):6.2f} max {rel[:, :, 4][sel].max():6.2f}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["gateup", "qkv", "down", "wo"])
