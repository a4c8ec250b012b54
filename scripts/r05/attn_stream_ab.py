#!/usr/bin/env python3
"""Decode attention (split + merge per call), K / V through LDS rings (attn_stream_kernel, ns_hip_set_tuning("attn_stream", 1)) against through
registers (attn_split_kernel, 0): 32 layers' worth of distinct fp16 caches in one HIP graph, alternating replays on one box, microseconds per
call and whether the outputs are the same bits.  Shapes: Llama-2-7B (32 heads), GQA 32 / 8 (Mistral / Llama-3), head size 64; position-major
and head-major caches.
Usage: scripts/r05/attn_stream_ab.py [ctx ...]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
ctxs = [int(x) for x in sys.argv[1:]] or [2048]
nl = 32
out_rows = []
for ctx in ctxs:
    for heads, hkv, hs, layout in ((32, 32, 128, "position-major"), (32, 32, 128, "head-major"), (32, 8, 128, "position-major"), (32, 32, 64, "position-major"), (32, 8, 64, "position-major"), (71, 1, 64, "position-major")):
        q = torch.randn((1, 1, heads, hs), device="cuda")
        out = torch.zeros_like(q)
        out16 = torch.zeros((1, 1, heads, hs), device="cuda", dtype=torch.float16)
        shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
        ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
        shp = (1, ctx, hkv, hs) if layout == "position-major" else (1, hkv, ctx, hs)
        kc = [torch.randn(shp, device="cuda").half() for _ in range(nl)]
        vc = [torch.randn(shp, device="cuda").half() for _ in range(nl)]

        def step():
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for il in range(nl):
                a = pkg.attn_args(q.data_ptr(), kc[il].data_ptr(), vc[il].data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
                if layout == "head-major":
                    a.step_k_head_num = a.step_v_head_num = ctx * hs
                    a.step_k_sl = a.step_v_sl = hs
                a.tmp = ws.data_ptr()
                pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), out16.data_ptr(), st))
        graphs, refs = {}, {}
        for mode in (0, 1):
            L.ns_hip_set_tuning(b"attn_stream", mode)
            step(); torch.cuda.synchronize()
            refs[mode] = out.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            graphs[mode] = g
        t = {0: [], 1: []}
        for rnd in range(4):
            for mode in (0, 1):
                g = graphs[mode]
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record(); torch.cuda.synchronize()
                t[mode].append(e0.elapsed_time(e1) / 20 / nl * 1e3)
        byt = 2 * ctx * hkv * hs * 2
        row = {"ctx": ctx, "heads": heads, "heads_kv": hkv, "head_size": hs, "layout": layout,
               "registers_us": [round(x, 2) for x in t[0]], "lds_rings_us": [round(x, 2) for x in t[1]],
               "registers_TBps": round(byt / min(t[0]) / 1e6, 2), "lds_rings_TBps": round(byt / min(t[1]) / 1e6, 2),
               "same_bits": bool(torch.equal(refs[0], refs[1])), "finite": bool(torch.isfinite(refs[1]).all().item())}
        print(json.dumps(row), flush=True)
        del kc, vc
L.ns_hip_set_tuning(b"attn_stream", 1)
