#!/usr/bin/env python3
"""Ring decode kernel at batch 8 (256+ base workgroups): workgroup target of the range rule."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
nl = 8
for ctx in [int(x) for x in sys.argv[1:]] or [2048]:
    for bs, heads, hkv, hs in ((8, 32, 32, 128), (8, 32, 8, 128), (4, 32, 32, 128), (16, 32, 8, 128)):
        q = torch.randn((bs, 1, heads, hs), device="cuda"); out = torch.zeros_like(q)
        kc = [torch.randn((bs, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
        vc = [torch.randn((bs, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
        row = {"ctx": ctx, "batch": bs, "heads": heads, "heads_kv": hkv, "MB": round(2 * bs * ctx * hkv * hs * 2 / 1e6, 1)}
        for name, stream, tgt in (("registers", 0, 256), ("rings_256", 1, 256), ("rings_512", 1, 512), ("rings_1024", 1, 1024)):
            L.ns_hip_set_tuning(b"attn_stream", stream); L.ns_hip_set_tuning(b"attn_stream_wg_target", tgt)
            shape = pkg.AttnShape(bs, heads, hkv, hs, 1, ctx)
            ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
            def step():
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for il in range(nl):
                    a = pkg.attn_args(q.data_ptr(), kc[il].data_ptr(), vc[il].data_ptr(), out.data_ptr(), bs, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
                    a.tmp = ws.data_ptr()
                    pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), None, st))
            step(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            for _ in range(5): g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g.replay()
            e1.record(); torch.cuda.synchronize()
            row[name] = round(e0.elapsed_time(e1) / 20 / nl * 1e3, 2)
        print(json.dumps(row), flush=True)
        del kc, vc
L.ns_hip_set_tuning(b"attn_stream", 1); L.ns_hip_set_tuning(b"attn_stream_wg_target", 256)
