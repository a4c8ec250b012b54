#!/usr/bin/env python3
"""Register decode kernel (head sizes <= 32 and 256) under its own range rule (1024 workgroups, >= 128 keys) and under the ring kernel's (256, >= 32)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
nl = 16
for ctx in [int(x) for x in sys.argv[1:]] or [2048]:
    for heads, hkv, hs in ((16, 16, 256), (8, 1, 256), (16, 8, 256), (32, 32, 32), (32, 4, 32)):
        q = torch.randn((1, 1, heads, hs), device="cuda"); out = torch.zeros_like(q)
        kc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
        vc = [torch.randn((1, ctx, hkv, hs), device="cuda").half() for _ in range(nl)]
        row = {"ctx": ctx, "heads": heads, "heads_kv": hkv, "head_size": hs}
        for name, tgt, mk in (("rule_1024_128", 1024, 128), ("rule_256_32", 256, 32), ("rule_256_64", 256, 64), ("rule_512_64", 512, 64)):
            L.ns_hip_set_tuning(b"attn_wg_target", tgt); L.ns_hip_set_tuning(b"attn_min_keys", mk)
            shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
            ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
            def step():
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for il in range(nl):
                    a = pkg.attn_args(q.data_ptr(), kc[il].data_ptr(), vc[il].data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
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
            for _ in range(30): g.replay()
            e1.record(); torch.cuda.synchronize()
            row[name] = round(e0.elapsed_time(e1) / 30 / nl * 1e3, 2)
        print(json.dumps(row), flush=True)
L.ns_hip_set_tuning(b"attn_wg_target", 1024); L.ns_hip_set_tuning(b"attn_min_keys", 128)
