import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
torch.manual_seed(0)
for heads, hkv, hs, ctx in ((32, 32, 128, 64), (32, 32, 128, 128), (32, 32, 128, 100), (32, 32, 128, 2048), (32, 8, 128, 777), (32, 32, 64, 512)):
    q = torch.randn((1, 1, heads, hs), device="cuda")
    kc = torch.randn((1, ctx, hkv, hs), device="cuda").half(); vc = torch.randn((1, ctx, hkv, hs), device="cuda").half()
    shape = pkg.AttnShape(1, heads, hkv, hs, 1, ctx)
    ws = torch.empty(max(64, L.bestla_fusion_attn_workspace_size(C.byref(shape))), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for mode in (0, 1):
        L.ns_hip_set_tuning(b"attn_stream", mode)
        out = torch.zeros_like(q)
        a = pkg.attn_args(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), 1, heads, hkv, hs, 1, ctx, hs ** -0.5, pkg.ATTN_CAUSAL)
        a.tmp = ws.data_ptr()
        pkg.check(L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(C.byref(a), None, st))
        torch.cuda.synchronize()
        outs[mode] = out
    g = heads // hkv
    kk = kc.float().repeat_interleave(g, dim=2); vv = vc.float().repeat_interleave(g, dim=2)
    sc = torch.einsum("bqhd,bkhd->bhqk", q, kk) * hs ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), vv)
    d0 = (outs[0] - ref).abs().amax(dim=-1).flatten(); d1 = (outs[1] - ref).abs().amax(dim=-1).flatten()
    print(heads, hkv, hs, ctx, "regs err", float(d0.max()), "rings err", float(d1.max()), "equal", torch.equal(outs[0], outs[1]), flush=True)
    if d1.max() > 1e-3:
        print("  per head rings err", [round(float(x), 4) for x in d1[:8]], "per dim (head 0)", [round(float(x), 3) for x in (outs[1] - ref)[0, 0, 0, :16].abs()], flush=True)
