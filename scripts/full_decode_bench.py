#!/usr/bin/env python3
"""Whole Llama-2-7B decode token, device resident, in one HIP graph, at a few context lengths: bench.py's full_token()
(the fused form: norms carried across the GEMMs, RoPE + kv-append as the QKV epilogue; and one launch per operator)
next to the GEMM-only chain.  Usage: scripts/full_decode_bench.py [ctx ...]   (default 128 512 2048)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import bench

pkg = ge.load_package()
torch.cuda.set_device(0)
chain = bench.Chain(pkg, bench.CFG["n_layer"], 0, 1)
res = {}
only_fused = os.environ.get("NS_FULL_ONLY_FUSED") == "1"  # profiling: nothing but the fused graph's kernels
for ctx in [int(a) for a in sys.argv[1:]] or [128, 512, 2048]:
    f, lf = bench.full_token(chain, pkg, ctx, fused=True)
    if only_fused:
        res["ctx_%d" % ctx] = {"fused": f}
        continue
    u, lu = bench.full_token(chain, pkg, ctx, fused=False)
    rel = float((lf.double() - lu.double()).norm() / lu.double().norm())
    res["ctx_%d" % ctx] = {"fused": f, "one_launch_per_operator": u, "logits_rel_l2": round(rel, 6)}
print(json.dumps(res, indent=1))
