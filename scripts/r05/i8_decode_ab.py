#!/usr/bin/env python3
"""bench.py's int8-reference decode leg alone (the Llama-2-7B GEMV chain under NS_COMPUTE_REF_INT8, one HIP graph): tokens/s.
Run once with NS_I8_INKERNEL=0 (quantizer launches in front of every GEMV, round 4) and once without (gemv_kernel XV = 5)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import __graft_entry__ as ge
pkg = ge.load_package()
torch.cuda.set_device(0)
chain = bench.Chain(pkg, bench.CFG["n_layer"], 0, 1)
import ctypes as C


def step():  # (capture switches the current stream: the handle is re-read inside, like bench.py's Step)
    chain.st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    chain.step()


vals = [bench.decode_ref_int8(step, pkg) for _ in range(3)]
print(json.dumps({"in_kernel_quantizer": os.environ.get("NS_I8_INKERNEL", "1") != "0", "tokens_per_s_3_runs": vals}))
