"""The native tensor-parallel layer (csrc/ns_tp.cpp: RCCL through a C ABI, no torch in the data path) on ONE GPU:
RCCL refuses two ranks per device, so what a single-GPU box can exercise is everything BUT the wire — library
loading (dlopen), unique-id creation, communicator set-up, the collective calls on a one-rank communicator
(NS_TP_FORCE_RCCL=1 keeps them from being short-cut), stream ordering and HIP-graph capture of them, and the
host-pointer forms ne_compute_forward_all_reduce uses (ne_layers.c:5466-5476).  Runs in a fresh interpreter (the env
switch is read once)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
idb = C.create_string_buffer(128)
assert L.ns_tp_unique_id(idb) == 0, pkg.last_error()
assert any(idb.raw), "unique id is all zero"
tp = L.ns_tp_init(0, 1, idb.raw, 0)
assert tp, pkg.last_error()
assert L.ns_tp_size(tp) == 1 and L.ns_tp_rank(tp) == 0 and L.ns_tp_is_master(tp) == 1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(8192, device="cuda"); ref = x.clone()
assert L.ns_tp_reduce_add(tp, x.data_ptr(), x.data_ptr(), x.numel(), st) == 0, pkg.last_error()   # ncclAllReduce, one rank
y = torch.empty_like(x)
assert L.ns_tp_reduce_add(tp, x.data_ptr(), y.data_ptr(), x.numel(), st) == 0
assert L.ns_tp_broadcast(tp, x.data_ptr(), x.numel(), st) == 0
assert L.ns_tp_barrier(tp, st) == 0
torch.cuda.synchronize()
assert torch.equal(x, ref) and torch.equal(y, ref)
# inside a HIP graph, replayed
g = torch.cuda.CUDAGraph()
z = torch.randn(4096, device="cuda"); zr = z.clone()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g.capture_begin()
    sst = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z.mul_(2.0)
    assert L.ns_tp_reduce_add(tp, z.data_ptr(), z.data_ptr(), z.numel(), sst) == 0, pkg.last_error()
    g.capture_end()
torch.cuda.current_stream().wait_stream(side)
g.replay(); g.replay(); torch.cuda.synchronize()
assert torch.equal(z, zr * 4.0)
# host-pointer forms
h = np.arange(1000, dtype=np.float32); o = np.zeros_like(h)
assert L.ns_tp_reduce_add_host(tp, h.ctypes.data, o.ctypes.data, h.size) == 0 and np.array_equal(h, o)
assert L.ns_tp_broadcast_host(tp, h.ctypes.data, h.size) == 0 and L.ns_tp_barrier_host(tp) == 0
L.ns_tp_destroy(tp)
print("TP_NATIVE_OK")
'''


@pytest.mark.gpu
def test_native_tp_layer_single_rank_through_rccl():
    env = dict(os.environ, NS_TP_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert "TP_NATIVE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
