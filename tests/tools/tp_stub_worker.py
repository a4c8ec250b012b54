"""Worker of tests/test_gpu_tp_stub.py (launched by torch.distributed.run, one process per rank, every rank on cuda:0):
libns_hip.so's tensor-parallel layer (csrc/ns_tp.cpp) with world > 1 over the shared-memory stand-in for RCCL
(tests/tools/stub_rccl.cpp, bound through NS_TP_RCCL_LIB).  gloo is only the side channel (unique id, IPC handles) —
what bench.py uses torch.distributed for.  argv: <repo root> <glue .so>"""
import ctypes as C
import os
import sys

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# NS_TP_WORKER_DEVICE=local (scripts/tp_first_contact.sh on a multi-GPU node): one GPU per rank, the real collective library;
# default: every rank on cuda:0 over the stand-in.  NS_TP_WORKER_EXACT=0: the collective library's sums are compared with the
# fp64 sums to 1e-6 (a ring does not add in rank order); the peer-memory kernel adds in rank order and stays bit-exact.
DEV = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("NS_TP_WORKER_DEVICE") == "local" else 0
EXACT = os.environ.get("NS_TP_WORKER_EXACT", "1") != "0"
torch.cuda.set_device(DEV)
pkg = ge.load_package()
L = pkg.lib()
L.ns_tp_init.restype = C.c_void_p
L.ns_tp_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
for f in ("ns_tp_reduce_add", "ns_tp_alltoall"):
    getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
L.ns_tp_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
L.ns_tp_barrier.argtypes = [C.c_void_p, C.c_void_p]
for f in ("ns_tp_reduce_add_host", "ns_tp_alltoall_host"):
    getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
L.ns_tp_broadcast_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for f in ("ns_tp_barrier_host", "ns_tp_destroy", "ns_tp_size", "ns_tp_rank", "ns_tp_is_master"):
    getattr(L, f).argtypes = [C.c_void_p]
L.ns_tp_attach_p2p.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]


def data(r, n, seed):
    """rank r's vector: every rank can rebuild every other rank's (the expected sums are computed locally)"""
    g = torch.Generator().manual_seed(1000 * seed + r)
    return torch.randn(n, generator=g)


def rank_order_sum(n, seed):
    acc = data(0, n, seed).clone()
    for r in range(1, world):
        acc += data(r, n, seed)  # fp32, rank order: what the stand-in (and a ring that starts at rank 0) computes
    return acc


def is_sum(got, n, seed, exact=None):
    """got (cpu tensor / numpy) == the ranks' sum: bit-equal to the rank-order fp32 sum, or (EXACT off) within 1e-6 of the fp64 sum"""
    got = torch.as_tensor(got)
    if EXACT if exact is None else exact:
        return torch.equal(got, rank_order_sum(n, seed))
    ref = sum(data(r, n, seed).double() for r in range(world))
    return float((got.double() - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))


# ---- 1. the unique id travels, every rank builds its communicator ------------------------------------------------
idbuf = C.create_string_buffer(128)
if rank == 0:
    assert L.ns_tp_unique_id(idbuf) == 0, pkg.last_error()
box = [bytes(idbuf.raw)]
dist.broadcast_object_list(box, src=0)
tp = L.ns_tp_init(rank, world, box[0], DEV)
assert tp, pkg.last_error()
assert L.ns_tp_size(tp) == world and L.ns_tp_rank(tp) == rank and L.ns_tp_is_master(tp) == int(rank == 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

# ---- 2. device-pointer collectives ----------------------------------------------------------------------------------
for n, seed in ((1, 1), (4096, 2), (11008, 3), (2048 * 4096, 4)):  # a decode vector ... a prefill activation (32 MB)
    x = data(rank, n, seed).cuda()
    y = torch.empty_like(x)
    assert L.ns_tp_reduce_add(tp, x.data_ptr(), y.data_ptr(), n, st) == 0, pkg.last_error()  # out of place
    assert L.ns_tp_reduce_add(tp, x.data_ptr(), x.data_ptr(), n, st) == 0, pkg.last_error()  # in place
    torch.cuda.synchronize()
    assert is_sum(y.cpu(), n, seed) and is_sum(x.cpu(), n, seed), (n, float((x.cpu() - rank_order_sum(n, seed)).abs().max()))
b = (data(0, 5000, 9) if rank == 0 else torch.zeros(5000)).cuda()
assert L.ns_tp_broadcast(tp, b.data_ptr(), b.numel(), st) == 0, pkg.last_error()
torch.cuda.synchronize()
assert torch.equal(b.cpu(), data(0, 5000, 9))
cnt = 777  # per peer
send = torch.cat([data(rank, cnt, 20 + p) for p in range(world)]).cuda()  # chunk p goes to rank p
recv = torch.empty_like(send)
assert L.ns_tp_alltoall(tp, send.data_ptr(), recv.data_ptr(), cnt, st) == 0, pkg.last_error()
assert L.ns_tp_barrier(tp, st) == 0, pkg.last_error()
assert torch.equal(recv.cpu(), torch.cat([data(p, cnt, 20 + rank) for p in range(world)]))

# ---- 3. captured in a HIP graph, replayed twice -----------------------------------------------------------------------
z = (torch.arange(4096, dtype=torch.float32) * (rank + 1)).cuda()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g.capture_begin()
    sst = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z.mul_(2.0)
    assert L.ns_tp_reduce_add(tp, z.data_ptr(), z.data_ptr(), z.numel(), sst) == 0, pkg.last_error()
    g.capture_end()
torch.cuda.current_stream().wait_stream(side)
g.replay()
g.replay()
torch.cuda.synchronize()
S = world * (world + 1) // 2
assert torch.equal(z.cpu(), torch.arange(4096, dtype=torch.float32) * float(2 * S * 2 * world)), z[:4]

# ---- 4. the host-pointer forms ne_compute_forward_all_reduce uses ------------------------------------------------------
h = data(rank, 3000, 30).numpy().copy()
o = np.zeros_like(h)
assert L.ns_tp_reduce_add_host(tp, h.ctypes.data, o.ctypes.data, h.size) == 0, pkg.last_error()
assert is_sum(o, 3000, 30)
hb = data(0, 100, 31).numpy().copy() if rank == 0 else np.zeros(100, np.float32)
assert L.ns_tp_broadcast_host(tp, hb.ctypes.data, hb.size) == 0 and np.array_equal(hb, data(0, 100, 31).numpy())
hs = np.concatenate([data(rank, 50, 40 + p).numpy() for p in range(world)])
hr = np.zeros_like(hs)
assert L.ns_tp_alltoall_host(tp, hs.ctypes.data, hr.ctypes.data, 50) == 0
assert np.array_equal(hr, np.concatenate([data(p, 50, 40 + rank).numpy() for p in range(world)]))
assert L.ns_tp_barrier_host(tp) == 0

# ---- 5. routing: with the peer-memory context attached, decode-sized in-place sums take the one-shot kernel, everything
#      else the collective library; both give the sum ---------------------------------------------------------------------
L.ns_hip_p2p_create.restype = C.c_void_p
L.ns_hip_p2p_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_char_p]
L.ns_hip_p2p_connect.argtypes = [C.c_void_p, C.c_char_p]
for f in ("ns_hip_p2p_disconnect", "ns_hip_p2p_destroy", "ns_hip_p2p_error"):
    getattr(L, f).argtypes = [C.c_void_p]
handle = C.create_string_buffer(64)
p2p = L.ns_hip_p2p_create(rank, world, 1 << 20, handle)
infos = [None] * world
dist.all_gather_object(infos, (bool(p2p), bytes(handle.raw)))
routed = "stand-in only"
if all(ok for ok, _ in infos) and L.ns_hip_p2p_connect(p2p, b"".join(hh for _, hh in infos)) == 0:
    oks = [None] * world
    dist.all_gather_object(oks, True)
    assert L.ns_tp_attach_p2p(tp, p2p, 1 << 20) == 0
    for n, seed in ((4096, 50), (1 << 18, 51), ((1 << 18) + 4, 52)):  # 16 KB, 1 MB (the slot), just above it
        x = data(rank, n, seed).cuda()
        assert L.ns_tp_reduce_add(tp, x.data_ptr(), x.data_ptr(), n, st) == 0, pkg.last_error()
        torch.cuda.synchronize()
        got, want = x.cpu(), rank_order_sum(n, seed)
        # the kernel adds in rank order too (ns_p2p.hip): bit-equal whatever the collective library does; above the slot the
        # library serves the call
        assert is_sum(got, n, seed, exact=True if n * 4 <= (1 << 20) else None), (n, float((got - want).abs().max()))
    assert L.ns_hip_p2p_error(p2p) == 0
    routed = "peer-memory kernel + stand-in"
    L.ns_tp_attach_p2p(tp, None, 0)
    torch.cuda.synchronize()
    dist.barrier()
    L.ns_hip_p2p_disconnect(p2p)
    dist.barrier()
    L.ns_hip_p2p_destroy(p2p)
else:
    L.ns_hip_reset_error()

torch.cuda.synchronize()
dist.barrier()
L.ns_tp_destroy(tp)

# ---- 6. glue/parallel_context_hip.cpp: the reference's eight functions, bootstrap through the id file ------------------
C.CDLL(pkg.LIB_PATH, mode=C.RTLD_GLOBAL)  # the glue leaves ns_tp_* undefined, like the reference's objects would
G = C.CDLL(sys.argv[2])
G.init_parallel_context.restype = C.c_void_p
for f in ("get_tp_size", "get_tp_rank", "is_master", "barrier"):
    getattr(G, f).argtypes = [C.c_void_p]
G.is_master.restype = C.c_bool
G.reduce_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
G.alltoall.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
G.broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
ctx = G.init_parallel_context()
assert G.get_tp_size(ctx) == world and G.get_tp_rank(ctx) == rank and G.is_master(ctx) == (rank == 0)
t = data(rank, 4096, 60).numpy().copy()  # ne_compute_forward_all_reduce: reduce_add(dst->data, dst->data, ...) in place
G.reduce_add(ctx, t.ctypes.data, t.ctypes.data, t.size)
assert is_sum(t, 4096, 60)
bb = data(0, 64, 61).numpy().copy() if rank == 0 else np.zeros(64, np.float32)
G.broadcast(ctx, bb.ctypes.data, bb.size)
assert np.array_equal(bb, data(0, 64, 61).numpy())
a2s = np.concatenate([data(rank, 32, 70 + p).numpy() for p in range(world)])
a2r = np.zeros_like(a2s)
G.alltoall(ctx, a2s.ctypes.data, a2r.ctypes.data, 32)
assert np.array_equal(a2r, np.concatenate([data(p, 32, 70 + rank).numpy() for p in range(world)]))
G.barrier(ctx)
dist.barrier()
print("TP_STUB_OK rank %d of %d (%s)" % (rank, world, routed), flush=True)
dist.destroy_process_group()
