#!/usr/bin/env python3
"""Load-path numbers for profiles/ (SURVEY 8f-3): one rank of TP = 8 of a Llama-2-70B-shaped model, Q4_0 g32 bf16 scales.
Per layer: the seven full-size blobs exist on the host (as the reference's loader reads them from the file, model_files.h:1593-1640),
the rank cuts its shard out of each ON THE HOST (ns_bestla_split_weight), and hands only the shard to the reference's device load
entry (bestla_device_load_storage -> the streaming layout lands in the slice the graph reserved).  `layers` different layers are
really processed; seconds and HBM are reported per layer and extrapolated to 80 layers.
usage: load_bench.py [layers=4] [rank=3]"""
import ctypes as C, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
from neural_speed_amd import parallel as par
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
world, d, ff, kvd = 8, 8192, 28672, 1024
vp = C.c_void_p
L.bestla_create_device.restype = vp; L.bestla_create_device.argtypes = [C.c_bool]
L.bestla_get_device_queue.restype = vp; L.bestla_get_device_queue.argtypes = [vp]
L.bestla_device_malloc.restype = vp; L.bestla_device_malloc.argtypes = [C.c_size_t, vp]
L.bestla_device_storage_size.restype = C.c_size_t
L.bestla_device_load_storage.argtypes = [vp, vp, vp, vp]
L.bestla_device_sync.argtypes = [vp]
L.ns_hip_device_load_stats.argtypes = [vp]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def blob(n, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    b = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(b.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
    torch.cuda.synchronize()
    return b.cpu().numpy()


names = [("attention.wq.weight", d, d), ("attention.wk.weight", kvd, d), ("attention.wv.weight", kvd, d), ("attention.wo.weight", d, d),
         ("feed_forward.w1.weight", ff, d), ("feed_forward.w3.weight", ff, d), ("feed_forward.w2.weight", d, ff)]
ctx = par.ParallelContext.__new__(par.ParallelContext)
ctx.rank, ctx.world = rank, world
dev = L.bestla_create_device(False); q = L.bestla_get_device_queue(dev)
free0 = torch.cuda.mem_get_info()[0]
t_slice = t_load = 0.0
full_bytes = shard_bytes = 0
keep = []
for il in range(layers):
    full = [(nm, blob(n, k, 10 * il + j)) for j, (nm, n, k) in enumerate(names)]
    torch.cuda.empty_cache()
    free_before = torch.cuda.mem_get_info()[0]
    for nm, b in full:
        full_bytes += b.size
        t0 = time.perf_counter()
        sh = ctx.shard_blob(b, par.calc_split_type("layers.%d.%s" % (il, nm)))
        t_slice += time.perf_counter() - t0
        shard_bytes += sh.size
        size = int(np.frombuffer(sh[:8].tobytes(), np.uint64)[0])
        dptr = L.bestla_device_malloc((size + 255) // 256 * 256, q)     # the slice the graph reserves for the tensor
        stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
        t0 = time.perf_counter()
        L.bestla_device_load_storage(sh.ctypes.data, stor.ctypes.data, dptr, q)
        t_load += time.perf_counter() - t0
        keep.append((stor, dptr))
t0 = time.perf_counter(); L.bestla_device_sync(q); t_load += time.perf_counter() - t0
stats = (C.c_uint64 * 6)(); L.ns_hip_device_load_stats(stats)
print(json.dumps({
    "model": "Llama-2-70B-shaped, Q4_0 g32 bf16, rank %d of TP = 8" % rank, "layers_processed": layers,
    "per_layer": {"full_blobs_MB_on_host": round(full_bytes / layers / 1e6, 1), "rank_shard_blobs_MB": round(shard_bytes / layers / 1e6, 1),
                  "host_cut_ms": round(1e3 * t_slice / layers, 1), "upload_and_relayout_ms": round(1e3 * t_load / layers, 2)},
    "extrapolated_80_layers": {"host_cut_s": round(80 * t_slice / layers, 2), "upload_and_relayout_s": round(80 * t_load / layers, 2),
                                "uploaded_GB": round(80 * shard_bytes / layers / 1e9, 2), "a_rank_that_uploaded_everything_GB": round(80 * full_bytes / layers / 1e9, 2)},
    "hbm": {"streaming_layout_in_graph_slices_MB": round(stats[2] / 1e6, 1), "own_allocations_MB": round(stats[3] / 1e6, 1),
            "slices_reserved_MB": round(stats[1] / 1e6, 1), "ratio_streaming_bytes_over_reserved": round(stats[2] / max(1, stats[1]), 3)}}))
