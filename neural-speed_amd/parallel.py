"""Tensor-parallel communication layer: the MI355X replacement of neural-speed's `parallel_class`
(/root/reference/neural_speed/core/parallel_context.{h,cpp}, shared_memory_ccl.hpp).

The reference bootstraps oneCCL over MPI, one process per CPU socket, and exposes eight functions
(parallel_context.h:40-47): init_parallel_context, get_tp_size, get_tp_rank, is_master, barrier, broadcast,
alltoall, reduce_add.  Here it is one process per GPU under torch.distributed — backend "nccl" IS RCCL on ROCm, i.e.
all-reduce over the xGMI mesh — and "gloo" on CPU for the multi-process tests.  Same names, same argument meaning:
buffers are fp32 tensors reduced/broadcast IN PLACE (ne_layers.c:5466-5476 does the same on dst->data).

Split rules for Llama-family weights (model_files.h:145-190): wq/wk/wv/w1/w3 are split along N (TENSOR_1D_ROW),
wo/w2 along K (TENSOR_1D_COLUMN) and followed by one all-reduce; everything else is replicated.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

TENSOR_NO_CHANGE, TENSOR_1D_ROW, TENSOR_1D_COLUMN, TENSOR_1D_QKV_ROW, TENSOR_1D_QKV_COLUMN, TENSOR_1D_ONLY_MASTER = range(6)

# model_load_tensor::calc_split_type, model_files.h:145-190 — the full key table, in the reference's order of precedence
# (later groups override earlier ones).  ".mlp.fc_in.bias" sits in the reference's COLUMN group: a 1-D tensor split
# along its only axis, i.e. the N slice that goes with fc_in.weight's ROW split.
_ROW_KEYS = (".attn.q_proj.weight", ".attn.k_proj.weight", ".attn.v_proj.weight", ".mlp.fc_in.weight",
             ".mlp.gate_proj.weight", ".mlp.up_proj.weight",                 # baichuan
             ".mlp.dense_h_to_4h.weight",                                      # chatglm2
             ".attention.wq.weight", ".attention.wk.weight", ".attention.wv.weight", ".feed_forward.w1.weight",
             ".feed_forward.w3.weight")                                        # llama
_QKV_ROW_KEYS = (".self_attn.W_pack.weight", ".self_attention.query_key_value.weight")
_QKV_COL_KEYS = (".self_attention.query_key_value.bias",)
_COL_KEYS = (".mlp.fc_in.bias", ".mlp.fc_out.weight", ".attn.out_proj.weight", ".self_attention.dense.weight",
             ".self_attn.o_proj.weight", ".mlp.down_proj.weight",            # baichuan
             ".mlp.dense_4h_to_h.weight",                                      # chatglm2
             ".attention.wo.weight", ".feed_forward.w2.weight")
_MASTER_KEYS = (".mlp.fc_out.bias",)


def calc_split_type(name):
    """model_load_tensor::calc_split_type (model_files.h:145-190): how a tensor is cut for tensor parallelism.
    ROW = N slice (independent output columns), COLUMN = K slice (partial sums, all-reduced afterwards), QKV_ROW /
    QKV_COLUMN = the fused q|k|v tensor cut inside each of its three parts, ONLY_MASTER = kept on rank 0 and zero on the
    others (a bias that is added once, after the all-reduce).  Everything else is replicated."""
    t = TENSOR_NO_CHANGE
    if any(k in name for k in _ROW_KEYS):
        t = TENSOR_1D_ROW
    if any(k in name for k in _QKV_ROW_KEYS):
        t = TENSOR_1D_QKV_ROW
    if any(k in name for k in _QKV_COL_KEYS):
        t = TENSOR_1D_QKV_COLUMN
    if any(k in name for k in _COL_KEYS):
        t = TENSOR_1D_COLUMN
    if any(k in name for k in _MASTER_KEYS):
        t = TENSOR_1D_ONLY_MASTER
    return t


class ParallelContext:
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._p2p, self._p2p_max = None, 0
        self._tp = None  # native communicator (ns_tp_*, RCCL without torch in the data path)
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend)

    # parallel_context.h:40-47 ---------------------------------------------------------------------------------
    def get_tp_size(self):
        return self.world

    def get_tp_rank(self):
        return self.rank

    def is_master(self):
        return self.rank == 0

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def broadcast(self, buf, root=0):
        """parallel_context.cpp:59-62: broadcast `buf` (in place) from `root`"""
        if self.world > 1:
            dist.broadcast(buf, src=root)
        return buf

    def alltoall(self, send, recv):
        """declared in the reference, no caller (parallel_context.cpp:63-65)"""
        if self.world > 1:
            dist.all_to_all_single(recv, send)
        else:
            recv.copy_(send)
        return recv

    def reduce_add(self, buf):
        """parallel_context.cpp:47-58: fp32 sum over ranks, in place.  Like the reference, which hands decode-sized
        buffers to shm_all_reduce (shared_memory_ccl.hpp:100-139) and the rest to oneCCL, fp32 device buffers that fit
        the peer-memory slot go through the one-shot xGMI kernel (enable_p2p), everything else through RCCL."""
        if self.world > 1:
            native_ok = buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
            if native_ok and self._tp is not None:
                # ns_tp_reduce_add: the one-shot xGMI kernel for buffers that fit the attached peer-memory context, RCCL
                # otherwise — C ABI end to end, asynchronous on the current stream, capturable
                from . import check, lib
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                check(lib().ns_tp_reduce_add(self._tp, buf.data_ptr(), buf.data_ptr(), buf.numel(), st), "ns_tp_reduce_add")
            elif (native_ok and self._p2p is not None and buf.numel() * 4 <= self._p2p_max and buf.data_ptr() % 16 == 0):
                from . import check, lib
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                check(lib().ns_hip_p2p_all_reduce_f32(self._p2p, buf.data_ptr(), buf.numel(), st), "p2p all-reduce")
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return buf

    # native communicator: RCCL through the C ABI (csrc/ns_tp.cpp) ---------------------------------------------------
    def enable_native(self, device=None):
        """Collective.  Rank 0 draws the RCCL unique id, the process group carries its 128 bytes to the others, every
        rank builds its communicator with ns_tp_init.  True when ALL ranks succeeded; otherwise every rank stays on
        torch.distributed."""
        if self.world == 1 or self._tp is not None:
            return self._tp is not None
        from . import lib
        L = lib()
        idbuf = C.create_string_buffer(128)
        ok = True
        if self.rank == 0:
            ok = L.ns_tp_unique_id(idbuf) == 0
        box = [(ok, bytes(idbuf.raw))]
        dist.broadcast_object_list(box, src=0)
        ok, raw = box[0]
        tp = None
        if ok:
            # the launcher's LOCAL_RANK, unless the caller names a device (argument, or NS_TP_LOCAL_RANK: several ranks of a
            # test share one GPU).  NOT torch.cuda.current_device(): with a process group the caller initialised itself (or
            # gloo) nobody has called set_device, every rank would report device 0 and RCCL would refuse the duplicate GPU
            if device is None:
                device = int(os.environ["NS_TP_LOCAL_RANK"]) if os.environ.get("NS_TP_LOCAL_RANK") else self.local_rank
            dev = int(device)
            if torch.cuda.is_available() and dev < torch.cuda.device_count():
                torch.cuda.set_device(dev)
            tp = L.ns_tp_init(self.rank, self.world, raw, dev)
        oks = [None] * self.world
        dist.all_gather_object(oks, bool(tp))
        if not all(oks):
            L.ns_hip_reset_error()
            if tp:
                L.ns_tp_destroy(tp)
            return False
        self._tp = tp
        if self._p2p is not None:
            L.ns_tp_attach_p2p(self._tp, self._p2p, self._p2p_max)
        return True

    def native_enabled(self):
        return self._tp is not None

    def disable_native(self):
        if self._tp is None:
            return
        from . import lib
        torch.cuda.synchronize()
        dist.barrier()
        lib().ns_tp_destroy(self._tp)
        self._tp = None

    # one-shot all-reduce over peer-mapped HBM (csrc/ns_p2p.hip) ----------------------------------------------------
    def enable_p2p(self, max_bytes=1 << 20):
        """Collective.  Every rank allocates its segment, the IPC handles travel through the process group, every rank
        maps every peer.  True when ALL ranks connected (then reduce_add uses the kernel for buffers <= max_bytes);
        on any failure every rank drops back to RCCL together."""
        if self.world == 1:
            return False
        if self._p2p is not None:
            return True
        from . import lib
        L = lib()
        handle = C.create_string_buffer(64)
        ctx = L.ns_hip_p2p_create(self.rank, self.world, max_bytes, handle)
        infos = [None] * self.world
        dist.all_gather_object(infos, (bool(ctx), bytes(handle.raw)))
        ok = all(o for o, _ in infos)
        if ok:
            ok = L.ns_hip_p2p_connect(ctx, b"".join(h for _, h in infos)) == 0
        oks = [None] * self.world
        dist.all_gather_object(oks, bool(ok))
        if not all(oks):
            L.ns_hip_reset_error()
            if ctx:
                L.ns_hip_p2p_disconnect(ctx)
            dist.barrier()  # nobody frees a segment a peer still maps
            if ctx:
                L.ns_hip_p2p_destroy(ctx)
            return False
        self._p2p, self._p2p_max = ctx, max_bytes
        if self._tp is not None:
            L.ns_tp_attach_p2p(self._tp, ctx, max_bytes)
        return True

    def p2p_enabled(self):
        return self._p2p is not None

    def p2p_error(self):
        """Collective: True when a flag wait timed out on ANY rank (synchronises the device)."""
        if self._p2p is None:
            return False
        from . import lib
        torch.cuda.synchronize()
        bad = [None] * self.world
        dist.all_gather_object(bad, lib().ns_hip_p2p_error(self._p2p) != 0)
        return any(bad)

    def disable_p2p(self):
        """Collective: unmap the peers, barrier, free the own segment; reduce_add is RCCL again."""
        if self._p2p is None:
            return
        from . import lib
        L = lib()
        torch.cuda.synchronize()
        if self._tp is not None:
            L.ns_tp_attach_p2p(self._tp, None, 0)
        L.ns_hip_p2p_disconnect(self._p2p)
        dist.barrier()
        L.ns_hip_p2p_destroy(self._p2p)
        self._p2p = None

    def shard_range(self, size, quantum=1):
        """[begin, end) of this rank's slice of an axis of `size` split evenly (the reference requires divisibility,
        model_files.h:1603-1680); `quantum` guards the kernel's alignment needs (16 columns / one k-step)."""
        per = size // self.world
        if per * self.world != size or per % quantum:
            raise ValueError("axis %d does not split %d ways in units of %d" % (size, self.world, quantum))
        return self.rank * per, (self.rank + 1) * per

    def shard_weight(self, weight, split_type, stream=None):
        """TP shard of a device weight (ns_hip_weight_slice): ROW -> N slice, COLUMN -> K slice."""
        if self.world == 1 or split_type == TENSOR_NO_CHANGE:
            return weight
        if split_type in (TENSOR_1D_QKV_ROW, TENSOR_1D_QKV_COLUMN, TENSOR_1D_ONLY_MASTER):
            # fused q|k|v tensors are cut inside each third (model_files.h:1603-1680) and master-only biases are fp32
            # vectors: neither is a single N / K slice of a quantized weight — refuse instead of replicating silently
            raise NotImplementedError("split type %d is not a plain N / K slice of a packed weight" % split_type)
        if split_type == TENSOR_1D_ROW:
            n0, n1 = self.shard_range(weight.n, 16)
            return weight.slice(n0, n1, 0, weight.k, stream)
        k0, k1 = self.shard_range(weight.k, 1)
        return weight.slice(0, weight.n, k0, k1, stream)


    def shard_blob(self, blob, split_type):
        """TP shard of a reference-format blob ON THE HOST (numpy uint8 array in, new 64-byte aligned array out), before anything
        is uploaded — the reference's per-rank bestla_split_weight at load (model_files.h:1593-1640) without its fp32 round trip:
        ns_bestla_split_weight copies the rank's codes / scales / zero points / block sums, byte-identical to packing the cut
        matrix.  ROW -> this rank's columns, COLUMN -> this rank's slice of K (cuts must fall on group boundaries)."""
        import ctypes as C
        import numpy as np
        if self.world == 1 or split_type == TENSOR_NO_CHANGE:
            return blob
        if split_type in (TENSOR_1D_QKV_ROW, TENSOR_1D_QKV_COLUMN, TENSOR_1D_ONLY_MASTER):
            raise NotImplementedError("split type %d is not a plain N / K slice of a packed weight" % split_type)
        from . import lib
        L = lib()
        L.ns_bestla_split_weight_size.restype = C.c_ulonglong
        L.ns_bestla_split_weight_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ns_bestla_split_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_int]
        n, k = C.c_int(0), C.c_int(0)
        L.ns_blob_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if L.ns_blob_shape(blob.ctypes.data, C.byref(n), C.byref(k)) != 0:
            raise ValueError("not a BTLA blob")
        if split_type == TENSOR_1D_ROW:
            (n0, n1), (k0, k1) = self.shard_range(n.value, 1), (0, k.value)
        else:
            (n0, n1), (k0, k1) = (0, n.value), self.shard_range(k.value, 1)
        need = L.ns_bestla_split_weight_size(blob.ctypes.data, n1 - n0, k1 - k0)
        if not need:
            raise ValueError("ns_bestla_split_weight_size failed")
        raw = np.zeros(int(need) + 64, np.uint8)
        off = (-raw.ctypes.data) % 64
        out = raw[off:off + int(need)]
        rc = L.ns_bestla_split_weight(blob.ctypes.data, out.ctypes.data, need, n0, n1, k0, k1)
        if rc != 0:
            raise ValueError("ns_bestla_split_weight: rc %d (a cut inside a quantisation group needs the re-quantising route)" % rc)
        return out


_ctx = None


def init_parallel_context(backend=None):
    global _ctx
    if _ctx is None:
        _ctx = ParallelContext(backend)
    return _ctx


def get_tp_size():
    return init_parallel_context().get_tp_size()


def get_tp_rank():
    return init_parallel_context().get_tp_rank()


def is_master():
    return init_parallel_context().is_master()


def barrier():
    init_parallel_context().barrier()


def broadcast(buf, root=0):
    return init_parallel_context().broadcast(buf, root)


def alltoall(send, recv):
    return init_parallel_context().alltoall(send, recv)


def reduce_add(buf):
    return init_parallel_context().reduce_add(buf)


def enable_p2p(max_bytes=1 << 20):
    return init_parallel_context().enable_p2p(max_bytes)
