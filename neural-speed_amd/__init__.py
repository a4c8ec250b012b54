"""neural-speed_amd — MI355X (gfx950) backend behind neural-speed's BesTLA operator surface.

The product is the C-ABI shared library ``libns_hip.so`` (sources in ``csrc/``, interface in
``include/ns_bestla.h``).  This Python module is only the thin ctypes loader used by tests, bench.py and
__graft_entry__.py; it contains no compute and imports nothing from ``oracle/``.

Because the directory name carries a hyphen (it mirrors the reference's repository name) it is loaded with
``importlib`` — see ``load_package()`` in ``__graft_entry__.py``.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NS_LIB_PATH") or os.path.join(HERE, "libns_hip.so")  # NS_LIB_PATH: diagnostics builds

# BTLA_DTYPE codes (reference: bestla/bestla/bestla.h:38-87)
F32 = 32
F16 = 16
BF16 = 16 | (1 << 16)
S8 = 8 | (1 << 8)
S4 = 4 | (1 << 8)
F4_E2M1 = 4
F4_BNB = 4 | (1 << 16)
F4_NF4 = 4 | (2 << 16)
F8_E4M3 = 8
F8_E5M2 = 8 | (1 << 16)
F8_E8M0 = 8 | (3 << 16)  # scale type of the fp8 weights: shared exponent
INT_TYPES = {b: b | (1 << 8) for b in range(1, 9)}
# ne_comp_type (neural_speed/core/data_types.h:57-63)
COMP_UNDEF, COMP_F32, COMP_BF16, COMP_F16, COMP_INT8 = range(5)
# enum ns_epilogue
EPI_NONE, EPI_ADD, EPI_MUL, EPI_ADD_GELU, EPI_GELU, EPI_SILU = range(6)
CORE_AUTO = -1
# ne_attn_flags_t (neural_speed/core/ne_layers.h:65-72)
ATTN_CAUSAL, ATTN_ALIBI8, ATTN_PREFER_FP32, ATTN_TANH30 = 1, 2, 4, 8


class NormLink(C.Structure):
    """ns_norm_link (include/ns_bestla.h): an RMS norm carried from the operator that produces a tensor to the GEMM that
    consumes it"""
    _fields_ = [("in_ssq", C.c_void_p), ("in_parts", C.c_int), ("in_stride", C.c_int), ("eps", C.c_float),
                ("norm_size", C.c_int), ("out_gamma", C.c_void_p), ("out_ssq", C.c_void_p), ("out_stride", C.c_int)]


class QkvRope(C.Structure):
    """ns_qkv_rope (include/ns_bestla.h)"""
    _fields_ = ([("kcache16", C.c_void_p), ("vcache16", C.c_void_p), ("cos_sin", C.c_void_p)] +
                [(n, C.c_int) for n in ("heads", "heads_kv", "head_size", "n_past", "n_dims", "mode")] +
                [("cache_step_sl", C.c_longlong), ("cache_step_head", C.c_longlong), ("flags", C.c_int)])


class AttnShape(C.Structure):
    """attn_shape_t (mha_dense.h:24-26)"""
    _fields_ = [(n, C.c_int) for n in ("batch_size", "head_num", "heads_kv", "head_size", "sl_q", "sl_kv")]


class KvShape(C.Structure):
    """kv_shape_t (mha_dense.h:29-33)"""
    _fields_ = [(n, C.c_uint32) for n in ("heads_kv", "head_size", "sl_kv_max")]


class KvCacheInfo(C.Structure):
    """kv_cache_info_t (mha_dense.h:49-54): byte sizes per batch entry and BYTE strides of the library's cache layout"""
    _fields_ = ([("k_bytes", C.c_size_t), ("v_bytes", C.c_size_t), ("k_layout", C.c_int), ("v_layout", C.c_int)] +
                [(n, C.c_int) for n in ("stride_k_head_num", "stride_k_sl", "stride_k_head_size",
                                        "stride_v_head_num", "stride_v_sl", "stride_v_head_size")])


class KvUpdateArgs(C.Structure):
    """bestla_fusion_attn_fp32_update_kv_args_t (mha_dense.h:130-136); steps of src in ELEMENTS"""
    _fields_ = ([("src", C.c_void_p), ("cache", C.c_void_p)] +
                [(n, C.c_int) for n in ("batch_size", "heads_kv", "head_size", "seq_off", "seq_size", "seq_max",
                                        "step_bs", "step_head_num", "step_seq", "step_head_size")] +
                [("no_zeroing", C.c_bool)])


class KvBatchCpyArgs(C.Structure):
    """bestla_fusion_attn_fp32_batch_cpy_kv_args_t (mha_dense.h:145-150)"""
    _fields_ = ([("src", C.c_void_p), ("dst", C.c_void_p)] +
                [(n, C.c_int) for n in ("heads_kv", "head_size", "seq_off", "seq_size", "seq_max")] +
                [("no_zeroing", C.c_bool)])


class ReorderedAttnArgs(C.Structure):
    """bestla_reordered_attn_fp32_fp32_fwd_args_t (mha_dense.h:156-171): Q / dst steps in elements, K / V strides in BYTES"""
    _fields_ = ([("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("dst", C.c_void_p),
                 ("Q_sc", C.c_float), ("K_sc", C.c_float), ("V_sc", C.c_float), ("dst_sc", C.c_float),
                 ("tmp", C.c_void_p), ("QK_scale", C.c_float), ("attn_flags", C.c_uint32)] +
                [(n, C.c_int) for n in ("batch_size", "head_num", "heads_kv", "head_size", "sl_q", "sl_kv",
                                        "Q_layout", "K_layout", "V_layout", "dst_layout",
                                        "step_q_bs", "step_q_head_num", "step_q_sl",
                                        "stride_k_bs", "stride_k_head_num", "stride_k_sl", "stride_k_head_size",
                                        "stride_v_bs", "stride_v_head_num", "stride_v_sl", "stride_v_head_size",
                                        "step_dst_bs", "step_dst_head_num", "step_dst_sl")])


class AttnArgs(C.Structure):
    """attn_fp32_fp16_fp16_fp32_fwd_args_t (mha_dense.h:66-81)"""
    _fields_ = ([("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("dst", C.c_void_p),
                 ("Q_sc", C.c_float), ("K_sc", C.c_float), ("V_sc", C.c_float), ("dst_sc", C.c_float),
                 ("tmp", C.c_void_p), ("QK_scale", C.c_float), ("attn_flags", C.c_uint32)] +
                [(n, C.c_int) for n in ("batch_size", "head_num", "heads_kv", "head_size", "sl_q", "sl_kv",
                                        "Q_layout", "K_layout", "V_layout", "dst_layout",
                                        "step_q_bs", "step_q_head_num", "step_q_sl",
                                        "step_k_bs", "step_k_head_num", "step_k_sl", "step_k_head_size",
                                        "step_v_bs", "step_v_head_num", "step_v_sl", "step_v_head_size",
                                        "step_dst_bs", "step_dst_head_num", "step_dst_sl")])


def attn_args(q_ptr, k_ptr, v_ptr, dst_ptr, bs, hn, hkv, hs, sl_q, sl_kv, qk_scale, flags=0, k_trans=False):
    """args for the tensor layouts of mha_dense_tests.cpp:232-262: q/dst [bs][sl][heads][hs], k/v [bs][sl_kv][hkv][hs]
    (k_trans: k is [bs][hkv][hs][sl_kv])."""
    a = AttnArgs()
    a.Q, a.K, a.V, a.dst = q_ptr, k_ptr, v_ptr, dst_ptr
    a.Q_sc = a.K_sc = a.V_sc = a.dst_sc = 1.0
    a.tmp = None
    a.QK_scale, a.attn_flags = qk_scale, flags
    a.batch_size, a.head_num, a.heads_kv, a.head_size, a.sl_q, a.sl_kv = bs, hn, hkv, hs, sl_q, sl_kv
    a.Q_layout = a.K_layout = a.V_layout = a.dst_layout = 0
    a.step_q_bs, a.step_q_head_num, a.step_q_sl = sl_q * hn * hs, hs, hn * hs
    a.step_k_bs = sl_kv * hkv * hs
    if k_trans:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs * sl_kv, 1, sl_kv
    else:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs, hkv * hs, 1
    a.step_v_bs, a.step_v_head_num, a.step_v_sl, a.step_v_head_size = sl_kv * hkv * hs, hs, hkv * hs, 1
    a.step_dst_bs, a.step_dst_head_num, a.step_dst_sl = sl_q * hn * hs, hs, hn * hs
    return a


def build(verbose=False):
    """Compile every HIP source for gfx950 into neural-speed_amd/libns_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises loudly when it has not been built: there is no Python/CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libns_hip.so is missing: run neural-speed_amd.build() (make -C neural-speed_amd/csrc)")
        try:  # torch bundles its own libamdhip64.so.7: let it load first so both share ONE HIP runtime
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i, u32, sz, f, b = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t, C.c_float, C.c_bool
        L.ns_hip_last_error.restype = C.c_char_p
        L.ns_hip_weight_from_blob.restype = vp
        L.ns_hip_weight_from_blob.argtypes = [vp, vp]
        L.ns_hip_weight_from_device_blob.restype = vp
        L.ns_hip_weight_from_device_blob.argtypes = [vp, sz, vp]
        L.ns_hip_cache_clear.restype = None
        L.ns_hip_weight_slice.restype = vp
        L.ns_hip_weight_slice.argtypes = [vp, i, i, i, i, vp]
        L.ns_hip_weight_free.argtypes = [vp]
        L.ns_hip_weight_free.restype = None
        L.ns_hip_weight_stream_bytes.restype = C.c_uint64
        L.ns_hip_weight_stream_bytes.argtypes = [vp]
        L.ns_hip_weight_prefetch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, vp]
        L.ns_hip_rope_f32_glm.argtypes = [vp, vp, i, i, i, i, i, i, i, f, i, vp, vp]
        L.ns_hip_expert_group_create.restype = vp
        L.ns_hip_expert_group_create.argtypes = [vp, i]
        L.ns_hip_expert_group_free.restype = None
        L.ns_hip_expert_group_free.argtypes = [vp]
        L.ns_hip_mul_mat_id.argtypes = [vp, vp, i, i, vp, vp, i, i, i, i, vp, i, vp]
        L.ns_hip_p2p_create.restype = vp
        L.ns_hip_p2p_create.argtypes = [i, i, sz, vp]
        L.ns_hip_p2p_connect.argtypes = [vp, vp]
        L.ns_hip_p2p_all_reduce_f32.argtypes = [vp, vp, sz, vp]
        L.ns_hip_p2p_error.argtypes = [vp]
        L.ns_hip_p2p_disconnect.restype = None
        L.ns_hip_p2p_disconnect.argtypes = [vp]
        L.ns_hip_p2p_destroy.restype = None
        L.ns_hip_p2p_destroy.argtypes = [vp]
        L.ns_tp_unique_id.argtypes = [vp]
        L.ns_tp_init.restype = vp
        L.ns_tp_init.argtypes = [i, i, vp, i]
        L.ns_tp_destroy.restype = None
        L.ns_tp_destroy.argtypes = [vp]
        for fn_ in (L.ns_tp_size, L.ns_tp_rank, L.ns_tp_is_master):
            fn_.argtypes = [vp]
        L.ns_tp_attach_p2p.argtypes = [vp, vp, sz]
        L.ns_tp_reduce_add.argtypes = [vp, vp, vp, sz, vp]
        L.ns_tp_broadcast.argtypes = [vp, vp, sz, vp]
        L.ns_tp_alltoall.argtypes = [vp, vp, vp, sz, vp]
        L.ns_tp_barrier.argtypes = [vp, vp]
        L.ns_tp_reduce_add_host.argtypes = [vp, vp, vp, sz]
        L.ns_tp_broadcast_host.argtypes = [vp, sz]
        L.ns_tp_alltoall_host.argtypes = [vp, vp, vp, sz]
        L.ns_tp_barrier_host.argtypes = [vp]
        L.ns_hip_set_tuning.argtypes = [C.c_char_p, i]
        L.ns_hip_weight_info.argtypes = [vp] + [vp] * 5
        L.ns_hip_f32f32_forward.argtypes = [vp, vp, vp, i, i, i, i, vp, i, vp]
        L.ns_hip_fusion_qkv_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
        L.ns_hip_fusion_ffn3_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, vp]
        L.ns_hip_f32f32_forward_h.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp, i, vp]
        L.ns_hip_fusion_qkv_forward_h.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp]
        L.ns_hip_fusion_ffn3_gateup_h.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, vp]
        L.ns_hip_f32f32_forward_x.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp, i, vp, vp]
        L.ns_hip_fusion_qkv_forward_x.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp]
        L.ns_hip_fusion_ffn3_gateup_x.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, vp, vp]
        L.ns_hip_norm_prep.argtypes = [i, i, vp, i, vp, vp, vp, i, vp]
        L.ns_hip_rope_cos_sin.argtypes = [i, i, i, f, f, f, vp, vp]
        L.ns_hip_fusion_qkv_rope_forward_x.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp]
        L.ns_hip_fusion_ffn3_forward_h.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
        L.ns_hip_fusion_ffn3_gateup.argtypes = [vp, vp, vp, vp, vp, i, i, vp]
        L.ns_hip_fusion_ffn2_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, b, vp]
        L.ns_hip_quant_pack_device.argtypes = [vp, vp, sz, sz, sz, sz, u32, u32, b, i, b, vp]
        L.ns_hip_layernormalization.argtypes = [i, i, b, f, vp, vp, vp]
        L.ns_hip_mul.argtypes = [i, i, vp, vp, i, vp, vp]
        L.ns_hip_silu_f32.argtypes = [vp, vp, sz, vp]
        L.ns_hip_dup_f32.argtypes = [vp, vp, vp, vp, vp, b, vp]
        L.ns_hip_norm_mul_h.argtypes = [i, i, b, f, vp, vp, vp, vp, vp]
        L.ns_hip_rope_qkv_append.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, f, f, f, f, C.c_longlong, C.c_longlong, vp]
        L.ns_hip_rope_f32.argtypes = [vp, vp, i, i, i, i, i, i, i, f, f, f, f, vp]
        L.ns_hip_rope_f32_yarn.argtypes = [vp, vp, i, i, i, i, i, i, i, f, f, i, f, f, f, f, vp]
        L.ns_hip_rope_f32_longrope.argtypes = [vp, vp, i, i, i, i, i, i, f, f, i, f, f, f, f, vp, f, vp]
        L.ns_hip_add.argtypes = [i, i, vp, vp, i, vp, vp]
        L.ns_hip_quantize_fp_u8_colblock.argtypes = [i, i, vp, i, vp, i, vp, i, vp, i, vp, vp]
        L.bestla_fusion_attn_workspace_size.restype = sz
        L.bestla_fusion_attn_workspace_size.argtypes = [vp]
        L.bestla_fusion_attn_fp32_fp16_fp16_fp32_support.restype = b
        L.bestla_fusion_attn_fp32_fp16_fp16_fp32_support.argtypes = [vp]
        L.bestla_reordered_attn_fp32_support.restype = b
        L.bestla_reordered_attn_fp32_support.argtypes = [vp]
        for fn in ("bestla_reordered_attn_fp32_update_k", "bestla_reordered_attn_fp32_update_v",
                   "bestla_fusion_attn_fp32_batch_cpy_k", "bestla_fusion_attn_fp32_batch_cpy_v",
                   "bestla_reordered_attn_fp32_forward"):
            getattr(L, fn).restype = None
            getattr(L, fn).argtypes = [vp]
        L.bestla_reordered_attn_fp32_batch_kv_info.restype = None
        L.bestla_reordered_attn_fp32_batch_kv_info.argtypes = [vp, vp]
        L.bestla_reordered_attn_fp32_shift_rope_k.restype = None
        L.bestla_reordered_attn_fp32_shift_rope_k.argtypes = [vp, vp, i, i, i, i, i]
        L.bestla_fusion_attn_fp32_fp16_fp16_fp32_forward.restype = None
        L.bestla_fusion_attn_fp32_fp16_fp16_fp32_forward.argtypes = [vp]
        L.ns_hip_attn_fp32_fp16_fp16_fp32_forward.argtypes = [vp, vp]
        L.ns_hip_attn_set_head_partition.argtypes = [i, i]
        L.ns_hip_attn_fp32_fp16_fp16_fp32_forward_h.argtypes = [vp, vp, vp]
        L.ns_hip_blob_validate.argtypes = [vp, sz]
        L.ns_BTLAGemmPackBSize.restype = sz
        L.ns_BTLAGemmPackBSize.argtypes = [sz, sz, sz, u32, u32, b, i, vp]
        L.ns_BTLAGemmQuantPackB.restype = b
        L.ns_BTLAGemmQuantPackB.argtypes = [vp, vp, sz, sz, sz, sz, u32, u32, b, i, b, vp]
        L.ns_BTLAGemmPackB.restype = b
        L.ns_BTLAGemmPackB.argtypes = [vp, vp, vp, vp, sz, sz, sz, sz, u32, u32, b, i, vp, vp]
        L.ns_BTLAGemmUnPackB.restype = b
        L.ns_BTLAGemmUnPackB.argtypes = [vp, vp, sz, sz, sz, vp]
        L.ns_set_pack_core.argtypes = [i]
        L.ns_set_pack_core.restype = None
        L.bestla_f32f32_get_workspace_size.restype = C.c_ulonglong
        L.bestla_f32f32_get_workspace_size.argtypes = [i, i, i, vp]
        L.bestla_f32f32_forward.restype = None
        L.bestla_f32f32_forward.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
        L.bestla_fusion_add_f32f32_support.restype = b
        L.bestla_fusion_add_f32f32_support.argtypes = [vp, i, i, i]
        L.bestla_fusion_add_f32f32_forward.restype = None
        L.bestla_fusion_add_f32f32_forward.argtypes = [vp, vp, vp, vp, i, i, i, i, i, b, vp]
        L.bestla_fusion_QKV_f32f32_support.restype = b
        L.bestla_fusion_QKV_f32f32_support.argtypes = [vp, vp, vp, i, i, i]
        L.bestla_fusion_QKV_f32f32_forward.restype = None
        L.bestla_fusion_QKV_f32f32_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        for name in ("SiLu", "Gelu_Mul"):
            fs = getattr(L, "bestla_fusion_FFN_%s_f32f32_support" % name)
            fs.restype = b
            fs.argtypes = [vp, vp, vp, i, i, i, i]
            ff = getattr(L, "bestla_fusion_FFN_%s_f32f32_forward" % name)
            ff.restype = None
            ff.argtypes = [vp] * 7 + [i, i, i, i, vp]
        L.bestla_fusion_FFN_GeLu_f32f32_support.restype = b
        L.bestla_fusion_FFN_GeLu_f32f32_support.argtypes = [vp, vp, i, i, i, i]
        L.bestla_fusion_FFN_GeLu_f32f32_forward.restype = None
        L.bestla_fusion_FFN_GeLu_f32f32_forward.argtypes = [vp] * 5 + [i, i, i, i, vp]
        L.bestla_fusion_FFN_Add_GeLu_f32f32_support.restype = b
        L.bestla_fusion_FFN_Add_GeLu_f32f32_support.argtypes = [vp, vp, i, i, i, i]
        L.bestla_fusion_FFN_Add_GeLu_f32f32_forward.restype = None
        L.bestla_fusion_FFN_Add_GeLu_f32f32_forward.argtypes = [vp] * 7 + [i, i, i, i, b, vp]
        L.bestla_unpackweight_fp32.restype = None
        L.bestla_unpackweight_fp32.argtypes = [vp, i, i, vp, i]
        L.bestla_packweight_copyattr.restype = None
        L.bestla_packweight_copyattr.argtypes = [vp, vp, i, i, i, vp]
        L.bestla_layernormalization.restype = None
        L.bestla_layernormalization.argtypes = [i, i, b, f, vp, vp]
        L.bestla_mul.restype = None
        L.bestla_mul.argtypes = [i, i, vp, vp, i, vp]
        L.bestla_add.restype = None
        L.bestla_add.argtypes = [i, i, vp, vp, i, vp]
        _lib = L
    return _lib


def last_error():
    return lib().ns_hip_last_error().decode()


def check(rc, what="call"):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, last_error()))


class Weight:
    """RAII wrapper of an ns_weight* (device-resident weight in the MI355X layout)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("weight creation failed: " + last_error())
        self.h = handle
        n, k, bits, bs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        db = C.c_uint64()
        lib().ns_hip_weight_info(self.h, C.byref(n), C.byref(k), C.byref(bits), C.byref(bs), C.byref(db))
        self.n, self.k, self.bits, self.blocksize, self.device_bytes = n.value, k.value, bits.value, bs.value, db.value
        self.stream_bytes = lib().ns_hip_weight_stream_bytes(self.h)

    @classmethod
    def from_host_blob(cls, blob_ptr, stream=None):
        return cls(lib().ns_hip_weight_from_blob(blob_ptr, stream))

    @classmethod
    def from_device_blob(cls, dev_ptr, nbytes, stream=None):
        return cls(lib().ns_hip_weight_from_device_blob(dev_ptr, nbytes, stream))

    def slice(self, n0, n1, k0, k1, stream=None):
        """tensor-parallel shard: columns [n0, n1) x rows [k0, k1), quantization preserved bit for bit"""
        return Weight(lib().ns_hip_weight_slice(self.h, n0, n1, k0, k1, stream))

    def free(self):
        if self.h:
            lib().ns_hip_weight_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
