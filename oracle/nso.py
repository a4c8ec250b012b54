"""ctypes/numpy binding of the CPU ORACLE (oracle/libns_oracle.so) and of the real reference scalar kernels
(oracle/_ref/libkernel_ref.so).  TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package (neural-speed_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# BTLA_DTYPE codes (bestla/bestla/bestla.h:38-87)
F32 = 32
F16 = 16
BF16 = 16 | (1 << 16)
S8 = 8 | (1 << 8)
S1, S2, S3, S4, S5, S6, S7 = [b | (1 << 8) for b in range(1, 8)]
F4_E2M1 = 4
F4_BNB = 4 | (1 << 16)
F4_NF4 = 4 | (2 << 16)
F8_E4M3 = 8
F8_E5M2 = 8 | (1 << 16)
F8_E8M0 = 8 | (3 << 16)
DQ8_BNB = 8 | (4 << 16)   # scale dtype: double-quantised scales (bestla.h:72)
INT_TYPES = {1: S1, 2: S2, 3: S3, 4: S4, 5: S5, 6: S6, 7: S7, 8: S8}

CORE_AVX2, CORE_AVX512F, CORE_AMX_BF16, CORE_AMX_FP16, CORE_AVX512_VNNI_KB, CORE_AVX512BW_KB, CORE_AVX_VNNI_KB, \
    CORE_AVX2_VNNI_KB, CORE_AMX_INT8_KB = range(9)
CORE_NAMES = ["avx2", "avx512f", "amx_bf16", "amx_fp16", "avx512_vnni_kb", "avx512bw_kb", "avx_vnni_kb",
              "avx2_vnni_kb", "amx_int8_kb"]


class BlobInfo(C.Structure):
    _fields_ = [("size", C.c_uint64), ("prologue_id", C.c_uint32), ("core_id", C.c_uint64),
                ("npad", C.c_int32), ("kpad", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("dtype", C.c_uint32), ("blocksize", C.c_int32), ("dq_blocksize", C.c_int32),
                ("scale_dtype", C.c_uint32), ("zp_dtype", C.c_uint32), ("red_dtype", C.c_uint32),
                ("cstep", C.c_int32), ("csize", C.c_uint64),
                ("ntile", C.c_int32), ("packrow", C.c_int32), ("comp", C.c_int32), ("isa", C.c_int32),
                ("is_asym", C.c_int32), ("has_reduce", C.c_int32), ("has_shuffle", C.c_int32),
                ("q_off", C.c_uint64), ("q_bytes", C.c_uint64), ("scale_off", C.c_uint64),
                ("scale_bytes", C.c_uint64), ("zp_off", C.c_uint64), ("zp_bytes", C.c_uint64),
                ("red_off", C.c_uint64), ("red_bytes", C.c_uint64), ("shuf_off", C.c_uint64),
                ("shuf_bytes", C.c_uint64), ("dq_off", C.c_uint64), ("dq_bytes", C.c_uint64)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def build(force=False):
    """(re)build the oracle .so files with oracle/Makefile.  Building the checker is not using it."""
    so = os.path.join(HERE, "libns_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "ns_oracle.cpp")):
        subprocess.check_call(["make", "-C", HERE, "libns_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/neural_speed/core/ne_layers.c"):
        nref = os.path.join(HERE, "_ref", "libne_ref.so")
        srcs = [os.path.join(HERE, f) for f in ("ne_ref_harness.c", "ne_ref_stubs.c")]
        if force or not os.path.exists(nref) or os.path.getmtime(nref) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", HERE, "neref"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/neural_speed/models/llama/llama.cpp"):
        lref = os.path.join(HERE, "_ref", "libne_llama_ref.so")
        root = os.path.dirname(HERE)
        srcs = [os.path.join(HERE, "llama_ref_harness.cpp"), os.path.join(HERE, "ne_ref_stubs.c"),
                os.path.join(root, "glue", "ne_bestla_hip_glue.c"), os.path.join(root, "glue", "bestla_gemm_hip.cpp"),
                os.path.join(root, "glue", "shim", "core", "layers", "bestla_common.hpp"),
                os.path.join(root, "glue", "shim", "bestla", "bestla_parallel.h")]
        if force or not os.path.exists(lref) or os.path.getmtime(lref) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", HERE, "nellama"], stdout=subprocess.DEVNULL)
        pref = os.path.join(HERE, "_ref", "llama_cpp.so")   # the reference's pybind module (needs pybind11's headers)
        if force or not os.path.exists(pref) or os.path.getmtime(pref) < max(os.path.getmtime(f) for f in srcs[1:]):
            try:
                subprocess.check_call(["make", "-C", HERE, "nepy"], stdout=subprocess.DEVNULL)
            except subprocess.CalledProcessError:
                pass
    if os.path.exists("/root/reference/neural_speed/models/llama/llama.cpp"):
        # the same model code with the reference's device switch (-DNS_SYCL) on the product's bestla_device_* set
        dref = os.path.join(HERE, "_ref", "libne_llama_dev_ref.so")
        root = os.path.dirname(HERE)
        srcs = [os.path.join(HERE, "llama_ref_harness.cpp"), os.path.join(HERE, "ne_ref_stubs.c"),
                os.path.join(root, "glue", "ne_bestla_hip_glue.c"), os.path.join(root, "glue", "ne_bestla_hip_device.c"),
                os.path.join(root, "glue", "bestla_gemm_hip.cpp")]
        if force or not os.path.exists(dref) or os.path.getmtime(dref) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", HERE, "nellamadev"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/neural_speed/core/layers/mha_dense_wrapper.h"):
        # the reference's own bestla_fusion_attn_forward_ref, cut out of its header at build time
        aref = os.path.join(HERE, "_ref", "libattn_ref.so")
        if force or not os.path.exists(aref) or os.path.getmtime(aref) < os.path.getmtime(os.path.join(HERE, "attn_shim.cpp")):
            subprocess.check_call(["make", "-C", HERE, "attnref"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/neural_speed/core/parallel_context.h"):
        # glue/parallel_context_hip.cpp against the reference header, for the multi-rank GPU test
        gref = os.path.join(HERE, "_ref", "libpc_glue.so")
        src = os.path.join(os.path.dirname(HERE), "glue", "parallel_context_hip.cpp")
        if force or not os.path.exists(gref) or os.path.getmtime(gref) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "pcglue"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/bestla/bestla/bestla_storage.h"):
        sref = os.path.join(HERE, "_ref", "libstor_ref.so")
        if force or not os.path.exists(sref) or os.path.getmtime(sref) < os.path.getmtime(os.path.join(HERE, "stor_shim.cpp")):
            subprocess.check_call(["make", "-C", HERE, "storref"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/bestla/bestla/kernel_avx512f.h"):
        aref = os.path.join(HERE, "_ref", "libkernel_avx_ref.so")
        if force or not os.path.exists(aref) or os.path.getmtime(aref) < os.path.getmtime(os.path.join(HERE, "avx_shim.cpp")):
            try:
                subprocess.check_call(["make", "-C", HERE, "avxref"], stdout=subprocess.DEVNULL)
            except subprocess.CalledProcessError:
                pass   # a compiler without the AVX512 intrinsics: the comparison test skips
    if os.path.exists("/root/reference/bestla/bestla/bestla_prologue_b.h"):
        pref = os.path.join(HERE, "_ref", "libpack_ref.so")
        srcs = [os.path.join(HERE, f) for f in ("pack_shim.cpp", "standins/kernel_jit.h", "standins/xbyak/xbyak_util.h")]
        if force or not os.path.exists(pref) or os.path.getmtime(pref) < max(os.path.getmtime(f) for f in srcs):
            try:
                subprocess.check_call(["make", "-C", HERE, "packref"], stdout=subprocess.DEVNULL)
            except subprocess.CalledProcessError:
                pass   # a compiler without the AVX512 intrinsics: the comparison test skips
    if os.path.exists("/root/reference/bestla/bestla/kernel_ref.h"):
        ref = os.path.join(HERE, "_ref", "libkernel_ref.so")
        if force or not os.path.exists(ref) or os.path.getmtime(ref) < os.path.getmtime(os.path.join(HERE, "ref_shim.cpp")):
            subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(HERE, "libns_oracle.so"))
        _lib.nso_pack_size.restype = C.c_size_t
        _lib.nso_pack_size_gidx.restype = C.c_size_t
        _lib.nso_qbytes.restype = C.c_size_t
        _lib.nso_core_id.restype = C.c_uint64
        _lib.nso_bf16_to_f32.restype = C.c_float
        _lib.nso_f16_to_f32.restype = C.c_float
        _lib.nso_f32_to_bf16.restype = C.c_uint16
        _lib.nso_f32_to_f16.restype = C.c_uint16
        _lib.nso_f4_unpack.restype = C.c_float
        _lib.nso_f8_to_f32.restype = C.c_float
        _lib.nso_f8_to_f32.argtypes = [C.c_uint32, C.c_int]
        _lib.nso_f8_quantize.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float]
        _lib.nso_gelu.restype = C.c_float
        _lib.nso_silu.restype = C.c_float
        _lib.nso_bf16_to_f32.argtypes = [C.c_uint16]
        _lib.nso_f16_to_f32.argtypes = [C.c_uint16]
        _lib.nso_f32_to_bf16.argtypes = [C.c_float]
        _lib.nso_f32_to_f16.argtypes = [C.c_float]
        _lib.nso_f4_unpack.argtypes = [C.c_uint32, C.c_int]
        _lib.nso_f4_quantize.argtypes = [C.c_uint32, C.c_float]
        _lib.nso_gelu.argtypes = [C.c_float]
        _lib.nso_silu.argtypes = [C.c_float]
    return _lib


def ref():
    """The real reference kernels; None when oracle/_ref was never built (then tests relying on it skip)."""
    global _ref
    if _ref is None:
        build()
        p = os.path.join(HERE, "_ref", "libkernel_ref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        _ref.ref_f4_unpack.restype = C.c_float
        _ref.ref_f8_to_f32.restype = C.c_float
        _ref.ref_f8_to_f32.argtypes = [C.c_uint32, C.c_int]
        _ref.ref_lut.restype = C.c_float
        _ref.ref_bf16_to_f32.restype = C.c_float
        _ref.ref_f16_to_f32.restype = C.c_float
        _ref.ref_f32_to_bf16.restype = C.c_uint16
        _ref.ref_f32_to_f16.restype = C.c_uint16
        _ref.ref_postop.restype = C.c_float
        _ref.ref_f4_quantize.argtypes = [C.c_uint32, C.c_float]
        _ref.ref_f32_to_bf16.argtypes = [C.c_float]
        _ref.ref_f32_to_f16.argtypes = [C.c_float]
        _ref.ref_bf16_to_f32.argtypes = [C.c_uint16]
        _ref.ref_f16_to_f32.argtypes = [C.c_uint16]
        _ref.ref_cast_f32_s8.argtypes = [C.c_float]
        _ref.ref_cast_f32_u8.argtypes = [C.c_float]
        _ref.ref_postop.argtypes = [C.c_float, C.c_int]
    return _ref


_stor = None


def storref():
    """The reference's own blob container classes (bestla_storage.h compiled into oracle/_ref/libstor_ref.so by
    `make -C oracle storref`, see oracle/stor_shim.cpp); None when it was never built."""
    global _stor
    if _stor is None:
        p = os.path.join(HERE, "_ref", "libstor_ref.so")
        if not os.path.exists(p) and os.path.exists("/root/reference/bestla/bestla/bestla_storage.h"):
            subprocess.check_call(["make", "-C", HERE, "storref"], stdout=subprocess.DEVNULL)
        if not os.path.exists(p):
            return None
        _stor = C.CDLL(p)
        _stor.stor_assign.restype = C.c_uint64
        _stor.stor_assign.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _stor.stor_deserialize.argtypes = [C.c_void_p, C.c_void_p]
    return _stor


_neref = None


def neref(product_lib_path=None):
    """The reference's own graph executor (ne_layers.c compiled into oracle/_ref/libne_ref.so) behind the flat entry
    points of ne_ref_harness.c; None when it was never built.  product_lib_path: load that library with RTLD_GLOBAL
    first, so that the graph's bestla_* calls land in it (the drop-in test); must be given on the FIRST call."""
    global _neref
    if _neref is None:
        build()
        p = os.path.join(HERE, "_ref", "libne_ref.so")
        if not os.path.exists(p):
            return None
        if product_lib_path:
            C.CDLL(product_lib_path, mode=C.RTLD_GLOBAL)
        _neref = C.CDLL(p)
        _neref.provider = product_lib_path
        f, i, vp = C.c_float, C.c_int, C.c_void_p
        _neref.neref_rope.argtypes = [vp, vp, i, i, i, i, i, i, i, i, f, f, i, f, f, f, f, vp, vp, f]
        _neref.neref_mul_mat.argtypes = [vp, vp, C.c_size_t, vp, i, i, i]
        _neref.neref_ffn_silu.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, i, i, i]
        _neref.neref_mul_qkv.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, i, i, i]
        _neref.neref_norm.argtypes = [vp, vp, i, i, f, i]
        _neref.neref_flash_attn.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, f, C.c_uint]
        _neref.neref_fused.argtypes = [i, vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, i, i, i]
        _neref.neref_decoder_layer.argtypes = [vp, vp, i, i, i, i, f, f, vp, vp] + [vp, C.c_size_t] * 7
        _neref.neref_mul_mat_id.argtypes = [vp, vp, vp, i, vp, i, i, vp, i, i, i]
        _neref.neref_ffn_id.argtypes = [i, vp, vp, vp, i, vp, i, i, vp, i, i, i]
        _neref.neref_attn_unfused.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, i]
    elif product_lib_path and _neref.provider != product_lib_path:
        raise RuntimeError("libne_ref.so is already loaded without (or with another) bestla_* provider")
    return _neref


def neref_rope(x, n_past, n_dims, mode, freq_base=10000.0, freq_scale=1.0, prompt_size=0, n_orig_ctx=0, ext_factor=0.0,
               attn_factor=1.0, beta_fast=0.0, beta_slow=0.0, n_padding=None, factors=None, scale_factor=0.0):
    """RoPE through the reference graph (ne_rope_impl + ne_graph_compute).  freq_scale is the EFFECTIVE scale (what the
    forward multiplies by); the graph parameter is its reciprocal (ne_layers.c:9262), so use powers of two."""
    x = np.ascontiguousarray(x, np.float32)
    b, s, h, hs = x.shape
    out = np.zeros_like(x)
    pad = None if n_padding is None else np.ascontiguousarray(n_padding, np.int32)
    fac = None if factors is None else np.ascontiguousarray(factors, np.float32)
    rc = neref().neref_rope(ptr(x), ptr(out), b, s, h, hs, n_past, n_dims, mode, prompt_size, freq_base, 1.0 / freq_scale,
                            n_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow, ptr(pad), ptr(fac), scale_factor)
    assert rc == 0
    return out


def neref_norm(x, eps, is_rms):
    """ne_rms_norm / ne_norm of the reference graph on fp32 [rows][cols]"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    assert neref().neref_norm(ptr(x), ptr(out), x.shape[0], x.shape[1], eps, 1 if is_rms else 0) == 0
    return out


def neref_flash_attn(q, k, v, qk_scale, flags):
    """the reference graph's fused-attention node (ne_flash_attn); needs a bestla_* provider.  attn_ref's layouts."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float16)
    v = np.ascontiguousarray(v, np.float16)
    bs, sl_q, hn, hs = q.shape
    sl_kv, hkv = v.shape[1], v.shape[2]
    out = np.zeros_like(q)
    assert neref().neref_flash_attn(ptr(q), ptr(k), ptr(v), ptr(out), bs, hn, hkv, hs, sl_q, sl_kv, qk_scale, flags) == 0
    return out


def neref_reordered_attn(q, kall, vall, n_ctx, qk_scale, flags, split=False):
    """the reference graph's library-managed kv-cache nodes (update_k / update_v / flash_attn over
    NE_TYPE_BTLA caches, llama.cpp:496-571); needs a bestla_* provider.  q [bs][sl_q][heads][hs] fp32;
    kall / vall [bs][sl_kv][heads_kv][hs] fp32 (what the graph appends)."""
    q = np.ascontiguousarray(q, np.float32)
    kall = np.ascontiguousarray(kall, np.float32)
    vall = np.ascontiguousarray(vall, np.float32)
    bs, sl_q, hn, hs = q.shape
    sl_kv, hkv = vall.shape[1], vall.shape[2]
    out = np.zeros_like(q)
    f = neref().neref_reordered_attn
    f.argtypes = [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_float, C.c_uint, C.c_int]
    rc = f(ptr(q), ptr(kall), ptr(vall), ptr(out), bs, hn, hkv, hs, sl_q, sl_kv, n_ctx, qk_scale, flags, int(split))
    assert rc == 0, rc
    return out


def rope_shift_f16_ref(k16, shift_n, n_keep, freq_base=10000.0):
    """restatement of the shift-RoPE of a K cache (ne_compute_forward_rope_f16's is_shift branch, ne_layers.c:9494-9530; the
    BTLA-cache twin computes the same cos / sin table, :9640-9650, and hands it to bestla_reordered_attn_fp32_shift_rope_k).
    PARITY UNPINNED: the reference cannot execute either branch (parameter-tensor size assert, see ne_ref_harness.c).
    k16 fp16 [bs][seq][heads][hs] -> (rotated copy, the fp16 {cos, sin} table).  One angle set for -shift_n positions, cos / sin ROUNDED TO
    FP16, fp32 arithmetic on adjacent pairs, result rounded to fp16; rows < n_keep untouched."""
    k = np.asarray(k16, np.float16).copy()
    hs = k.shape[-1]
    theta_scale = np.float32(np.power(np.float32(freq_base), np.float32(-2.0) / np.float32(hs)))
    theta = np.float32(-shift_n)
    cs = np.zeros(hs, np.float16)
    for i0 in range(0, hs, 2):
        cs[i0], cs[i0 + 1] = np.float16(np.cos(theta, dtype=np.float32)), np.float16(np.sin(theta, dtype=np.float32))
        theta = np.float32(theta * theta_scale)
    c, s = cs[0::2].astype(np.float32), cs[1::2].astype(np.float32)
    x0, x1 = k[:, n_keep:, :, 0::2].astype(np.float32), k[:, n_keep:, :, 1::2].astype(np.float32)
    k[:, n_keep:, :, 0::2] = (x0 * c - x1 * s).astype(np.float16)
    k[:, n_keep:, :, 1::2] = (x1 * c + x0 * s).astype(np.float16)
    return k, cs


def neref_attn_unfused(q, k, v, qk_scale, causal):
    """attention through the reference's UNFUSED graph (mul_mat -> scale -> diag_mask_inf -> soft_max -> mul_mat), fp32.
    q [1][sl_q][heads][hs], k / v fp16 [1][sl_kv][heads_kv][hs] (attn_ref's layouts) -> dst [1][sl_q][heads][hs]"""
    q = np.ascontiguousarray(q, np.float32)
    assert q.shape[0] == 1
    _, sl_q, hn, hs = q.shape
    sl_kv, hkv = v.shape[1], v.shape[2]
    qh = np.ascontiguousarray(q[0].transpose(1, 0, 2))                       # [heads][sl_q][hs]
    kh = np.ascontiguousarray(np.asarray(k[0], np.float32).transpose(1, 0, 2))  # [heads_kv][sl_kv][hs]
    vh = np.ascontiguousarray(np.asarray(v[0], np.float32).transpose(1, 0, 2))
    out = np.zeros_like(qh)
    rc = neref().neref_attn_unfused(ptr(qh), ptr(kh), ptr(vh), ptr(out), hn, hkv, hs, sl_q, sl_kv, qk_scale, 1 if causal else 0)
    assert rc == 0
    return np.ascontiguousarray(out.transpose(1, 0, 2))[None]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def aligned_bytes(nbytes, align=64, fill=0):
    """uint8 buffer whose data pointer is `align`-byte aligned (blob bytes depend on the base address mod 64)."""
    raw = np.full(nbytes + align, fill, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


def is_int_type(qtype):
    return ((qtype >> 8) & 0xff) == 1


def is_f8_type(qtype):
    return qtype in (F8_E4M3, F8_E5M2)


def nblk(k, blocksize):
    return (k + blocksize - 1) // blocksize


# ------------------------------------------------------------------ oracle wrappers
def quantize(w_kn, blocksize, qtype, asym=False, stype=F32):
    """w_kn: fp32 [K][N].  returns (q int8 [K][N], scales f32 [nblk][N], zps int8 [nblk][N] | None).
    `stype` matters for the fp8 weight types only (F8_E8M0: the scales are shared exponents)."""
    w_kn = np.ascontiguousarray(w_kn, dtype=np.float32)
    k, n = w_kn.shape
    bs = k if blocksize <= 0 else blocksize
    q = np.zeros((k, n), np.int8)
    sc = np.zeros((nblk(k, bs), n), np.float32)
    if is_int_type(qtype):
        zp = np.zeros((nblk(k, bs), n), np.int8) if asym else None
        rc = lib().nso_quantize_int_rowblock(ptr(w_kn), ptr(q), k, n, n, n, ptr(sc), ptr(zp), bs, C.c_uint32(qtype))
    elif is_f8_type(qtype):
        zp = None
        rc = lib().nso_quantize_f8_rowblock(ptr(w_kn), ptr(q), k, n, n, n, ptr(sc), bs, C.c_uint32(qtype),
                                            C.c_uint32(stype))
    else:
        zp = None
        rc = lib().nso_quantize_f4_rowblock(ptr(w_kn), ptr(q), k, n, n, n, ptr(sc), bs, C.c_uint32(qtype))
    assert rc == 0
    return q, sc, zp


def pack_size(n, k, blocksize, qtype, stype, asym, core):
    return lib().nso_pack_size(n, k, blocksize, C.c_uint32(qtype), C.c_uint32(stype), int(asym), core)


def quant_pack(w, blocksize, qtype, stype=BF16, asym=False, core=CORE_AVX512_VNNI_KB, is_trans=True, fill=0):
    """BTLAGemmQuantPackB.  w is [N][K] when is_trans (torch layout) else [K][N].  returns the blob (uint8, 64-aligned)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, k = w.shape if is_trans else w.shape[::-1]
    size = pack_size(n, k, blocksize, qtype, stype, asym, core)
    assert size > 0, "unsupported combination"
    blob = aligned_bytes(size, fill=fill)
    rc = lib().nso_quant_pack(ptr(blob), ptr(w), n, k, w.shape[1], blocksize, C.c_uint32(qtype), C.c_uint32(stype),
                              int(asym), core, int(is_trans))
    assert rc == 0
    return blob


def pack_q(q_kn, scales, zps, blocksize, qtype, stype=BF16, core=CORE_AVX512_VNNI_KB, fill=0, g_idx=None):
    """BTLAGemmPackB.  g_idx (int32 [K], group of every input channel) makes an activation-shuffle blob; q_kn's rows
    must then already be in group-sorted order (see sort_rows_by_group)."""
    q_kn = np.ascontiguousarray(q_kn, dtype=np.int8)
    k, n = q_kn.shape
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    asym = zps is not None
    if asym:
        zps = np.ascontiguousarray(zps, dtype=np.int8)
    if g_idx is not None:
        g_idx = np.ascontiguousarray(g_idx, dtype=np.int32)
        size = lib().nso_pack_size_gidx(n, k, blocksize, C.c_uint32(qtype), C.c_uint32(stype), int(asym), core)
        assert size > 0
        blob = aligned_bytes(size, fill=fill)
        rc = lib().nso_pack_q_gidx(ptr(blob), ptr(q_kn), n, ptr(scales), ptr(zps), n, k, blocksize, C.c_uint32(qtype),
                                   C.c_uint32(stype), int(asym), core, ptr(g_idx))
        assert rc == 0
        return blob
    size = pack_size(n, k, blocksize, qtype, stype, asym, core)
    assert size > 0
    blob = aligned_bytes(size, fill=fill)
    rc = lib().nso_pack_q(ptr(blob), ptr(q_kn), n, ptr(scales), ptr(zps), n, k, blocksize, C.c_uint32(qtype),
                          C.c_uint32(stype), int(asym), core)
    assert rc == 0
    return blob


def sort_rows_by_group(x_kn, g_idx, blocksize):
    """what the reference converter does to GPTQ act-order weights before packing (convert/common.py:667-681): row i
    goes to position g_idx[i] * group_size + (how many rows of that group came before it)."""
    out = np.empty_like(x_kn)
    count = {}
    for i, g in enumerate(np.asarray(g_idx).tolist()):
        c = count.get(g, 0)
        out[g * blocksize + c] = x_kn[i]
        count[g] = c + 1
    return out


def parse(blob):
    bi = BlobInfo()
    rc = lib().nso_blob_parse(ptr(blob), C.byref(bi))
    assert rc == 0, "not a BTLA k-block blob"
    return bi


def unpack_fp32(blob):
    bi = parse(blob)
    out = np.zeros((bi.k, bi.n), np.float32)
    assert lib().nso_unpack_fp32(ptr(blob), ptr(out), bi.n) == 0
    return out


def unpack_canonical(blob):
    bi = parse(blob)
    q = np.zeros((bi.k, bi.n), np.int8)
    nb = nblk(bi.k, bi.blocksize)
    sc = np.zeros((nb, bi.n), np.float32)
    zp = np.zeros((nb, bi.n), np.int8)
    assert lib().nso_unpack_canonical(ptr(blob), ptr(q), ptr(sc), ptr(zp)) == 0
    return q, sc, zp


def gemm_f64(a, blob, a16=False):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bi = parse(blob)
    m = a.shape[0]
    c = np.zeros((m, bi.n), np.float64)
    fn = lib().nso_gemm_f64_a16 if a16 else lib().nso_gemm_f64
    assert fn(ptr(a), a.shape[1], ptr(blob), ptr(c), bi.n, m) == 0
    return c


def gemm_f64_pair(a, blob):
    """(gemm_f64(a, blob), gemm_f64(a, blob, a16=True)) from one unpack of the blob"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    bi = parse(blob)
    m = a.shape[0]
    c = np.zeros((m, bi.n), np.float64)
    c16 = np.zeros((m, bi.n), np.float64)
    assert lib().nso_gemm_f64_pair(ptr(a), a.shape[1], ptr(blob), ptr(c), ptr(c16), bi.n, m) == 0
    return c, c16


def gemv_f32(a, blob, nthreads=0):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bi = parse(blob)
    m = a.shape[0]
    c = np.zeros((m, bi.n), np.float32)
    assert lib().nso_gemv_f32(ptr(a), a.shape[1], ptr(blob), ptr(c), bi.n, m, nthreads) == 0
    return c


def gemv_u8s8(a, blob, nthreads=0):
    """the reference's default decode numerics (u8 activations x s8/s4 weights), streamed from the packed blob"""
    a = np.ascontiguousarray(a, np.float32)
    bi = parse(blob)
    m = a.shape[0]
    c = np.zeros((m, bi.n), np.float32)
    assert lib().nso_gemv_u8s8_f32(ptr(a), a.shape[1], ptr(blob), ptr(c), bi.n, m, nthreads) == 0
    return c


_avx = None


def avxref():
    """oracle/_ref/libkernel_avx_ref.so (the reference's AVX512 / AVX2 kernels, oracle/avx_shim.cpp) or None when it is
    not built or this CPU lacks AVX512-VNNI"""
    global _avx
    if _avx is None:
        _avx = False
        so = os.path.join(HERE, "_ref", "libkernel_avx_ref.so")
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            flags = ""
        if os.path.exists(so) and all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512_vnni")):
            _avx = C.CDLL(so)
            _avx.avx512vnni_gemv_4bit_u8s8.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_void_p]
    return _avx or None


def gemv_u8s8_avx512vnni(a, blob, nthreads=0, _scratch={}):
    """one row through the REFERENCE's own decode kernels on this CPU: avx512f::quantize_fp_u8_colblock +
    avx512f::vnni::gemv_4bit_u8s8_fp32<ScaleT, 48, 1> per 48-column tile (int4 blobs of the AVX512_VNNI k-block core, fp32 or
    bf16 scales), tiles over OpenMP threads.  -> [1][n] fp32"""
    bi = parse(blob)
    assert a.shape[0] == 1 and bi.ntile == 48 and bi.packrow == 4 and bi.scale_dtype in (F32, BF16)
    c = _scratch.get(("c", bi.npad))
    if c is None:
        c = _scratch[("c", bi.npad)] = np.zeros((1, bi.npad), np.float32)
    sc = _scratch.get(("s", bi.k))
    if sc is None:
        sc = _scratch[("s", bi.k)] = np.zeros(bi.k + 128 + 8 * nblk(bi.k, bi.blocksize), np.uint8)
    base = blob.ctypes.data
    rc = avxref().avx512vnni_gemv_4bit_u8s8(ptr(a), base + bi.q_off, base + bi.scale_off, int(bi.scale_dtype == BF16),
                                            base + bi.zp_off if bi.is_asym else None, bi.cstep, bi.kpad, bi.n, bi.k, bi.blocksize,
                                            ptr(c), nthreads, ptr(sc))
    assert rc == 0
    return c[:, :bi.n]


def _blob_table(blobs):
    ptrs = (C.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    sizes = (C.c_size_t * len(blobs))(*[b.size for b in blobs])
    return ptrs, sizes


def neref_mul_mat_id(a, blobs, ids, id_):
    """the reference's ne_mul_mat_id node over BTLA expert weights; a [m][k] fp32, ids [m][n_ids] int32"""
    a = np.ascontiguousarray(a, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    bi = parse(blobs[0])
    out = np.zeros((a.shape[0], bi.n), np.float32)
    ptrs, sizes = _blob_table(blobs)
    assert neref().neref_mul_mat_id(ptr(a), ptrs, sizes, len(blobs), ptr(ids), ids.shape[1], id_, ptr(out), a.shape[0], bi.n,
                                    a.shape[1]) == 0
    return out


def neref_ffn_id(a, gate, down, up, ids, id_, gelu=False):
    """ne_mul_id_ffn_silu / _gelu over BTLA expert weights (the expert of token row 0 serves every row)"""
    a = np.ascontiguousarray(a, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    d, ff = a.shape[1], parse(gate[0]).n
    out = np.zeros((a.shape[0], d), np.float32)
    ptrs, sizes = _blob_table(list(gate) + list(down) + list(up))
    assert neref().neref_ffn_id(int(gelu), ptr(a), ptrs, sizes, len(gate), ptr(ids), ids.shape[1], id_, ptr(out), a.shape[0],
                                d, ff) == 0
    return out


class AttnArgs(C.Structure):
    """nso_attn_args (ns_oracle.h)"""
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("dst", C.c_void_p),
                ("q_sc", C.c_float), ("k_sc", C.c_float), ("v_sc", C.c_float), ("dst_sc", C.c_float),
                ("qk_scale", C.c_float), ("flags", C.c_uint32),
                ("batch_size", C.c_int), ("head_num", C.c_int), ("heads_kv", C.c_int), ("head_size", C.c_int),
                ("sl_q", C.c_int), ("sl_kv", C.c_int),
                ("step_q_bs", C.c_longlong), ("step_q_head_num", C.c_longlong), ("step_q_sl", C.c_longlong),
                ("step_k_bs", C.c_longlong), ("step_k_head_num", C.c_longlong), ("step_k_sl", C.c_longlong),
                ("step_k_head_size", C.c_longlong),
                ("step_v_bs", C.c_longlong), ("step_v_head_num", C.c_longlong), ("step_v_sl", C.c_longlong),
                ("step_dst_bs", C.c_longlong), ("step_dst_head_num", C.c_longlong), ("step_dst_sl", C.c_longlong)]


_attn = None


def attnref():
    """oracle/_ref/libattn_ref.so: the reference's OWN bestla_fusion_attn_forward_ref<float, fp16, fp16, float> (oracle/Makefile
    attnref); None where it was never built"""
    global _attn
    if _attn is None:
        so = os.path.join(HERE, "_ref", "libattn_ref.so")
        if not os.path.exists(so) and os.path.isdir("/root/reference"):
            subprocess.call(["make", "-C", HERE, "attnref"], stdout=subprocess.DEVNULL)
        _attn = C.CDLL(so) if os.path.exists(so) else False
        if _attn:
            for f in ("attnref_flag_causal", "attnref_flag_alibi8", "attnref_flag_prefer_fp32"):
                getattr(_attn, f).restype = C.c_uint
    return _attn or None


def _attn_args(q, k, v, dst, qk_scale, flags, k_trans, scales):
    bs, sl_q, hn, hs = q.shape
    sl_kv, hkv = v.shape[1], v.shape[2]
    a = AttnArgs()
    a.q, a.k, a.v, a.dst = q.ctypes.data, k.ctypes.data, v.ctypes.data, dst.ctypes.data
    a.q_sc, a.k_sc, a.v_sc, a.dst_sc = scales
    a.qk_scale, a.flags = qk_scale, flags
    a.batch_size, a.head_num, a.heads_kv, a.head_size, a.sl_q, a.sl_kv = bs, hn, hkv, hs, sl_q, sl_kv
    a.step_q_bs, a.step_q_head_num, a.step_q_sl = sl_q * hn * hs, hs, hn * hs
    a.step_k_bs = sl_kv * hkv * hs
    if k_trans:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs * sl_kv, 1, sl_kv
    else:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs, hkv * hs, 1
    a.step_v_bs, a.step_v_head_num, a.step_v_sl = sl_kv * hkv * hs, hs, hkv * hs
    a.step_dst_bs, a.step_dst_head_num, a.step_dst_sl = sl_q * hn * hs, hs, hn * hs
    return a


def attn_reference(q, k, v, qk_scale, causal=False, alibi8=False, prefer_fp32=False, k_trans=False, scales=(1.0, 1.0, 1.0, 1.0)):
    """The reference's own function (attnref()) on attn_ref's layouts.  Without prefer_fp32 it rounds Q, K, P and V to bf16
    when K is not transposed (IS_BF16_GEMM, mha_dense_wrapper.h:1389-1394)."""
    L = attnref()
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float16)
    v = np.ascontiguousarray(v, np.float16)
    dst = np.zeros_like(q)
    flags = (L.attnref_flag_causal() if causal else 0) | (L.attnref_flag_alibi8() if alibi8 else 0) | \
            (L.attnref_flag_prefer_fp32() if prefer_fp32 else 0)
    a = _attn_args(q, k, v, dst, qk_scale, flags, k_trans, scales)
    assert L.attnref_forward_f32_f16_f16_f32(C.byref(a)) == 0
    return dst


def attn_ref(q, k, v, qk_scale, flags=0, k_trans=False, bf16_gemm=False, scales=(1.0, 1.0, 1.0, 1.0), ref_exp=False):
    """q fp32 [bs][sl_q][heads][hs]; k, v fp16 [bs][sl_kv][heads_kv][hs] (k_trans: k is [bs][heads_kv][hs][sl_kv]).
    Returns dst fp32 [bs][sl_q][heads][hs] — the tensor layouts of mha_dense_tests.cpp:232-262."""
    q = np.ascontiguousarray(q, np.float32)
    k = np.ascontiguousarray(k, np.float16)
    v = np.ascontiguousarray(v, np.float16)
    bs, sl_q, hn, hs = q.shape
    sl_kv, hkv = v.shape[1], v.shape[2]
    dst = np.zeros_like(q)
    a = AttnArgs()
    a.q, a.k, a.v, a.dst = q.ctypes.data, k.ctypes.data, v.ctypes.data, dst.ctypes.data
    a.q_sc, a.k_sc, a.v_sc, a.dst_sc = scales
    a.qk_scale, a.flags = qk_scale, flags
    a.batch_size, a.head_num, a.heads_kv, a.head_size, a.sl_q, a.sl_kv = bs, hn, hkv, hs, sl_q, sl_kv
    a.step_q_bs, a.step_q_head_num, a.step_q_sl = sl_q * hn * hs, hs, hn * hs
    a.step_k_bs = sl_kv * hkv * hs
    if k_trans:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs * sl_kv, 1, sl_kv
    else:
        a.step_k_head_num, a.step_k_sl, a.step_k_head_size = hs, hkv * hs, 1
    a.step_v_bs, a.step_v_head_num, a.step_v_sl = sl_kv * hkv * hs, hs, hkv * hs
    a.step_dst_bs, a.step_dst_head_num, a.step_dst_sl = sl_q * hn * hs, hs, hn * hs
    assert lib().nso_attn_ref(C.byref(a), (1 if bf16_gemm else 0) | (2 if ref_exp else 0)) == 0
    return dst


def rope_f32(x, n_past, n_dims, mode, freq_base=10000.0, freq_scale=1.0, attn_factor=1.0):
    """x fp32 [batch][seq][heads][head_size] -> rotated copy (ne_compute_forward_rope_f32 restatement)."""
    x = np.ascontiguousarray(x, np.float32)
    b, s, h, hs = x.shape
    out = np.zeros_like(x)
    rc = lib().nso_rope_f32(ptr(x), ptr(out), b, s, h, hs, n_past, n_dims, mode, C.c_float(freq_base), C.c_float(freq_scale),
                            C.c_float(attn_factor))
    assert rc == 0
    return out


def rope_f32_glm(x, n_past, n_dims, mode, freq_base, prompt_size, n_padding):
    """GLM branch of ne_compute_forward_rope_f32 (mode & 4); rows the reference leaves untouched keep x's values"""
    x = np.ascontiguousarray(x, np.float32)
    b, s, h, hs = x.shape
    out = x.copy()
    pad = np.ascontiguousarray(n_padding, np.int32)
    assert pad.shape == (b,)
    rc = lib().nso_rope_f32_glm(ptr(x), ptr(out), b, s, h, hs, n_past, n_dims, mode, C.c_float(freq_base), prompt_size,
                                ptr(pad))
    assert rc == 0
    return out


def rope_f32_yarn(x, n_past, n_dims, mode, freq_base, freq_scale, n_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow):
    x = np.ascontiguousarray(x, np.float32)
    b, s, h, hs = x.shape
    out = np.zeros_like(x)
    rc = lib().nso_rope_f32_yarn(ptr(x), ptr(out), b, s, h, hs, n_past, n_dims, mode, C.c_float(freq_base),
                                 C.c_float(freq_scale), n_orig_ctx, C.c_float(ext_factor), C.c_float(attn_factor),
                                 C.c_float(beta_fast), C.c_float(beta_slow))
    assert rc == 0
    return out


def rope_f32_longrope(x, n_past, n_dims, freq_base, freq_scale, n_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow,
                      factors, scale_factor):
    x = np.ascontiguousarray(x, np.float32)
    factors = np.ascontiguousarray(factors, np.float32)
    b, s, h, hs = x.shape
    out = np.zeros_like(x)
    rc = lib().nso_rope_f32_longrope(ptr(x), ptr(out), b, s, h, hs, n_past, n_dims, C.c_float(freq_base), C.c_float(freq_scale),
                                     n_orig_ctx, C.c_float(ext_factor), C.c_float(attn_factor), C.c_float(beta_fast),
                                     C.c_float(beta_slow), ptr(factors), C.c_float(scale_factor))
    assert rc == 0
    return out


def gemm_u8s8(a, blob):
    a = np.ascontiguousarray(a, dtype=np.float32)
    bi = parse(blob)
    m = a.shape[0]
    c = np.zeros((m, bi.n), np.float32)
    assert lib().nso_gemm_u8s8_f32(ptr(a), a.shape[1], ptr(blob), ptr(c), bi.n, m) == 0
    return c


def compress(codes, qtype):
    codes = np.ascontiguousarray(codes, dtype=np.int8).ravel()
    nb = lib().nso_qbytes(C.c_size_t(codes.size), C.c_uint32(qtype))
    out = np.zeros(nb, np.uint8)
    lib().nso_compress(ptr(codes), ptr(out), C.c_size_t(codes.size), C.c_uint32(qtype))
    return out


def decompress(packed, size, qtype):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.zeros(size, np.int8)
    lib().nso_decompress(ptr(packed), ptr(out), C.c_size_t(size), C.c_uint32(qtype))
    return out


def rel_l2(y, yref):
    """the reference's own metric: ||a-b|| / ||b||  (tests/test_python_api.py:27-33 cmpData diff2)"""
    y = np.asarray(y, np.float64)
    yref = np.asarray(yref, np.float64)
    return float(np.linalg.norm(y - yref) / max(np.linalg.norm(yref), 1e-30))
