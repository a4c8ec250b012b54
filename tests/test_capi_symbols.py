"""CPU-side checks of the drop-in boundary: the C-ABI library loads (no GPU needed) and exports every symbol
include/ns_bestla.h declares; size functions and error behaviour work without a device."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ns_bestla.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b((?:bestla|ns)_[A-Za-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.startswith("ns_comp") and not n.startswith("ns_core")))


def test_every_declared_symbol_is_exported(L):
    names = _declared_functions()
    assert len(names) > 35
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_reference_surface_names_present(L):
    # ne_bestla.h:21-83 (the part of the reference surface that does not take ne_tensor structs)
    for n in ["bestla_init", "bestla_timer", "bestla_set_threads", "bestla_get_thread_handle",
              "bestla_f32f32_get_workspace_size", "bestla_f32f32_forward", "bestla_fusion_add_f32f32_support",
              "bestla_fusion_add_f32f32_forward", "bestla_fusion_QKV_f32f32_get_workspace_size",
              "bestla_fusion_QKV_f32f32_support", "bestla_fusion_QKV_f32f32_forward",
              "bestla_fusion_FFN_f32f32_get_workspace_size", "bestla_fusion_FFN_Gelu_Mul_f32f32_support",
              "bestla_fusion_FFN_Gelu_Mul_f32f32_forward", "bestla_fusion_FFN_SiLu_f32f32_support",
              "bestla_fusion_FFN_SiLu_f32f32_forward", "bestla_fusion_FFN_GeLu_f32f32_support",
              "bestla_fusion_FFN_GeLu_f32f32_forward", "bestla_fusion_FFN_Add_GeLu_f32f32_support",
              "bestla_fusion_FFN_Add_GeLu_f32f32_forward", "bestla_unpackweight_fp32", "bestla_packweight_copyattr",
              "bestla_layernormalization", "bestla_mul", "bestla_add"]:
        assert hasattr(L, n), n


def test_workspace_sizes_match_reference_formula(L):
    # inner_product.cpp:20-25: m * padto(k,128) * 4
    assert L.bestla_f32f32_get_workspace_size(3, 4096, 4000, None) == 3 * 4096 * 4
    assert L.bestla_f32f32_get_workspace_size(1, 11008, 11008, None) == 11008 * 4


def test_pack_size_matches_oracle(L, pkg, nso):
    # the blob size is host logic: must agree with the oracle for every format/core
    cases = [(pkg.S4, pkg.BF16, False, pkg.COMP_INT8, 32, nso.CORE_AVX512_VNNI_KB),
             (pkg.S4, pkg.F32, True, pkg.COMP_INT8, 128, nso.CORE_AMX_INT8_KB),
             (pkg.S8, pkg.BF16, False, pkg.COMP_F32, 32, nso.CORE_AVX512F),
             (pkg.F4_NF4, pkg.BF16, False, pkg.COMP_BF16, 128, nso.CORE_AMX_BF16),
             (pkg.INT_TYPES[3], pkg.F16, False, pkg.COMP_F32, 64, nso.CORE_AVX512F),
             (pkg.INT_TYPES[7], pkg.F32, True, pkg.COMP_INT8, 32, nso.CORE_AVX512_VNNI_KB)]
    for qt, st, asym, comp, bs, core in cases:
        for n, k in [(4096, 4096), (100, 96), (48, 128), (11008, 4096)]:
            if k % bs:
                continue
            mine = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st, asym, comp, None)
            assert mine == nso.pack_size(n, k, bs, qt, st, asym, core), (qt, st, asym, comp, bs, n, k)
    # per-channel (blocksize -1)
    assert L.ns_BTLAGemmPackBSize(64, 256, -1 & 0xFFFFFFFFFFFFFFFF, pkg.S4, pkg.F32, False, pkg.COMP_F32, None) == \
        nso.pack_size(64, 256, -1, nso.S4, nso.F32, False, nso.CORE_AVX512F)


def test_forced_core_and_unsupported(L, pkg, nso):
    try:
        for core in range(9):
            L.ns_set_pack_core(core)
            assert L.ns_BTLAGemmPackBSize(96, 128, 64, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None) == \
                nso.pack_size(96, 128, 64, nso.S4, nso.BF16, False, core)
    finally:
        L.ns_set_pack_core(pkg.CORE_AUTO)
    assert L.ns_BTLAGemmPackBSize(96, 128, 64, 0x1234, pkg.BF16, False, pkg.COMP_INT8, None) == 0  # unknown dtype
    # g_idx: the blob grows by the int[K] shuffle section (enable_shuffle, bestla_storage.h:761-765)
    idx = np.zeros(128, np.int32)
    L.ns_set_pack_core(nso.CORE_AVX512_VNNI_KB)
    try:
        with_idx = L.ns_BTLAGemmPackBSize(96, 128, 64, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, nso.ptr(idx))
    finally:
        L.ns_set_pack_core(pkg.CORE_AUTO)
    assert with_idx == nso.lib().nso_pack_size_gidx(96, 128, 64, nso.S4, nso.BF16, 0, nso.CORE_AVX512_VNNI_KB)
    assert with_idx > nso.pack_size(96, 128, 64, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)


def test_no_device_fails_loudly(L, pkg, nso):
    """Without a GPU the product must refuse (no CPU fallback); with one this test is a no-op."""
    if L.ns_hip_device_count() > 0:
        return
    w = np.zeros((16, 128), np.float32)
    blob = nso.quant_pack(w, 32, nso.S4)
    assert not L.ns_hip_weight_from_blob(nso.ptr(blob), None)
    assert "no HIP device" in pkg.last_error()
    out = np.full((1, 16), 7.0, np.float32)
    a = np.zeros((1, 128), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), 1, 16, 128, 128, 16, None)  # prints Err, as the reference
    assert np.all(out == 7.0)  # output untouched: nothing computed on the CPU
    assert not L.bestla_fusion_QKV_f32f32_support(nso.ptr(blob), nso.ptr(blob), nso.ptr(blob), 1, 16, 128)
    # library-managed kv cache: support() is what makes the graph builder choose it — false without a device; an update
    # that is called anyway leaves the cache untouched
    import ctypes as C
    assert not L.bestla_reordered_attn_fp32_support(C.byref(pkg.AttnShape(1, 4, 4, 64, 1, 8)))
    cache = np.full(4 * 8 * 64 * 2, 0x55, np.uint8)
    cur = np.ones((1, 2, 4, 64), np.float32)
    u = pkg.KvUpdateArgs(cur.ctypes.data, cache.ctypes.data, 1, 4, 64, 0, 2, 8, 2 * 4 * 64, 64, 4 * 64, 1, False)
    L.bestla_reordered_attn_fp32_update_k(C.byref(u))
    assert np.all(cache == 0x55) and "no HIP device" in pkg.last_error()


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/ns_bestla.h must compile as C99 on its own (no C++, no HIP, no torch
    types in any signature) — what a cgo / JNI / ctypes binding generator would consume"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    src = tmp_path / "h.c"
    src.write_text('#include "ns_bestla.h"\nint main(void) { return (int)sizeof(attn_shape_t) * 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_fallback_of_the_reference_harness_is_interposed_by_the_library(pkg):
    """oracle/ne_ref_stubs.c holds an aborting fallback for every `bestla_*` / `ns_BTLAGemm*` symbol the reference's graph
    executor and model code reference; libns_hip.so must export each of them (it is loaded RTLD_GLOBAL first and wins the
    symbol resolution), otherwise a drop-in run would abort in the fallback"""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = re.findall(r"NE_REF_STUB\((bestla_\w+|ns_BTLA\w+)\)", open(os.path.join(root, "oracle", "ne_ref_stubs.c")).read())
    assert len(names) >= 25
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH], text=True).splitlines() if ln.split()}
    missing = [n for n in names if n not in exported]
    assert not missing, missing


def test_tuning_keys_documented_in_the_header_are_accepted_without_a_device(L):
    """ns_hip_set_tuning: every key the header documents is known (process-wide switches: no device needed), unknown keys are
    refused; the values are put back to their defaults"""
    src = open(os.path.join(ROOT, "include", "ns_bestla.h")).read()
    block = src[src.index("Diagnostics / A-B switches of the kernels"):src.index("int ns_hip_set_tuning")]
    keys = re.findall(r'^\s*\*\s+"([a-z0-9_]+)"', block, flags=re.M) + re.findall(r'",\s*"([a-z0-9_]+)"', block)
    assert {"gemv2", "g3_bm", "g3_min_m", "i8_mfma", "i8_tile", "gv_nw", "attn_wg_target", "attn_min_keys"} <= set(keys), keys
    defaults = {"gemv2": 1, "i8_mfma": 2, "planes": 1, "planes_load": 0}
    for k in sorted(set(keys)):
        assert L.ns_hip_set_tuning(k.encode(), defaults.get(k, 0)) == 0, k
    assert L.ns_hip_set_tuning(b"no_such_key", 1) == -1
    L.ns_hip_reset_error()
