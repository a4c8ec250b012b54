#!/usr/bin/env python3
"""Mints tests/golden/btla_golden.npz — known-answer vectors for the BesTLA weight-only-quant path.

The reference holds no golden files for this path, so the vectors are produced HERE from the real
reference scalar kernels: codes/scales/zero-points come from bestla/bestla/kernel_ref.h (compiled from /root/reference
into oracle/_ref/libkernel_ref.so by oracle/Makefile), the blob container + GEMM outputs from the oracle restatement,
which tests/test_oracle_vs_ref.py pins bit-exactly to those kernels.  Run in the container that has /root/reference:
    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import nso  # noqa: E402

CASES = [
    # name, qtype, stype, asym, core, blocksize, n, k
    ("q4_0_g32_bf16_vnni", "S4", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 96, 256),   # BesTLA "Q4_0" (core/README.md:97)
    ("s4_asym_g32_f32_avx512f", "S4", "F32", True, "CORE_AVX512F", 32, 100, 160),
    ("s4_sym_g128_bf16_amxint8", "S4", "BF16", False, "CORE_AMX_INT8_KB", 128, 48, 256),
    ("s4_perchannel_f32_avx2", "S4", "F32", False, "CORE_AVX2", -1, 50, 96),
    ("s8_sym_g32_bf16_amxbf16", "S8", "BF16", False, "CORE_AMX_BF16", 32, 96, 128),
    ("s8_asym_g64_f16_avx512f", "S8", "F16", True, "CORE_AVX512F", 64, 64, 192),
    ("nf4_g128_bf16_avx512f", "F4_NF4", "BF16", False, "CORE_AVX512F", 128, 96, 256),
    ("nf4_g32_f32_amxbf16", "F4_NF4", "F32", False, "CORE_AMX_BF16", 32, 48, 128),
    ("fp4_bnb_g32_f32", "F4_BNB", "F32", False, "CORE_AVX512F", 32, 48, 128),
    ("fp4_e2m1_g64_bf16", "F4_E2M1", "BF16", False, "CORE_AVX512F", 64, 48, 128),
    ("s3_sym_g32_bf16_vnni", "S3", "BF16", False, "CORE_AVX512_VNNI_KB", 32, 48, 128),
    ("s5_asym_g32_f32_vnni", "S5", "F32", True, "CORE_AVX512_VNNI_KB", 32, 48, 128),
    ("s2_sym_g32_f16_avx512f", "S2", "F16", False, "CORE_AVX512F", 32, 48, 128),
    ("s7_asym_g32_bf16_vnni", "S7", "BF16", True, "CORE_AVX512_VNNI_KB", 32, 48, 128),
    ("fp8_e4m3_g32_e8m0_avx512f", "F8_E4M3", "F8_E8M0", False, "CORE_AVX512F", 32, 96, 128),   # quant_utils.cpp:336-341
    ("fp8_e5m2_g32_e8m0_amxbf16", "F8_E5M2", "F8_E8M0", False, "CORE_AMX_BF16", 32, 48, 128),
    ("fp8_e4m3_g128_f32_avx512f", "F8_E4M3", "F32", False, "CORE_AVX512F", 128, 48, 256),
    ("fp8_e5m2_g64_f32_avx512f", "F8_E5M2", "F32", False, "CORE_AVX512F", 64, 50, 192),
]


def main():
    ref = nso.ref()
    assert ref is not None, "needs oracle/_ref (the reference tree)"
    out = {}
    names = []
    for idx, (name, qt, st, asym, core, bs, n, k) in enumerate(CASES):
        rng = np.random.default_rng(20240000 + idx)
        w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
        w[0, :32] = 0.0                       # all-zero group
        w[1, :32] = np.abs(w[1, :32]) + 0.01  # dominant-positive group -> negative scale
        w[2, 7] = 0.9                         # outlier
        a = rng.standard_normal((3, k)).astype(np.float32)
        qtype, stype = getattr(nso, qt), getattr(nso, st)
        bs_eff = k if bs <= 0 else bs
        # the REAL reference quantizer on the [K][N] matrix
        wkn = np.ascontiguousarray(w.T)
        q = np.zeros((k, n), np.int8)
        nb = (k + bs_eff - 1) // bs_eff
        sc = np.zeros((nb, n), np.float32)
        zp = np.zeros((nb, n), np.int8) if asym else None
        if nso.is_int_type(qtype):
            ref.ref_quantize_int(nso.ptr(wkn), nso.ptr(q), k, n, n, n, nso.ptr(sc), nso.ptr(zp), bs_eff, C.c_uint32(qtype))
        elif nso.is_f8_type(qtype):
            ref.ref_quantize_f8(nso.ptr(wkn), nso.ptr(q), k, n, n, n, nso.ptr(sc), bs_eff, C.c_uint32(qtype),
                                C.c_uint32(stype))
        else:
            ref.ref_quantize_f4(nso.ptr(wkn), nso.ptr(q), k, n, n, n, nso.ptr(sc), bs_eff, C.c_uint32(qtype))
        blob = nso.quant_pack(w, bs, qtype, stype, asym, getattr(nso, core))
        q2, sc2, zp2 = nso.unpack_canonical(blob)
        assert np.array_equal(q, q2), name  # blob carries exactly the reference kernel's codes
        c = nso.gemm_f64(a, blob)
        out[name + "/w"] = w
        out[name + "/a"] = a
        out[name + "/blob"] = np.array(blob)
        out[name + "/codes"] = q
        out[name + "/scales_f32"] = sc
        if asym:
            out[name + "/zps"] = zp
        out[name + "/dequant"] = nso.unpack_fp32(blob)
        out[name + "/c_f64"] = c
        out[name + "/meta"] = np.array([qtype, stype, int(asym), getattr(nso, core), bs, n, k], np.int64)
        names.append(name)
    out["names"] = np.array(names)
    path = os.path.join(HERE, "btla_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
