"""The oracle's blobs against the reference's REAL packer — prologue_b::gemm::WeightKBlockNInteger / WeightKBlockNFloat of
bestla/bestla/bestla_prologue_b.h (createStorage -> assign -> packTransposeWeight: quantize, padding-interleave, bit-plane
compress, reduce; the body of BTLAGemmQuantPackB, bestla_gemm.cpp:302-319), compiled from the reference tree into
oracle/_ref/libpack_ref.so (oracle/Makefile packref; oracle/pack_shim.cpp, oracle/standins/) and run in a subprocess with
NS_PACKREF_ISA=nosimd (the reference's runtime dispatch then takes its scalar kernels on any host) or with the host's own
ISA.  Since the GPU quantizer's blobs equal the oracle's byte for byte (tests/test_gpu_parity.py), this makes "bit-exact
with the reference" a statement about the reference's own packer, end to end: header, section offsets, codes, bit planes,
scales, zero points, reductions.

Compared: every byte a packer WRITES (the alignment gaps between sections and the tail of the reduce section are written by
neither side; found as the bytes two packs over differently pre-filled buffers agree on)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libpack_ref.so")

INT_CASES = [(q, s, a, bs) for q in ("S1", "S2", "S3", "S4", "S5", "S6", "S7", "S8") for s, a, bs in
             (("BF16", False, 32), ("F32", True, 128), ("F16", False, -1))]
FLT_CASES = [(q, s, False, bs) for q in ("F4_NF4", "F4_BNB", "F4_E2M1") for s, bs in (("BF16", 32), ("F32", 128))] + \
            [(q, s, False, 32) for q in ("F8_E4M3", "F8_E5M2") for s in ("F8_E8M0", "F32")]
# DQ8_BNB double-quantised scales (round 4): 32: the scale count is a multiple of the dq block; 128: it is not — the trailing
# block's maximum lands on the offset slot, as the reference's indexing has it (kernel_ref.h:1976)
# S4 and NF4 weights only: those are the two the reference can read back (bestla_prologue_b.h:742-751, :1298-1306; its packer writes
# DQ8 scales for other types too, byte-equal to the oracle's, but its own unpack of them is not one)
DQ_CASES = [("S4", "DQ8_BNB", False, 32), ("S4", "DQ8_BNB", False, 128), ("S4", "DQ8_BNB", False, 64),
            ("F4_NF4", "DQ8_BNB", False, 32), ("F4_NF4", "DQ8_BNB", False, 128)]


def worker(isa):
    """runs in a subprocess: prints one JSON list of mismatches"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nso
    P = C.CDLL(SO)
    P.packref_size.restype = C.c_size_t
    P.packref_size.argtypes = [C.c_int] * 3 + [C.c_uint32] * 2 + [C.c_int] * 2
    P.packref_quant_pack.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_uint32] * 2 + [C.c_int] * 3
    P.packref_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    P.packref_core_id.restype = C.c_uint64
    nso.lib().nso_core_id.restype = C.c_uint64
    rng = np.random.default_rng(5)
    bad, n_cases = [], 0
    for core in range(9):
        assert P.packref_core_id(core) == nso.lib().nso_core_id(core)
        ktile = [1, 1, 32, 32, 4, 4, 4, 4, 64][core]
        for qn, sn, asym, bs in INT_CASES + FLT_CASES + DQ_CASES:
            qt, st = getattr(nso, qn), getattr(nso, sn)
            is_flt = qn.startswith("F")
            if isa != "nosimd" and qn.startswith("F4"):
                continue   # the AVX512 F4 quantizer is a different encoder (tests/test_oracle_vs_avx.py)
            if is_flt and core >= 4:
                continue   # float weights never meet an integer-compute core (bestla_gemm.cpp:262-300 dispatches them to
                           # the fp32 / bf16 / fp16 cores only)
            for n, k in ((100, 256), (48, 192)):
                b = k if bs <= 0 else bs
                if b % ktile or (qn == "S8" and asym and core >= 4):
                    continue
                size = P.packref_size(n, k, bs, qt, st, int(asym), core)
                osz = nso.pack_size(n, k, bs, qt, st, asym, core)
                if size != osz:
                    bad.append((core, qn, sn, asym, bs, n, k, "size %d != %d" % (size, osz)))
                    continue
                if size == 0:
                    continue
                n_cases += 1
                w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
                b0, b1 = nso.aligned_bytes(size, fill=0), nso.aligned_bytes(size, fill=0xFF)
                for buf in (b0, b1):
                    assert P.packref_quant_pack(nso.ptr(buf), nso.ptr(w), n, k, k, bs, qt, st, int(asym), core, 1) == 0
                mine = nso.quant_pack(w, bs, qt, st, asym, core, fill=0)
                written = b0 == b1
                # two places where the reference's SCALAR dispatch is not usable as a definition (its vector dispatch is,
                # and agrees with the oracle — second test):
                #  * fp16 scales: the scalar unpackWeight cannot read them (1e32-sized garbage), and reduceWeight sums that
                #    unpack (bestla_prologue_b.h:455-470) -> the reduce section it writes is garbage; codes / scales / zero
                #    points are compared, the reduce section and the unpack are not
                #  * fp8 weights with fp32 scales: the scalar decompress_kblock_f8_fp drops the k-block offset of the scale row
                #    (kernel_ref.h:1017-1018) -> its whole-matrix unpack is wrong from the second k-block on
                scalar_f16 = isa == "nosimd" and sn == "F16" and not is_flt
                scalar_f8_f32 = isa == "nosimd" and qn.startswith("F8") and sn == "F32"
                if scalar_f16:
                    bi = nso.parse(mine)
                    written[bi.red_off:bi.red_off + bi.red_bytes] = False
                diff = written & (b0 != mine)
                if diff.any():
                    bad.append((core, qn, sn, asym, bs, n, k, "%d of %d written bytes differ, first at %d" % (
                        int(diff.sum()), int(written.sum()), int(np.argmax(diff)))))
                    continue
                # and back: the reference's unpackWeight on the ORACLE's blob == the oracle's unpack
                if not (scalar_f16 or scalar_f8_f32):
                    out = np.zeros((k, n), np.float32)
                    assert P.packref_unpack(nso.ptr(mine), nso.ptr(out), n, core, int(is_flt)) == 0
                    if not np.array_equal(out.view(np.uint32), nso.unpack_fp32(mine).view(np.uint32)):
                        bad.append((core, qn, sn, asym, bs, n, k, "unpack differs"))
    # BTLAGemmPackB: pre-quantized codes, with and without GPTQ act-order group indices (ShuffleIndices section)
    P.packref_pack_q.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_uint32] * 2 + [C.c_int] * 2 + [C.c_void_p]
    P.packref_size_gidx.restype = C.c_size_t
    P.packref_size_gidx.argtypes = [C.c_int] * 3 + [C.c_uint32] * 2 + [C.c_int] * 2
    for core in (1, 2, 4, 7):
        for qn, asym, bs in (("S4", False, 32), ("S4", True, 64), ("S8", False, 32), ("S3", False, 32), ("S2", True, 64)):
            for use_gidx in (False, True):
                n, k = 72, 256
                qt = getattr(nso, qn)
                if bs % [1, 1, 32, 32, 4, 4, 4, 4, 64][core]:
                    continue
                w = (rng.standard_normal((k, n)) * 0.02).astype(np.float32)
                q, sc, zp = nso.quantize(w, bs, qt, asym)
                g_idx = rng.permutation(np.repeat(np.arange(k // bs), bs)).astype(np.int32) if use_gidx else None
                mine = nso.pack_q(q, sc, zp, bs, qt, nso.BF16, core, fill=0, g_idx=g_idx)
                size = (P.packref_size_gidx if use_gidx else P.packref_size)(n, k, bs, qt, nso.BF16, int(asym), core)
                if size != mine.size:
                    bad.append((core, qn, "BF16", asym, bs, n, k, "pack_q size %d != %d" % (size, mine.size)))
                    continue
                n_cases += 1
                b0, b1 = nso.aligned_bytes(size, fill=0), nso.aligned_bytes(size, fill=0xFF)
                for buf in (b0, b1):
                    assert P.packref_pack_q(nso.ptr(buf), nso.ptr(q), n, nso.ptr(sc), nso.ptr(zp), n, k, bs, qt, nso.BF16, int(asym), core,
                                            nso.ptr(g_idx)) == 0
                diff = (b0 == b1) & (b0 != mine)
                if diff.any():
                    bad.append((core, qn, "BF16", asym, bs, n, k, "pack_q%s: %d written bytes differ, first at %d" % (
                        " + g_idx" if use_gidx else "", int(diff.sum()), int(np.argmax(diff)))))
    print("PACKREF_RESULT " + json.dumps({"cases": n_cases, "bad": bad}))


def run(isa):
    if not os.path.exists(SO):
        if not os.path.exists("/root/reference/bestla/bestla/bestla_prologue_b.h"):
            pytest.skip("oracle/_ref/libpack_ref.so not built (reference tree absent)")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "packref"], stdout=subprocess.DEVNULL)
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if not all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512_vnni")):
        pytest.skip("libpack_ref.so is compiled with AVX512 code generation enabled")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    if isa == "nosimd":
        env["NS_PACKREF_ISA"] = "nosimd"
    else:
        env.pop("NS_PACKREF_ISA", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), isa], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PACKREF_RESULT ")][-1]
    return json.loads(line[len("PACKREF_RESULT "):])


def test_oracle_blobs_equal_the_reference_packer_scalar_dispatch():
    """all nine cores x S1..S8 (sym / asym; bf16 / f32 / f16 scales; groups 32 / 128 / per-channel) x NF4 / FP4-BNB / FP4-E2M1 x
    FP8-E4M3 / E5M2 (E8M0 / f32 scales) x two shapes (ragged N: 100 columns in 24- / 48-wide tiles), reference kernels = scalar"""
    res = run("nosimd")
    assert res["cases"] > 380, res["cases"]
    assert not res["bad"], res["bad"][:5]


def test_oracle_blobs_equal_the_reference_packer_on_this_hosts_isa():
    """the same with the reference dispatching to its AVX512 / AVX2 kernels (interleave, compress, reduce, fp8 quantize, unpack):
    integer and fp8 blobs are still the oracle's byte for byte; the F4 family is excluded here because its AVX512 quantizer is a
    different encoder (signed scales), characterised in tests/test_oracle_vs_avx.py"""
    res = run("host")
    assert res["cases"] > 300, res["cases"]
    assert not res["bad"], res["bad"][:5]


def golden_worker():
    """runs in a subprocess (NS_PACKREF_ISA=nosimd): the real packer on the golden weights, against the golden blobs"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nso
    G = np.load(os.path.join(ROOT, "tests", "golden", "btla_golden.npz"))
    P = C.CDLL(SO)
    P.packref_size.restype = C.c_size_t
    P.packref_size.argtypes = [C.c_int] * 3 + [C.c_uint32] * 2 + [C.c_int] * 2
    P.packref_quant_pack.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_uint32] * 2 + [C.c_int] * 3
    bad, done = [], 0
    for name in [str(x) for x in G["names"]]:
        qt, st, asym, core, bs, n, k = [int(x) for x in G[name + "/meta"]]
        gold = G[name + "/blob"]
        w = np.ascontiguousarray(G[name + "/w"], np.float32)
        size = P.packref_size(n, k, bs, qt, st, asym, core)
        if size != gold.size:
            bad.append((name, "size %d != %d" % (size, gold.size)))
            continue
        b0, b1 = nso.aligned_bytes(size, fill=0), nso.aligned_bytes(size, fill=0xFF)
        for buf in (b0, b1):
            assert P.packref_quant_pack(nso.ptr(buf), nso.ptr(w), n, k, k, bs, qt, st, asym, core, 1) == 0
        written = b0 == b1
        if st == nso.F16 and nso.is_int_type(qt):   # the scalar dispatch cannot unpack fp16 scales: its reduce section is garbage
            bi = nso.parse(nso_aligned(nso, gold))
            written[bi.red_off:bi.red_off + bi.red_bytes] = False
        diff = written & (b0 != gold)
        done += 1
        if diff.any():
            bad.append((name, "%d of %d written bytes differ" % (int(diff.sum()), int(written.sum()))))
    print("PACKREF_RESULT " + json.dumps({"cases": done, "bad": bad}))


def nso_aligned(nso, arr):
    b = nso.aligned_bytes(arr.size)
    b[:] = arr
    return b


def test_committed_golden_blobs_are_what_the_reference_packer_writes():
    """tests/golden/btla_golden.npz (18 formats; checked against the GPU quantizer on the GPU box, where no reference tree
    exists: tests/test_golden.py) == the real packer's output on the same weights, here: the chain GPU == golden == reference
    packer closes through committed fixtures"""
    if not os.path.exists(SO) and not os.path.exists("/root/reference/bestla/bestla/bestla_prologue_b.h"):
        pytest.skip("oracle/_ref/libpack_ref.so not built (reference tree absent)")
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if not all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512dq", "avx512_vnni")):
        pytest.skip("libpack_ref.so is compiled with AVX512 code generation enabled")
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "packref"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, OMP_NUM_THREADS="4", NS_PACKREF_ISA="nosimd")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "golden"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("PACKREF_RESULT ")][-1][len("PACKREF_RESULT "):])
    assert res["cases"] == 18 and not res["bad"], res


if __name__ == "__main__":
    golden_worker() if sys.argv[1] == "golden" else worker(sys.argv[1])
