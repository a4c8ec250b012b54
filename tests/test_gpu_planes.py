"""Native bit-plane streaming (round 4): weights of the 1-3 / 5-7 bit formats carry a second device copy whose code
records have the format's own width (ns_weight::native, repack_planes_kernel) and the decode kernel rebuilds its nibble / byte
words from the plane words in registers.  Same fp16 operand values as from the widened records -> the SAME BITS out; and the
oracle's fp64 GEMM within north_star's 1e-3."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(autouse=True)
def _weights_get_their_native_copy(L):
    assert L.ns_hip_set_tuning(b"planes_load", 1) == 0
    yield
    L.ns_hip_set_tuning(b"planes_load", 0)


def _blob(L, pkg, nso, n, k, bits, st_dt, bs, comp, asym, seed):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    dW = torch.randn((n, k), generator=g, device="cuda") * 0.05
    qt = pkg.INT_TYPES[bits]
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, asym, comp, None)
    assert size > 0, pkg.last_error()
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, bs, qt, st_dt, asym, comp, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    wt = pkg.Weight.from_device_blob(dBlob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return blob, wt


def _fwd(L, pkg, wt, dA, m, k, n, shadow, epi=0, dD=None):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dA16 = dA.to(torch.float16) if shadow else None
    dC = torch.full((m, n), 7.0, dtype=torch.float32, device="cuda")
    pkg.check(L.ns_hip_f32f32_forward_h(dA.data_ptr(), dA16.data_ptr() if shadow else None, wt.h, dC.data_ptr(), None, m, k, n, epi,
                                        dD.data_ptr() if dD is not None else None, n if dD is not None else 0, st))
    torch.cuda.synchronize()
    return dC


CASES = [  # scale dtype, group, compute, asym, n, k
    ("BF16", 32, "COMP_INT8", False, 272, 1024),
    ("F32", 128, "COMP_F32", True, 144, 768),
    ("BF16", 64, "COMP_F32", True, 100, 1000),     # ragged N and K: padded columns and k-steps hold the code of zero
    ("F32", 4096, "COMP_F32", False, 48, 4096),    # per-channel scales, long K
]


@pytest.mark.parametrize("bits", [1, 2, 3, 5, 6, 7])
@pytest.mark.parametrize("case", CASES, ids=["g32_bf16", "g128_asym_f32", "g64_asym_ragged", "per_channel"])
def test_native_records_give_the_bits_of_the_widened_records(L, pkg, nso, bits, case):
    import torch
    st_dt, bs, comp, asym, n, k = case
    blob, wt = _blob(L, pkg, nso, n, k, bits, getattr(pkg, st_dt), bs, getattr(pkg, comp), asym, 100 + bits)
    try:
        for m, shadow in ((1, True), (1, False), (3, True), (4, False), (8, True)):
            if m > 1 and k % 128:
                continue  # (several rows need K to fill whole k-steps on this kernel: served elsewhere)
            g = torch.Generator(device="cuda").manual_seed(m)
            dA = torch.randn((m, k), generator=g, device="cuda")
            dD = torch.randn((m, n), generator=g, device="cuda")
            outs = {}
            for on in (0, 1):
                assert L.ns_hip_set_tuning(b"planes", on) == 0
                outs[on] = (_fwd(L, pkg, wt, dA, m, k, n, shadow), _fwd(L, pkg, wt, dA, m, k, n, shadow, pkg.EPI_ADD, dD))
            assert torch.equal(outs[0][0], outs[1][0]), (bits, m, shadow)
            assert torch.equal(outs[0][1], outs[1][1]), (bits, m, shadow, "add")
            a = dA.cpu().numpy()
            ref = nso.gemm_f64(a, blob, a16=shadow)
            assert nso.rel_l2(outs[1][0].cpu().numpy(), ref) < TOL
    finally:
        L.ns_hip_set_tuning(b"planes", 1)
        wt.free()


@pytest.mark.parametrize("bits", [3, 5])
def test_native_records_in_the_fused_launches(L, pkg, nso, bits):
    """fused Q/K/V (three matrices side by side) and gate/up . SiLU (two matrices in lockstep) on native records = on widened ones"""
    import torch
    k, m = 1024, 2
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = [_blob(L, pkg, nso, n, k, bits, pkg.BF16, 32, pkg.COMP_INT8, False, 7 + i)[1] for i, n in enumerate((256, 64, 64, 512, 512))]
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        dA = torch.randn((m, k), generator=g, device="cuda")
        dA16 = dA.half()
        res = {}
        for on in (0, 1):
            assert L.ns_hip_set_tuning(b"planes", on) == 0
            qkv = torch.zeros((3, m, 256), device="cuda")
            pkg.check(L.ns_hip_fusion_qkv_forward_h(dA.data_ptr(), dA16.data_ptr(), ws[0].h, ws[1].h, ws[2].h, qkv.data_ptr(), None, m, k, 256, st))
            t2 = torch.zeros((m, 512), device="cuda")
            pkg.check(L.ns_hip_fusion_ffn3_gateup_h(dA.data_ptr(), dA16.data_ptr(), ws[3].h, ws[4].h, None, t2.data_ptr(), None, m, pkg.EPI_SILU, st))
            torch.cuda.synchronize()
            res[on] = (qkv, t2)
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert float(res[1][1].abs().sum()) > 0
    finally:
        L.ns_hip_set_tuning(b"planes", 1)
        for w in ws:
            w.free()
