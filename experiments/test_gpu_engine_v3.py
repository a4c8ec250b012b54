"""Decode engine (csrc/ns_engine.hip: a chain of batch-1 GEMV operators as ONE persistent launch) against the
launch-per-operator path: every operator's fp32 output must equal ns_hip_*_forward_h's BIT FOR BIT when the launches add
a tile's partial sums in the engine's order (8 waves per tile, ns_hip_set_tuning("gv_nw", 8)) — and the launches
themselves are pinned to the oracle in tests/test_gpu_fullsize.py / test_gpu_parity.py, so the engine inherits the 1e-3
bar against the reference (checked directly for the last operator as well).  Also: relaunches (epoch-tagged hand-off
granules of the previous token must not be taken for fresh ones), a graph replay, the refusal of formats outside the
engine's envelope."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _weight(L, pkg, nso, n, k, seed, qt=None, st_dt=None, bs=32, asym=False, comp=None, scale=None):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qt = pkg.S4 if qt is None else qt
    st_dt = pkg.BF16 if st_dt is None else st_dt
    comp = pkg.COMP_INT8 if comp is None else comp
    g = torch.Generator(device="cuda").manual_seed(seed)
    dW = torch.randn((n, k), generator=g, device="cuda") * (scale if scale else 1.0 / np.sqrt(k))
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, asym, comp, None)
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, bs, qt, st_dt, asym, comp, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    wt = pkg.Weight.from_device_blob(dBlob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return wt, blob


@pytest.mark.parametrize("d,ff,vocab,layers", [(2048, 5632, 4000, 2), (4096, 11008, 32000, 1)])
def test_engine_chain_equals_launches_bit_for_bit(L, pkg, nso, d, ff, vocab, layers):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws, blobs = [], {}
    for il in range(layers):
        lw = {}
        for j, (name, n, k) in enumerate([("q", d, d), ("k", d, d), ("v", d, d), ("o", d, d), ("w1", ff, d), ("w3", ff, d), ("w2", d, ff)]):
            lw[name], b = _weight(L, pkg, nso, n, k, 100 + il * 8 + j, scale=(1.6 / np.sqrt(k) if name == "w2" else None))
            if il == layers - 1 and name == "w2":
                blobs["w2"] = b
        ws.append(lw)
    head, _ = _weight(L, pkg, nso, vocab, d, 99)
    g = torch.Generator(device="cuda").manual_seed(7)
    x0 = torch.randn((1, d), generator=g, device="cuda") * 1.7
    x0h = x0.half()
    f32 = lambda n: torch.full((1, n), 7.0, dtype=torch.float32, device="cuda")
    f16 = lambda n: torch.zeros((1, n), dtype=torch.float16, device="cuda")
    qkv, qkvh, attn, attnh, t2, t2h, x, xh, logits = f32(3 * d), f16(3 * d), f32(d), f16(d), f32(ff), f16(ff), f32(d), f16(d), f32(vocab)
    t2_in_h = f16(ff)

    def launches():
        inp, inph = x0, x0h
        snaps = []
        for il, lw in enumerate(ws):
            pkg.check(L.ns_hip_fusion_qkv_forward_h(inp.data_ptr(), inph.data_ptr(), lw["q"].h, lw["k"].h, lw["v"].h, qkv.data_ptr(), qkvh.data_ptr(), 1, d, d, st))
            pkg.check(L.ns_hip_f32f32_forward_h(qkv.data_ptr(), qkvh.data_ptr(), lw["o"].h, attn.data_ptr(), attnh.data_ptr(), 1, d, d, 0, None, 0, st))
            pkg.check(L.ns_hip_fusion_ffn3_gateup_h(attn.data_ptr(), attnh.data_ptr(), lw["w1"].h, lw["w3"].h, None, t2.data_ptr(), t2h.data_ptr(), 1, pkg.EPI_SILU, st))
            if il == layers - 1:
                t2_in_h.copy_(t2h)
            pkg.check(L.ns_hip_f32f32_forward_h(t2.data_ptr(), t2h.data_ptr(), lw["w2"].h, x.data_ptr(), xh.data_ptr(), 1, ff, d, 0, None, 0, st))
            torch.cuda.synchronize()
            snaps.append(dict(qkv=qkv.clone(), wo=attn.clone(), gateup=t2.clone(), down=x.clone()))
            inp, inph = x, xh
        pkg.check(L.ns_hip_f32f32_forward_h(inp.data_ptr(), inph.data_ptr(), head.h, logits.data_ptr(), None, 1, d, vocab, 0, None, 0, st))
        torch.cuda.synchronize()
        return snaps, logits.clone()

    assert L.ns_hip_set_tuning(b"gv_nw", 8) == 0
    try:
        ref, ref_logits = launches()
    finally:
        L.ns_hip_set_tuning(b"gv_nw", 0)

    # ---- the same operators as one engine chain; every operator writes its fp32 output ----
    outs, ops, prev = [], [], -1
    elog = f32(vocab)
    for il, lw in enumerate(ws):
        eo = dict(qkv=f32(3 * d), wo=f32(d), gateup=f32(ff), down=f32(d))
        outs.append(eo)
        iq = len(ops)
        ops.append(pkg.EngineOp(lw["q"].h, None, prev, eo["qkv"].data_ptr(), 0))
        ops.append(pkg.EngineOp(lw["k"].h, None, -2, eo["qkv"].data_ptr() + 4 * d, 0))
        ops.append(pkg.EngineOp(lw["v"].h, None, -2, eo["qkv"].data_ptr() + 8 * d, 0))
        io = len(ops)
        ops.append(pkg.EngineOp(lw["o"].h, None, iq, eo["wo"].data_ptr(), 0))
        ig = len(ops)
        ops.append(pkg.EngineOp(lw["w1"].h, lw["w3"].h, io, eo["gateup"].data_ptr(), pkg.EPI_SILU))
        prev = len(ops)
        ops.append(pkg.EngineOp(lw["w2"].h, None, ig, eo["down"].data_ptr(), 0))
    ops.append(pkg.EngineOp(head.h, None, prev, elog.data_ptr(), 0))
    arr = (pkg.EngineOp * len(ops))(*ops)
    eng = L.ns_hip_engine_create(arr, len(ops), x0h.data_ptr())
    assert eng, pkg.last_error()
    try:
        pkg.check(L.ns_hip_engine_launch(eng, st))
        assert L.ns_hip_engine_status(eng) == 0
        checks = [("layer %d %s" % (il, name), outs[il][name], ref[il][name]) for il in range(layers) for name in ("qkv", "wo", "gateup", "down")]
        checks.append(("logits", elog, ref_logits))
        for name, got, want in checks:
            assert not got.isnan().any() and not want.isnan().any(), name + ": NaN (test scaling)"
            nbad = int((got.view(torch.int32) != want.view(torch.int32)).sum())
            assert nbad == 0, "%s: %d of %d outputs differ from the launches (max |diff| %g, first at %d; NaN engine %d launches %d)" % (
                name, nbad, want.numel(), float((got - want).abs().max()), int((got != want).flatten().nonzero()[0]),
                int(got.isnan().sum()), int(want.isnan().sum()))
        # the last down projection against the ORACLE on the same blob and the same fp16 input row: north_star's bar
        a = t2_in_h.float().cpu().numpy()
        want = nso.gemm_f64(a, blobs["w2"])
        assert nso.rel_l2(outs[-1]["down"].cpu().numpy(), want) < 1e-3
        # relaunches: the granules of the previous token carry the previous epoch and must not satisfy a hand-in
        for _ in range(3):
            elog.fill_(7.0)
            pkg.check(L.ns_hip_engine_launch(eng, st))
            assert L.ns_hip_engine_status(eng) == 0
            assert torch.equal(elog.view(torch.int32), ref_logits.view(torch.int32))
        # a HIP-graph replay of the launch
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                pkg.check(L.ns_hip_engine_launch(eng, C.c_void_p(side.cuda_stream)))
            for _ in range(2):
                elog.fill_(7.0)
                gr.replay()
                side.synchronize()
                assert torch.equal(elog.view(torch.int32), ref_logits.view(torch.int32))
        assert L.ns_hip_engine_status(eng) == 0
    finally:
        L.ns_hip_engine_destroy(eng)


def test_engine_refuses_formats_outside_its_envelope(L, pkg, nso):
    import torch
    x = torch.zeros((1, 2048), dtype=torch.float16, device="cuda")
    for kw in (dict(qt=pkg.S8, comp=pkg.COMP_F32), dict(bs=128), dict(asym=True, st_dt=pkg.F32, comp=pkg.COMP_F32)):
        w, _ = _weight(L, pkg, nso, 256, 2048, 5, **kw)
        arr = (pkg.EngineOp * 1)(pkg.EngineOp(w.h, None, -1, None, 0))
        assert not L.ns_hip_engine_create(arr, 1, x.data_ptr())
        assert b"envelope" in L.ns_hip_last_error()
    w, _ = _weight(L, pkg, nso, 256, 2048, 6)
    arr = (pkg.EngineOp * 2)(pkg.EngineOp(w.h, None, -1, None, 0), pkg.EngineOp(w.h, None, 1, None, 0))
    assert not L.ns_hip_engine_create(arr, 2, x.data_ptr())  # an operator cannot read its own (or a later) output
