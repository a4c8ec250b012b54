"""gemvs_kernel (ns_gemvs.hip) — the small-batch / narrow-output streaming kernel: activations staged once per workgroup
and shared by the tiles it streams, split-K across workgroups, a service wave doing reduction + epilogue — against the
oracle's fp64 GEMM on the same blob (the bar of bestla's own GEMV tests: ut/bestla_prologue_b.cpp:548-785, 1e-3).

Every decomposition the planner can pick is forced here through ns_hip_set_tuning (slices 1..16, 1..15 streaming waves,
few workgroups = many tiles per workgroup, which exercises the LDS slot hand-back between the streaming waves and the
service wave), on plain / fused gate-up / fused QKV launches, every epilogue, ragged N, all five weight formats."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star


def _blob(L, pkg, nso, n, k, qt, st_dt, bs, comp, asym, seed, wscale=0.02):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(seed)
    dW = torch.randn((n, k), generator=g, device="cuda") * wscale
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, asym, comp, None)
    assert size > 0, pkg.last_error()
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, bs, qt, st_dt, asym, comp, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    wt = pkg.Weight.from_device_blob(dBlob.data_ptr(), size, st)
    torch.cuda.synchronize()
    return blob, wt, dBlob


@pytest.fixture(autouse=True)
def _always_this_kernel(L):
    """the library picks gemv_kernel or gemvs_kernel by shape (ns_gemvs.hip: launch_gemvs); these tests pin gemvs_kernel"""
    assert L.ns_hip_set_tuning(b"gvs", 3) == 0
    yield
    L.ns_hip_set_tuning(b"gvs", 1)


class _Tuning:
    def __init__(self, L, **kv):
        self.L, self.kv = L, kv

    def __enter__(self):
        for k, v in self.kv.items():
            assert self.L.ns_hip_set_tuning(k.encode(), v) == 0
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            self.L.ns_hip_set_tuning(k.encode(), {"gvs": 3, "gvs_table": -1, "gvs_finalize": 1}.get(k, 0))


def _fwd(L, pkg, wt, dA, m, k, n, epi=0, dD=None, shadow=True, want16=False):
    import torch
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dA16 = dA.to(torch.float16) if shadow else None
    dC = torch.full((m, n), 7.0, dtype=torch.float32, device="cuda")
    dC16 = torch.zeros((m, n), dtype=torch.float16, device="cuda") if want16 else None
    pkg.check(L.ns_hip_f32f32_forward_h(dA.data_ptr(), dA16.data_ptr() if shadow else None, wt.h, dC.data_ptr(),
                                        dC16.data_ptr() if want16 else None, m, k, n, epi, dD.data_ptr() if dD is not None else None,
                                        n if dD is not None else 0, st))
    torch.cuda.synchronize()
    return (dC.cpu().numpy(), dC16.cpu().numpy()) if want16 else dC.cpu().numpy()


FORMATS = [
    ("int4_g32_bf16", "S4", "BF16", 32, "COMP_INT8", False),
    ("int8_g32_bf16", "S8", "BF16", 32, "COMP_F32", False),
    ("nf4_g128_bf16", "F4_NF4", "BF16", 128, "COMP_BF16", False),
    ("int4_asym_g128_f32", "S4", "F32", 128, "COMP_F32", True),
    ("fp8_e4m3_g32_f32", "F8_E4M3", "F32", 32, "COMP_F32", False),
]


@pytest.mark.parametrize("fmt", FORMATS, ids=[f[0] for f in FORMATS])
@pytest.mark.parametrize("m", [2, 5, 8, 16])
def test_every_format_and_row_count_against_the_oracle(L, pkg, nso, fmt, m):
    """the planner's own decomposition on a ragged output (1000 columns: the last tile is partial) and K = 2048"""
    import torch
    name, qt, st_dt, bs, comp, asym = fmt
    if not hasattr(pkg, qt):
        pytest.skip("format constant %s not exported by the package" % qt)
    n, k = 1000, 2048
    blob, wt, _keep = _blob(L, pkg, nso, n, k, getattr(pkg, qt), getattr(pkg, st_dt), bs, getattr(pkg, comp), asym, seed=m + 31)
    g = torch.Generator(device="cuda").manual_seed(m)
    dA = torch.randn((m, k), generator=g, device="cuda")
    out = _fwd(L, pkg, wt, dA, m, k, n)
    with _Tuning(L, gvs=0):
        old = _fwd(L, pkg, wt, dA, m, k, n)
    ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
    assert nso.rel_l2(out, ref) < TOL, (name, m, nso.rel_l2(out, ref))
    assert nso.rel_l2(out, ref16) < (6e-4 if qt.startswith("F4") else 3e-5), (name, m, nso.rel_l2(out, ref16))
    # the one-tile-per-workgroup kernel computes the same products in another fp32 order
    assert nso.rel_l2(out, old) < 2e-6, (name, m, nso.rel_l2(out, old))
    wt.free()


@pytest.mark.parametrize("slices,waves,grid", [(1, 1, 0), (1, 3, 16), (1, 15, 8), (2, 8, 0), (4, 7, 32), (8, 4, 0), (16, 2, 64), (2, 15, 2)])
def test_every_decomposition_gives_the_oracles_result(L, pkg, nso, slices, waves, grid):
    """forced slices x streaming waves x workgroup count: few workgroups = many tiles per workgroup (the slot hand-back runs
    dozens of times per workgroup), many slices = a few k-steps per slice (some streaming waves own none)"""
    import torch
    n, k, m = 1552, 4096, 8   # 97 tiles: never a multiple of the group count
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=slices * 100 + waves)
    g = torch.Generator(device="cuda").manual_seed(waves)
    dA = torch.randn((m, k), generator=g, device="cuda")
    ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
    with _Tuning(L, gvs_slices=slices, gvs_waves=waves, gvs_grid=grid):
        out = _fwd(L, pkg, wt, dA, m, k, n)
        again = _fwd(L, pkg, wt, dA, m, k, n)
    assert nso.rel_l2(out, ref) < TOL and nso.rel_l2(out, ref16) < 3e-5, (slices, waves, grid, nso.rel_l2(out, ref16))
    assert np.max(np.abs(out - ref)) < 8e-3 * np.sqrt(np.mean(ref ** 2))
    assert np.array_equal(out.view(np.int32), again.view(np.int32))  # fixed summation order: run-to-run identical
    wt.free()


@pytest.mark.parametrize("slices,grid,m", [(2, 0, 8), (4, 16, 3), (8, 0, 16), (4, 4, 8)])
def test_split_k_finished_in_the_launch_equals_the_finalize_launch_bit_for_bit(L, pkg, nso, slices, grid, m):
    """split-K two ways: the tile's owner (slice = tile ordinal mod S, its own tiles streamed last) adds the slices inside the
    launch behind self-resetting tickets, or gemvs_finalize_kernel adds them in a second launch — the same slice order, so the
    same bits; run three times in a row (the tickets must be back at zero), fused gate/up included"""
    import torch
    n, k = 1552, 4096
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    blob, wt, _k0 = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=slices + m)
    blob3, wt3, _k3 = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=slices + m + 50)
    g = torch.Generator(device="cuda").manual_seed(m)
    dA = torch.randn((m, k), generator=g, device="cuda")
    dA16 = dA.half()

    def both():
        o = _fwd(L, pkg, wt, dA, m, k, n)
        t2 = torch.full((m, n), 7.0, device="cuda")
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(dA.data_ptr(), dA16.data_ptr(), wt.h, wt3.h, None, t2.data_ptr(), None, m, pkg.EPI_SILU, st))
        torch.cuda.synchronize()
        return o, t2.cpu().numpy()
    with _Tuning(L, gvs_slices=slices, gvs_grid=grid):
        want, want2 = both()
    L.ns_hip_set_tuning(b"gvs_finalize", 0)
    try:
        for _ in range(3):
            with _Tuning(L, gvs_slices=slices, gvs_grid=grid):
                got, got2 = both()
            assert np.array_equal(got.view(np.int32), want.view(np.int32)) and np.array_equal(got2.view(np.int32), want2.view(np.int32))
    finally:
        L.ns_hip_set_tuning(b"gvs_finalize", 1)
    with _Tuning(L, gvs_slices=slices, gvs_grid=grid):
        for _ in range(1):
            got, got2 = both()
            assert np.array_equal(got.view(np.int32), want.view(np.int32)) and np.array_equal(got2.view(np.int32), want2.view(np.int32))
    assert nso.rel_l2(got, nso.gemm_f64(dA.cpu().numpy(), blob)) < TOL
    wt.free(), wt3.free()


@pytest.mark.parametrize("epi,name", [(1, "add"), (2, "mul"), (3, "add_gelu"), (4, "gelu"), (5, "silu")])
@pytest.mark.parametrize("slices", [1, 4])
def test_epilogues_with_and_without_split_k(L, pkg, nso, epi, name, slices):
    """custom::epilogue::{Add, Mul, Add_Gelu, Gelu, Silu} applied by the service wave (no split) and by the finalize launch
    (split-K), with the fp16 shadow of the output = RNE(C)"""
    import torch
    n, k, m = 520, 1024, 6
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=epi)
    g = torch.Generator(device="cuda").manual_seed(epi + 40)
    dA = torch.randn((m, k), generator=g, device="cuda")
    dD = torch.randn((m, n), generator=g, device="cuda")
    with _Tuning(L, gvs_slices=slices):
        out, out16 = _fwd(L, pkg, wt, dA, m, k, n, epi=epi, dD=dD, want16=True)
    x = nso.gemm_f64(dA.cpu().numpy(), blob).astype(np.float64)
    d = dD.cpu().numpy().astype(np.float64)
    gelu = lambda v: 0.5 * v * (1.0 + np.tanh(0.7978845834732056 * (v + 0.044714998453855515 * v ** 3)))
    want = {1: x + d, 2: x * d, 3: gelu(x + d), 4: gelu(x), 5: x / (1.0 + np.exp(-x))}[epi]
    assert nso.rel_l2(out, want) < TOL, (name, slices, nso.rel_l2(out, want))
    assert np.array_equal(out16, out.astype(np.float16))
    wt.free()


@pytest.mark.parametrize("table", [0, 1])
@pytest.mark.parametrize("n,k,m,slices", [(528, 1024, 8, 1), (2064, 4096, 5, 1), (272, 2048, 16, 4)])
def test_f4_pair_table_and_valu_decode_agree_with_the_oracle(L, pkg, nso, n, k, m, slices, table):
    """NF4 / FP4 decode: through the LDS pair table (activations re-ordered to the code bytes' pairing) or by v_perm lookups —
    whichever the launch size would pick, both are forced here on small and large launches, for all three f4 tables"""
    import torch
    for qt in (pkg.F4_NF4, pkg.F4_E2M1, pkg.F4_BNB):
        blob, wt, _keep = _blob(L, pkg, nso, n, k, qt, pkg.BF16, 128, pkg.COMP_BF16, False, seed=n + m)
        g = torch.Generator(device="cuda").manual_seed(m + table)
        dA = torch.randn((m, k), generator=g, device="cuda")
        with _Tuning(L, gvs_table=table, gvs_slices=slices):
            out = _fwd(L, pkg, wt, dA, m, k, n)
        ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
        assert nso.rel_l2(out, ref) < TOL and nso.rel_l2(out, ref16) < 6e-4, (qt, table, nso.rel_l2(out, ref16))
        wt.free()


@pytest.mark.parametrize("slices,grid", [(1, 0), (1, 24), (2, 0), (8, 0)])
@pytest.mark.parametrize("m", [3, 8, 16])
def test_fused_gate_up(L, pkg, nso, m, slices, grid):
    """bestla_fusion_FFN_SiLu's first half as ONE launch: tmp1 = silu(A W1), out = (A W3) * tmp1 (ip_fusion_ffn.cpp:364-406)"""
    import torch
    n, k = 1376, 1024
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    b1, w1, _k1 = _blob(L, pkg, nso, n, k, pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16, False, seed=m)
    b3, w3, _k3 = _blob(L, pkg, nso, n, k, pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16, False, seed=m + 1)
    g = torch.Generator(device="cuda").manual_seed(m + 9)
    dA = torch.randn((m, k), generator=g, device="cuda")
    dA16 = dA.half()
    t1 = torch.full((m, n), 7.0, device="cuda")
    t2 = torch.full((m, n), 7.0, device="cuda")
    t216 = torch.zeros((m, n), device="cuda", dtype=torch.float16)
    with _Tuning(L, gvs_slices=slices, gvs_grid=grid):
        pkg.check(L.ns_hip_fusion_ffn3_gateup_h(dA.data_ptr(), dA16.data_ptr(), w1.h, w3.h, t1.data_ptr(), t2.data_ptr(), t216.data_ptr(),
                                                m, pkg.EPI_SILU, st))
        torch.cuda.synchronize()
    a = dA.cpu().numpy()
    x1, x3 = nso.gemm_f64(a, b1).astype(np.float64), nso.gemm_f64(a, b3).astype(np.float64)
    s = x1 / (1.0 + np.exp(-x1))
    assert nso.rel_l2(t1.cpu().numpy(), s) < TOL
    assert nso.rel_l2(t2.cpu().numpy(), s * x3) < TOL, (m, slices, nso.rel_l2(t2.cpu().numpy(), s * x3))
    assert np.array_equal(t216.cpu().numpy(), t2.cpu().numpy().astype(np.float16))
    w1.free(), w3.free()


@pytest.mark.parametrize("slices,grid", [(1, 0), (1, 12), (4, 0)])
@pytest.mark.parametrize("ns", [(512, 512, 512), (1024, 256, 256), (1000, 200, 136)], ids=["mha", "gqa", "ragged"])
def test_fused_qkv(L, pkg, nso, ns, slices, grid):
    """bestla_fusion_QKV_f32f32_forward as one launch: three matrices side by side in the tile list (GQA widths, matrices that
    end inside a tile); nothing is written beyond a matrix's own columns"""
    import torch
    k, m = 2048, 8
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    blobs, wts, keep = [], [], []
    for i, n in enumerate(ns):
        b, w, kp = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n + i)
        blobs.append(b), wts.append(w), keep.append(kp)
    g = torch.Generator(device="cuda").manual_seed(5)
    dA = torch.randn((m, k), generator=g, device="cuda")
    dA16 = dA.half()
    ldc = max(ns)
    out = torch.full((3, m, ldc), 7.0, device="cuda")
    out16 = torch.zeros((3, m, ldc), device="cuda", dtype=torch.float16)
    with _Tuning(L, gvs_slices=slices, gvs_grid=grid):
        pkg.check(L.ns_hip_fusion_qkv_forward_h(dA.data_ptr(), dA16.data_ptr(), wts[0].h, wts[1].h, wts[2].h, out.data_ptr(),
                                                out16.data_ptr(), m, k, ldc, st))
        torch.cuda.synchronize()
    a, f, f16 = dA.cpu().numpy(), out.cpu().numpy(), out16.cpu().numpy()
    for i, n in enumerate(ns):
        ref = nso.gemm_f64(a, blobs[i])
        assert nso.rel_l2(f[i][:, :n], ref) < TOL, (i, n, slices)
        assert np.array_equal(f16[i][:, :n], f[i][:, :n].astype(np.float16))
        assert (f[i][:, n:] == 7.0).all(), (i, n)
    for w in wts:
        w.free()


@pytest.mark.parametrize("n,k,m", [(4096, 14336, 8), (14336, 4096, 8), (1024, 4096, 8), (4096, 4096, 16), (4096, 11008, 16)])
def test_config4_shapes_full_size(L, pkg, nso, n, k, m):
    """BASELINE config 4 (Mistral-7B NF4 g128, 8 rows) at its real widths, the planner's own choice: 4096 x 14336 cannot
    hold 8 x 14336 fp16 in LDS and must be split; every output column against the fp64 oracle"""
    import torch
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16, False, seed=n + k)
    g = torch.Generator(device="cuda").manual_seed(m)
    dA = torch.randn((m, k), generator=g, device="cuda")
    out = _fwd(L, pkg, wt, dA, m, k, n)
    ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
    assert nso.rel_l2(out, ref) < TOL and nso.rel_l2(out, ref16) < 6e-4, (n, k, m, nso.rel_l2(out, ref16))
    assert np.max(np.abs(out - ref)) < 8e-3 * np.sqrt(np.mean(ref ** 2))
    wt.free()


@pytest.mark.parametrize("n,k", [(1280, 8192), (8192, 1024), (3584, 8192), (8192, 3584)])
@pytest.mark.parametrize("slices", [0, 4])
def test_single_row_when_asked_for(L, pkg, nso, n, k, slices):
    """one row (config 5's per-rank shard shapes) through this kernel when the knob selects it (gvs = 2)"""
    import torch
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=n * 3 + k)
    g = torch.Generator(device="cuda").manual_seed(2)
    dA = torch.randn((1, k), generator=g, device="cuda")
    with _Tuning(L, gvs=2, gvs_slices=slices):
        out = _fwd(L, pkg, wt, dA, 1, k, n)
    ref, ref16 = nso.gemm_f64_pair(dA.cpu().numpy(), blob)
    assert nso.rel_l2(out, ref) < TOL and nso.rel_l2(out, ref16) < 3e-5, (n, k, slices, nso.rel_l2(out, ref16))
    wt.free()


def test_fp32_only_caller_gets_the_same_bits(L, pkg, nso):
    """no fp16 shadow (bestla_device_f32f32_forward, ne_bestla.h:110-112): one conversion pass, then the same launch"""
    import torch
    n, k, m = 528, 3072, 7
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=3)
    dA = torch.randn((m, k), device="cuda")
    a = _fwd(L, pkg, wt, dA, m, k, n, shadow=True)
    b = _fwd(L, pkg, wt, dA, m, k, n, shadow=False)
    assert np.array_equal(a.view(np.int32), b.view(np.int32))
    wt.free()


def test_under_graph_capture(L, pkg, nso):
    """split-K scratch is taken from the per-stream pool, safe to bake into a graph; replays see new activations"""
    import torch
    n, k, m = 512, 8192, 16
    blob, wt, _keep = _blob(L, pkg, nso, n, k, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8, False, seed=11)
    dA = torch.randn((m, k), device="cuda")
    dA16 = dA.half()
    dC = torch.zeros((m, n), device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pkg.check(L.ns_hip_f32f32_forward_h(dA.data_ptr(), dA16.data_ptr(), wt.h, dC.data_ptr(), None, m, k, n, 0, None, 0, s))
    for _ in range(2):
        dA.copy_(torch.randn((m, k), device="cuda"))
        dA16.copy_(dA.half())
        g.replay()
        torch.cuda.synchronize()
        assert nso.rel_l2(dC.cpu().numpy(), nso.gemm_f64(dA.cpu().numpy(), blob)) < TOL
    wt.free()
