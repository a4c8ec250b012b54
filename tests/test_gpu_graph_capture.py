"""HIP-graph capture on a stream the library has never seen: every path that needs device scratch (fp16 copy of A,
split-K partials, attention partials, shuffled activations) must allocate it inside the capture without invalidating it,
keep it alive for the replays, and give the same numbers as the eager launch."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_capture_on_fresh_stream_matches_eager(L, pkg, nso):
    import torch
    rng = np.random.default_rng(4)
    n, k, bs = 256, 1024, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    wt = pkg.Weight.from_host_blob(nso.ptr(blob))
    outs = {}
    keep = []   # tensors a captured graph points at must outlive it (torch.cuda.graph() empties the cache on entry)
    for m in (130, 400):   # split-K with few output tiles, plain tiling with more
        a = torch.randn((m, k), device="cuda")
        c_eager = torch.zeros((m, n), device="cuda")
        c_graph = torch.zeros((m, n), device="cuda")
        keep += [a, c_eager, c_graph]
        side = torch.cuda.Stream()          # eager result on one fresh stream ...
        with torch.cuda.stream(side):
            pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c_eager.data_ptr(), m, k, n, 0, None, 0,
                                              C.c_void_p(side.cuda_stream)))
        side.synchronize()
        g = torch.cuda.CUDAGraph()          # ... the captured one on torch's capture stream, first use of it
        with torch.cuda.graph(g):
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c_graph.data_ptr(), m, k, n, 0, None, 0, s))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c_eager, c_graph)
        ref = nso.gemm_f64(a.cpu().numpy(), blob)
        assert nso.rel_l2(c_graph.cpu().numpy(), ref) < 1e-3
        # new activations, same graph: the scratch buffers baked into it are still there
        a.copy_(torch.randn((m, k), device="cuda"))
        g.replay()
        torch.cuda.synchronize()
        assert nso.rel_l2(c_graph.cpu().numpy(), nso.gemm_f64(a.cpu().numpy(), blob)) < 1e-3
        outs[m] = g   # keep the graphs alive while later captures grow the scratch
    # a larger request on the same capture stream later must not free what the first graph uses
    m = 130
    a = torch.randn((m, k), device="cuda")
    c = torch.zeros((m, n), device="cuda")
    keep += [a, c]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c.data_ptr(), m, k, n, 0, None, 0, s))
    for g in list(outs.values()) + [g2]:
        g.replay()
    torch.cuda.synchronize()
    assert nso.rel_l2(c.cpu().numpy(), nso.gemm_f64(a.cpu().numpy(), blob)) < 1e-3


def test_weight_prefetch_branch_leaves_results_unchanged(L, pkg, nso):
    """ns_hip_weight_prefetch on a forked stream inside a capture (the decode chain's second graph branch): a pure
    cache hint — the GEMV result is bit-identical with and without it, any offset / length / grid is accepted."""
    import torch
    rng = np.random.default_rng(9)
    n, k, bs = 512, 2048, 32
    w = (rng.standard_normal((n, k)) * 0.02).astype(np.float32)
    blob = nso.quant_pack(w, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    wt = pkg.Weight.from_host_blob(nso.ptr(blob))
    a = torch.randn((1, k), device="cuda")
    c0 = torch.zeros((1, n), device="cuda")
    c1 = torch.zeros((1, n), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c0.data_ptr(), 1, k, n, 0, None, 0, st))
    torch.cuda.synchronize()
    # eager: odd offset, oversized length, zero / huge grids
    for off, nbytes, grid in ((0, 1 << 40, 0), (17, 4096, 1), (1 << 40, 16, 8), (0, 0, 64), (32, 100000, 100000)):
        pkg.check(L.ns_hip_weight_prefetch(wt.h, off, nbytes, grid, st))
    torch.cuda.synchronize()
    assert L.ns_hip_weight_prefetch(None, 0, 16, 1, st) != 0   # null weight is an error, not a crash
    L.ns_hip_reset_error()
    pf = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        pf.wait_stream(cur)
        pkg.check(L.ns_hip_weight_prefetch(wt.h, 0, wt.stream_bytes, 64, C.c_void_p(pf.cuda_stream)))
        s = C.c_void_p(cur.cuda_stream)
        pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c1.data_ptr(), 1, k, n, 0, None, 0, s))
        cur.wait_stream(pf)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(c0, c1)
    assert nso.rel_l2(c1.cpu().numpy(), nso.gemm_f64(a.cpu().numpy(), blob)) < 1e-3
