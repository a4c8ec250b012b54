"""Expert-indexed matmul with device-side routing (csrc/ns_moe.hip) against the reference semantics of
ne_compute_forward_mul_mat_id_q_f32_bestla (ne_layers.c:7783-7916): dst[t] = src1[t] . W[ids[t][id]], one
bestla_f32f32_forward per (token, expert) — restated with the oracle's fp64 product per row."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMATS = [  # qtype, scale dtype, asym, group, core
    ("S4", "BF16", False, 32, "CORE_AVX512_VNNI_KB"),
    ("S4", "F32", True, 128, "CORE_AVX512F"),
    ("S8", "BF16", False, 32, "CORE_AVX512_VNNI_KB"),
    ("F4_NF4", "BF16", False, 64, "CORE_AVX512F"),
]


def _group(L, pkg, nso, rng, n_as, n, k, qt, st, asym, bs, core):
    blobs, weights = [], []
    for _ in range(n_as):
        w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
        blobs.append(nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), asym, getattr(nso, core)))
        weights.append(pkg.Weight.from_host_blob(nso.ptr(blobs[-1])))
    arr = (C.c_void_p * n_as)(*[w.h for w in weights])
    g = L.ns_hip_expert_group_create(arr, n_as)
    assert g, pkg.last_error()
    return blobs, weights, g


@pytest.mark.parametrize("qt,st,asym,bs,core", FORMATS)
def test_mul_mat_id_matches_per_token_forwards(L, pkg, nso, qt, st, asym, bs, core):
    import torch
    rng = np.random.default_rng(len(qt) * 13 + bs)
    n_as, n, k, m, topk = 4, 200, 832, 7, 2   # ragged N, K not a multiple of the 128-deep k-step
    blobs, weights, g = _group(L, pkg, nso, rng, n_as, n, k, qt, st, asym, bs, core)
    a = rng.standard_normal((m, k)).astype(np.float32)
    ids = rng.integers(0, n_as, size=(m, topk)).astype(np.int32)
    dA, dI = torch.from_numpy(a).cuda(), torch.from_numpy(ids).cuda()
    st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for sel in range(topk):
        dC = torch.full((m, n), 7.0, device="cuda")
        pkg.check(L.ns_hip_mul_mat_id(dA.data_ptr(), dI.data_ptr(), topk, sel, g, dC.data_ptr(), m, k, n, pkg.EPI_NONE,
                                      None, 0, st_))
        torch.cuda.synchronize()
        out = dC.cpu().numpy()
        ref = np.concatenate([nso.gemm_f64(a[t:t + 1], blobs[ids[t, sel]]) for t in range(m)], axis=0)
        assert nso.rel_l2(out, ref) < 1e-3
        ref16 = np.concatenate([nso.gemm_f64(a[t:t + 1], blobs[ids[t, sel]], a16=True) for t in range(m)], axis=0)
        # same fp16-rounded activations: only fp32 summation order differs — and, for the 4-bit float types at decode size, the
        # value table rounded to fp16 (the decode kernel's MFMA operand, like every other f4 launch of the library)
        assert nso.rel_l2(out, ref16) < (6e-4 if qt.startswith("F4") else 5e-5)
    L.ns_hip_expert_group_free(g)


def test_ffn_id_composition_in_a_graph_and_bad_ids(L, pkg, nso):
    """gate (SiLU) -> up (Mul) -> down through three mul_mat_id calls captured in ONE graph with the ids on the device
    (ffn_id_silu, ne_layers.c:8053-8170); an out-of-range id zeroes its row instead of asserting."""
    import torch
    rng = np.random.default_rng(3)
    n_as, d, ff, m = 3, 256, 512, 4
    bg, wg, gg = _group(L, pkg, nso, rng, n_as, ff, d, "S4", "BF16", False, 32, "CORE_AVX512_VNNI_KB")
    bu, wu, gu = _group(L, pkg, nso, rng, n_as, ff, d, "S4", "BF16", False, 32, "CORE_AVX512_VNNI_KB")
    bd, wd, gd = _group(L, pkg, nso, rng, n_as, d, ff, "S4", "BF16", False, 32, "CORE_AVX512_VNNI_KB")
    a = rng.standard_normal((m, d)).astype(np.float32)
    ids = np.array([[0], [2], [1], [2]], np.int32)
    dA, dI = torch.from_numpy(a).cuda(), torch.from_numpy(ids).cuda()
    t1 = torch.zeros((m, ff), device="cuda")
    t2 = torch.zeros((m, ff), device="cuda")
    out = torch.zeros((m, d), device="cuda")
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pkg.check(L.ns_hip_mul_mat_id(dA.data_ptr(), dI.data_ptr(), 1, 0, gg, t1.data_ptr(), m, d, ff, pkg.EPI_SILU, None, 0, s))
        pkg.check(L.ns_hip_mul_mat_id(dA.data_ptr(), dI.data_ptr(), 1, 0, gu, t2.data_ptr(), m, d, ff, pkg.EPI_MUL, t1.data_ptr(), ff, s))
        pkg.check(L.ns_hip_mul_mat_id(t2.data_ptr(), dI.data_ptr(), 1, 0, gd, out.data_ptr(), m, ff, d, pkg.EPI_NONE, None, 0, s))

    def expect(idv):
        rows = []
        for t in range(m):
            e = int(idv[t, 0])
            gate = nso.gemm_f64(a[t:t + 1], bg[e])
            up = nso.gemm_f64(a[t:t + 1], bu[e])
            h = (gate / (1 + np.exp(-gate)) * up).astype(np.float32)
            rows.append(nso.gemm_f64(h, bd[e]))
        return np.concatenate(rows, axis=0)

    gr.replay()
    torch.cuda.synchronize()
    assert nso.rel_l2(out.cpu().numpy(), expect(ids)) < 2e-3
    # new routing decision, same graph: only device memory changed
    ids2 = np.array([[1], [1], [0], [2]], np.int32)
    dI.copy_(torch.from_numpy(ids2))
    gr.replay()
    torch.cuda.synchronize()
    assert nso.rel_l2(out.cpu().numpy(), expect(ids2)) < 2e-3
    # out-of-range id: that row of the product is zero
    bad = torch.tensor([[0], [9], [-1], [1]], dtype=torch.int32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pkg.check(L.ns_hip_mul_mat_id(dA.data_ptr(), bad.data_ptr(), 1, 0, gg, t1.data_ptr(), m, d, ff, pkg.EPI_NONE, None, 0, s))
    torch.cuda.synchronize()
    r = t1.cpu().numpy()
    assert np.all(r[1] == 0) and np.all(r[2] == 0) and np.any(r[0] != 0) and np.any(r[3] != 0)
    # experts of different shapes do not form a group
    arr = (C.c_void_p * 2)(wg[0].h, wd[0].h)
    assert not L.ns_hip_expert_group_create(arr, 2)
    L.ns_hip_reset_error()
    for g in (gg, gu, gd):
        L.ns_hip_expert_group_free(g)


@pytest.mark.parametrize("qt,st,asym,bs,core", FORMATS[:3] + [("F4_NF4", "BF16", False, 64, "CORE_AVX512F")])
@pytest.mark.parametrize("epi", ["none", "silu", "mul"])
def test_mul_mat_id_at_prefill_size_groups_the_rows_by_expert(L, pkg, nso, qt, st, asym, bs, core, epi):
    """round 5: from 32 token rows on the rows are grouped by expert on the host (the reference groups them too: `matrix_rows`,
    ne_layers.c:7855-7866) and every expert's rows are ONE launch of the tiled GEMM instead of one weight stream per token; the epilogue
    is applied per token row by the scatter kernel.  Ragged groups (an expert with one row, an expert with none), an id outside the
    group (zero product, like the per-row kernels), every row against the oracle's fp64 product with ITS expert."""
    import torch
    rng = np.random.default_rng(len(qt) * 7 + bs + len(epi))
    n_as, n, k, m, topk = 5, 272, 512, 150, 2
    blobs, weights, g = _group(L, pkg, nso, rng, n_as, n, k, qt, st, asym, bs, core)
    a = rng.standard_normal((m, k)).astype(np.float32)
    ids = rng.integers(0, 3, size=(m, topk)).astype(np.int32)   # experts 0..2 share most rows
    ids[17, 1] = 3                                              # expert 3: exactly one row
    ids[40, 1] = 99                                             # outside the group; expert 4: no row at all
    d = rng.standard_normal((m, n)).astype(np.float32)
    dA, dI, dD = torch.from_numpy(a).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(d).cuda()
    st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = {"none": pkg.EPI_NONE, "silu": pkg.EPI_SILU, "mul": pkg.EPI_MUL}[epi]
    dC = torch.full((m, n + 3), 7.0, device="cuda")
    pkg.check(L.ns_hip_mul_mat_id(dA.data_ptr(), dI.data_ptr(), topk, 1, g, dC.data_ptr(), m, k, n + 3, code,
                                  dD.data_ptr() if epi == "mul" else None, n, st_))
    torch.cuda.synchronize()
    out = dC.cpu().numpy()
    assert np.all(out[:, n:] == 7.0)
    ref = np.zeros((m, n))
    for t in range(m):
        e = ids[t, 1]
        if 0 <= e < n_as:
            ref[t] = nso.gemm_f64(a[t:t + 1], blobs[e])[0]
    if epi == "silu":
        ref = ref / (1.0 + np.exp(-ref))
    if epi == "mul":
        ref = ref * d
    assert nso.rel_l2(out[:, :n], ref) < 1e-3
    assert np.all(out[40, :n] == 0.0)
    L.ns_hip_expert_group_free(g)
    for w in weights:
        w.free()
