"""GPU parity at the shapes of every BASELINE.json config: the weight is quantized and packed by the GPU
quantizer (bit-exact with the oracle, tests/test_gpu_parity.py), the forward is compared with the oracle's fp64 GEMM on
the same blob.  Sizes are chosen so that the oracle finishes in seconds (M and, for the widest matrices, N are reduced —
the kernels' tiling does not depend on them beyond the tile count)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star


def _case(L, pkg, nso, n, k, m, qt, st_dt, bs, comp, asym=False):
    import torch
    rng = np.random.default_rng(n * 3 + k + m)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(n + k)
    dW = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, bs, qt, st_dt, asym, comp, None)
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, bs, qt, st_dt, asym, comp, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    a = rng.standard_normal((m, k)).astype(np.float32)
    out = np.zeros((m, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    ref = nso.gemm_f64(a, blob)
    e = nso.rel_l2(out, ref)
    assert e < TOL, (n, k, m, e)
    L.ns_hip_cache_clear()


@pytest.mark.parametrize("n,k", [(2304, 768), (768, 768), (3072, 768), (768, 3072), (50257, 768)])
def test_config1_gpt2_small_q4_0_decode(L, pkg, nso, n, k):
    """config 1: 12-layer d=768 GPT-2-family decoder, int4 sym g32, batch 1 (no GPT-2 arch in the reference;
    these are the GEMM shapes such a model sends through bestla_f32f32_forward)."""
    _case(L, pkg, nso, n, k, 1, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)])
@pytest.mark.parametrize("m", [1, 4])
def test_config2_llama7b_q4_0_decode_full_size(L, pkg, nso, n, k, m):
    """config 2 — the headline workload — at FULL size: every distinct GEMM shape of Llama-2-7B (wq/wk/wv/wo, gate/up,
    down, lm_head), Q4_0 g32 bf16 scales, batch 1 (and 4: the widest row count of the M <= 4 envelope), against the
    ORACLE's streaming GEMV over the whole blob (nso.gemv_f32, the call bench.py times as cpu_baseline) — every output
    column, not a sample — and against the fp64 GEMM."""
    import torch
    rng = np.random.default_rng(n + k + m)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(n * 7 + k)
    dW = torch.randn((n, k), generator=g, device="cuda") * 0.02
    size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
    dBlob = torch.zeros(size, dtype=torch.uint8, device="cuda")
    pkg.check(L.ns_hip_quant_pack_device(dBlob.data_ptr(), dW.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False,
                                         pkg.COMP_INT8, True, st))
    torch.cuda.synchronize()
    blob = nso.aligned_bytes(size)
    blob[:] = dBlob.cpu().numpy()
    a = rng.standard_normal((m, k)).astype(np.float32)
    out = np.zeros((m, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(out), m, n, k, k, n, None)
    ref32 = nso.gemv_f32(a, blob)
    e32 = nso.rel_l2(out, ref32)
    assert e32 < TOL, (n, k, m, e32)
    # per-column: no single output may be off by more than 1e-3 of the output scale (a sampled-column check would
    # miss a bad tile)
    assert np.max(np.abs(out - ref32)) < 1e-3 * np.sqrt(np.mean(ref32.astype(np.float64) ** 2)) * 8
    ref64 = nso.gemm_f64(a, blob)
    assert nso.rel_l2(out, ref64) < TOL
    L.ns_hip_cache_clear()


@pytest.mark.parametrize("n,k", [(4096, 4096), (2752, 4096), (4096, 2752)])
def test_config3_llama7b_int8_prefill(L, pkg, nso, n, k):
    """config 3: INT8 weights, fp16 compute, prefill (M > 64 -> gemm2_kernel); M = 192 and a quarter of the FFN width
    keep the fp64 oracle fast."""
    _case(L, pkg, nso, n, k, 192, pkg.S8, pkg.BF16, 32, pkg.COMP_F32)
    _case(L, pkg, nso, n, k, 192, pkg.S8, pkg.F32, 128, pkg.COMP_F32)


@pytest.mark.parametrize("n,k", [(4096, 4096), (1024, 4096), (3584, 4096), (4096, 3584)])
def test_config4_mistral7b_nf4_g128_batch8(L, pkg, nso, n, k):
    """config 4: NF4 RTN g128, batch 8 decode; wk/wv are 1024 wide (GQA), FFN width 14336 sampled at a quarter."""
    _case(L, pkg, nso, n, k, 8, pkg.F4_NF4, pkg.BF16, 128, pkg.COMP_BF16)


@pytest.mark.parametrize("n,k", [(1024, 8192), (128, 8192), (8192, 1024), (3584, 8192), (8192, 3584)])
def test_config5_llama70b_q4_0_tp8_rank_shapes(L, pkg, nso, n, k):
    """config 5: the per-rank shards of Llama-2-70B under TP = 8 (wq 8192->1024, wk/wv 8192->128, wo K = 1024,
    w1/w3 8192->3584, w2 K = 3584), batch 1."""
    _case(L, pkg, nso, n, k, 1, pkg.S4, pkg.BF16, 32, pkg.COMP_INT8)
