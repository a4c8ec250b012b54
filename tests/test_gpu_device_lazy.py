"""Lazy peephole of the device route (csrc/ns_device.hip, round 4): rms_norm and silu nodes are recorded, a multiply that consumes the
recorded result runs ONE kernel writing both tensors; everything else launches the recorded node first.  Every tensor must hold
the bits the node-by-node kernels write."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dims(t):
    ne = list(reversed(t.shape)) + [1] * (4 - t.dim())
    nb, s = [], 4
    for e in ne:
        nb.append(s)
        s *= e
    return (C.c_longlong * 4)(*ne), (C.c_longlong * 4)(*nb)


def _mul(L, fn, a, b, d, st, is_mul=None):
    ne0, nb0 = _dims(a)
    ne1, nb1 = _dims(b)
    nb1 = (C.c_longlong * 4)(*[0 if (i > 0 and ne1[i] == 1) else nb1[i] for i in range(4)])
    _, nbd = _dims(d)
    args = [a.data_ptr(), b.data_ptr(), d.data_ptr(), ne0, nb0, ne1, nb1, nbd, st]
    return fn(*([is_mul] + args if is_mul is not None else args))


@pytest.fixture()
def fns(L):
    vp, ll4 = C.c_void_p, C.POINTER(C.c_longlong)
    L.ns_hip_lazy_flush.restype = C.c_int
    L.ns_hip_lazy_rms_norm.argtypes = [C.c_int, C.c_int, C.c_float, vp, vp, vp]
    L.ns_hip_lazy_silu.argtypes = [vp, vp, C.c_size_t, vp]
    L.ns_hip_lazy_mul.argtypes = [vp, vp, vp, ll4, ll4, ll4, ll4, ll4, vp]
    L.ns_hip_binary_nd_f32.argtypes = [C.c_int, vp, vp, vp, ll4, ll4, ll4, ll4, ll4, vp]
    L.ns_hip_layernormalization.argtypes = [C.c_int, C.c_int, C.c_bool, C.c_float, vp, vp, vp]
    L.ns_hip_silu_f32.argtypes = [vp, vp, C.c_size_t, vp]
    return L


@pytest.mark.parametrize("rows,cols", [(1, 4096), (5, 4096), (3, 5120), (2, 100)])
def test_norm_then_weight_is_one_launch_with_both_tensors(fns, pkg, rows, cols):
    import torch
    L = fns
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    x = torch.randn((rows, cols), generator=g, device="cuda") * 3
    gam = torch.randn((cols,), generator=g, device="cuda")
    a_ref, b_ref = torch.empty_like(x), torch.empty_like(x)
    pkg.check(L.ns_hip_layernormalization(rows, cols, True, 1e-5, x.data_ptr(), a_ref.data_ptr(), st))
    pkg.check(_mul(L, L.ns_hip_binary_nd_f32, a_ref, gam, b_ref, st, 1))
    a, b = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
    pkg.check(L.ns_hip_lazy_rms_norm(rows, cols, 1e-5, x.data_ptr(), a.data_ptr(), st))
    torch.cuda.synchronize()
    assert float((a - 7.0).abs().sum()) == 0.0  # recorded, not launched
    pkg.check(_mul(L, L.ns_hip_lazy_mul, a, gam, b, st))
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref) and torch.equal(b, b_ref)


def test_silu_then_up_in_either_operand_order(fns, pkg):
    import torch
    L = fns
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(5)
    gate = torch.randn((3, 11008), generator=g, device="cuda") * 4
    up = torch.randn((3, 11008), generator=g, device="cuda")
    s_ref, p_ref = torch.empty_like(gate), torch.empty_like(gate)
    pkg.check(L.ns_hip_silu_f32(gate.data_ptr(), s_ref.data_ptr(), gate.numel(), st))
    pkg.check(_mul(L, L.ns_hip_binary_nd_f32, s_ref, up, p_ref, st, 1))
    for silu_first in (True, False):
        s, p = torch.full_like(gate, 7.0), torch.full_like(gate, 7.0)
        pkg.check(L.ns_hip_lazy_silu(gate.data_ptr(), s.data_ptr(), gate.numel(), st))
        pkg.check(_mul(L, L.ns_hip_lazy_mul, s, up, p, st) if silu_first else _mul(L, L.ns_hip_lazy_mul, up, s, p, st))
        torch.cuda.synchronize()
        assert torch.equal(s, s_ref) and torch.equal(p, p_ref), silu_first


def test_anything_else_launches_the_recorded_node_first(fns, pkg):
    import torch
    L = fns
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((2, 4096), generator=g, device="cuda")
    other = torch.randn((2, 4096), generator=g, device="cuda")
    a_ref = torch.empty_like(x)
    pkg.check(L.ns_hip_layernormalization(2, 4096, True, 1e-6, x.data_ptr(), a_ref.data_ptr(), st))
    # an add that reads the normalised tensor; a multiply by an unrelated tensor; a flush on its own; a recorded node replaced by another
    a, d = torch.full_like(x, 7.0), torch.empty_like(x)
    pkg.check(L.ns_hip_lazy_rms_norm(2, 4096, 1e-6, x.data_ptr(), a.data_ptr(), st))
    pkg.check(_mul(L, L.ns_hip_binary_nd_f32, a, other, d, st, 0))
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref) and torch.equal(d, a_ref + other)
    a.fill_(7.0)
    pkg.check(L.ns_hip_lazy_rms_norm(2, 4096, 1e-6, x.data_ptr(), a.data_ptr(), st))
    pkg.check(_mul(L, L.ns_hip_lazy_mul, other, other, d, st))  # does not consume the recorded node
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref) and torch.equal(d, other * other)
    a.fill_(7.0)
    pkg.check(L.ns_hip_lazy_rms_norm(2, 4096, 1e-6, x.data_ptr(), a.data_ptr(), st))
    assert L.ns_hip_lazy_flush() == 0
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref)
    a.fill_(7.0)
    s = torch.empty_like(x)
    pkg.check(L.ns_hip_lazy_rms_norm(2, 4096, 1e-6, x.data_ptr(), a.data_ptr(), st))
    pkg.check(L.ns_hip_lazy_silu(x.data_ptr(), s.data_ptr(), x.numel(), st))  # records silu, launches the norm
    L.bestla_device_sync.argtypes = [C.c_void_p]
    L.bestla_device_sync(st)  # launches the silu
    assert torch.equal(a, a_ref) and torch.equal(s, x / (1 + torch.exp(-x)) ) or torch.allclose(s, torch.nn.functional.silu(x), rtol=1e-6, atol=1e-7)


def test_in_place_multiplies_are_not_fused(fns, pkg):
    """ne_mul_inplace (dst == the recorded node's output): the fused kernels store the node's result and the product
    to two addresses — aliased, the tensor would end up holding the norm / silu without the multiply.  The recorded node is
    launched first and the plain multiply runs over it."""
    import torch
    L = fns
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(77)
    rows, cols = 3, 4096
    x = torch.randn((rows, cols), generator=g, device="cuda") * 2
    gam = torch.randn((cols,), generator=g, device="cuda")
    a_ref, b_ref = torch.empty_like(x), torch.empty_like(x)
    pkg.check(L.ns_hip_layernormalization(rows, cols, True, 1e-5, x.data_ptr(), a_ref.data_ptr(), st))
    pkg.check(_mul(L, L.ns_hip_binary_nd_f32, a_ref, gam, b_ref, st, 1))
    a = torch.full_like(x, 7.0)
    pkg.check(L.ns_hip_lazy_rms_norm(rows, cols, 1e-5, x.data_ptr(), a.data_ptr(), st))
    pkg.check(_mul(L, L.ns_hip_lazy_mul, a, gam, a, st))  # in place
    torch.cuda.synchronize()
    assert torch.equal(a, b_ref)
    # silu(gate) * up, written over silu(gate)
    gate, up = torch.randn((rows, cols), generator=g, device="cuda"), torch.randn((rows, cols), generator=g, device="cuda")
    s_ref, p_ref = torch.empty_like(gate), torch.empty_like(gate)
    pkg.check(L.ns_hip_silu_f32(gate.data_ptr(), s_ref.data_ptr(), gate.numel(), st))
    pkg.check(_mul(L, L.ns_hip_binary_nd_f32, s_ref, up, p_ref, st, 1))
    s = torch.full_like(gate, 7.0)
    pkg.check(L.ns_hip_lazy_silu(gate.data_ptr(), s.data_ptr(), gate.numel(), st))
    pkg.check(_mul(L, L.ns_hip_lazy_mul, s, up, s, st))
    torch.cuda.synchronize()
    assert torch.equal(s, p_ref)


def test_dense_and_strided_binary_nodes_compute_the_same_values(L, pkg):
    """ns_hip_binary_nd_f32: dense tensors of one shape and dense-tensor x row-vector take a four-elements-per-thread kernel (round 5), everything else the
    strided one; both are exact fp32 adds / multiplies."""
    import ctypes as C
    import numpy as np
    import torch
    ll4 = C.POINTER(C.c_longlong)
    L.ns_hip_binary_nd_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, ll4, ll4, ll4, ll4, ll4, C.c_void_p]
    L.ns_hip_binary_nd_f32.restype = C.c_int
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ll = lambda *v: (C.c_longlong * 4)(*v)
    rng = np.random.default_rng(4)
    rows, cols = 300, 512
    a = torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.standard_normal(cols).astype(np.float32)).cuda()
    ne, nb = ll(cols, rows, 1, 1), ll(4, 4 * cols, 4 * cols * rows, 4 * cols * rows)
    d = torch.zeros_like(a)
    for op, ref in ((0, a + b), (1, a * b)):  # dense, same shape
        d.zero_()
        pkg.check(L.ns_hip_binary_nd_f32(op, a.data_ptr(), b.data_ptr(), d.data_ptr(), ne, nb, ne, nb, nb, st))
        torch.cuda.synchronize()
        assert torch.equal(d, ref)
    d.zero_()  # dense x row vector (the mul by a norm weight)
    pkg.check(L.ns_hip_binary_nd_f32(1, a.data_ptr(), g.data_ptr(), d.data_ptr(), ne, nb, ll(cols, 1, 1, 1), ll(4, 4 * cols, 4 * cols, 4 * cols), nb, st))
    torch.cuda.synchronize()
    assert torch.equal(d, a * g)
    # a view with a row stride (every second row of a taller tensor): the strided kernel
    tall = torch.from_numpy(rng.standard_normal((2 * rows, cols)).astype(np.float32)).cuda()
    d.zero_()
    pkg.check(L.ns_hip_binary_nd_f32(0, tall.data_ptr(), b.data_ptr(), d.data_ptr(), ne, ll(4, 8 * cols, 8 * cols * rows, 8 * cols * rows), ne, nb, nb, st))
    torch.cuda.synchronize()
    assert torch.equal(d, tall[0::2] + b)
    # in place (dst == src0), as ne_add_inplace hands it over
    a2 = a.clone()
    pkg.check(L.ns_hip_binary_nd_f32(0, a2.data_ptr(), b.data_ptr(), a2.data_ptr(), ne, nb, ne, nb, nb, st))
    torch.cuda.synchronize()
    assert torch.equal(a2, a + b)


@pytest.mark.parametrize("seq,heads,hs,n_ctx,pos", [(300, 8, 128, 512, 100), (70, 4, 64, 256, 0), (33, 4, 128, 64, 5)])
def test_prompt_sized_cache_writes_take_the_vector_and_the_transposing_copy(L, pkg, seq, heads, hs, n_ctx, pos):
    """The two kv-cache writes of a prompt as the reference's device graph issues them (llama.cpp:241-285: ne_cpy of the permuted K rows into
    [head][n_ctx][hs] cells, of the permuted V rows into the TRANSPOSED [head][hs][n_ctx] cells).  From 65 536 elements on ns_hip_dup_f32 serves the first with
    16-byte accesses (dup_vec4_kernel) and the second through 32 x 32 LDS tiles (dup_transpose_kernel; round 6: dup_kernel read 4 bytes out of every 16 KB row
    per thread); below that the element-wise kernel.  Every cell, and nothing but the written cells, against numpy."""
    import torch
    rng = np.random.default_rng(seq)
    k = rng.standard_normal((seq, heads, hs)).astype(np.float32)
    v = rng.standard_normal((seq, heads, hs)).astype(np.float32)
    kc = np.full((heads, n_ctx, hs), 7.0, np.float32)
    vc = np.full((heads, hs, n_ctx), 7.0, np.float32)
    dk, dv, dkc, dvc = (torch.from_numpy(a).cuda() for a in (k, v, kc, vc))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ll4 = C.POINTER(C.c_longlong)
    L.ns_hip_dup_f32.argtypes = [C.c_void_p, C.c_void_p, ll4, ll4, ll4, C.c_bool, C.c_void_p]
    ll = lambda *a: (C.c_longlong * 4)(*a)
    f4 = 4
    # K: destination extents (hs, seq, heads, 1); the source is indexed with the destination's coordinates
    pkg.check(L.ns_hip_dup_f32(dk.data_ptr(), dkc.data_ptr() + pos * hs * f4, ll(hs, seq, heads, 1), ll(f4, heads * hs * f4, hs * f4, seq * heads * hs * f4),
                               ll(f4, hs * f4, n_ctx * hs * f4, heads * n_ctx * hs * f4), False, st))
    # V: destination extents (seq, hs, heads, 1) of the transposed cache
    pkg.check(L.ns_hip_dup_f32(dv.data_ptr(), dvc.data_ptr() + pos * f4, ll(seq, hs, heads, 1), ll(heads * hs * f4, f4, hs * f4, seq * heads * hs * f4),
                               ll(f4, n_ctx * f4, hs * n_ctx * f4, heads * hs * n_ctx * f4), False, st))
    torch.cuda.synchronize()
    kc[:, pos:pos + seq, :] = k.transpose(1, 0, 2)
    vc[:, :, pos:pos + seq] = v.transpose(1, 2, 0)
    assert np.array_equal(dkc.cpu().numpy(), kc)
    assert np.array_equal(dvc.cpu().numpy(), vc)
