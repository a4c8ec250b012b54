"""Load path of the reference's device loader (model_files.h:1515-1527 -> bestla_device_load_storage): the
streaming layout is written INTO the slice the graph reserved for the tensor (no second copy of the model in HBM) whenever it
is no larger than the blob, nothing is synchronised per tensor, and the forward through the loaded storage equals — bit for
bit — the forward through the host-blob entry on the same blob."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(L):
    vp, sz = C.c_void_p, C.c_size_t
    L.bestla_create_device.restype = vp
    L.bestla_create_device.argtypes = [C.c_bool]
    L.bestla_get_device_queue.restype = vp
    L.bestla_get_device_queue.argtypes = [vp]
    L.bestla_release_device.argtypes = [vp]
    L.bestla_device_malloc.restype = vp
    L.bestla_device_malloc.argtypes = [sz, vp]
    L.bestla_device_free.argtypes = [vp, vp]
    L.bestla_device_storage_size.restype = sz
    L.bestla_device_load_storage.argtypes = [vp, vp, vp, vp]
    L.bestla_device_f32f32_forward.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.bestla_device_memcpy_sync.argtypes = [vp, vp, sz, vp]
    L.bestla_device_sync.argtypes = [vp]
    L.ns_hip_device_load_stats.argtypes = [vp]
    L.ns_hip_device_storage_release.argtypes = [vp]
    L.ns_hip_weight_is_external.argtypes = [vp]


CASES = [("int4_g32", "S4", 32, "BF16", False, True), ("int8_g128_asym", "S8", 128, "F32", True, True), ("nf4_g64", "F4_NF4", 64, "BF16", False, True),
         ("int3_g32 (widened to nibbles: larger than its blob)", "S3", 32, "BF16", False, False)]


@pytest.mark.parametrize("name,qt,bs,st,asym,in_slice", CASES, ids=[c[0].split(" ")[0] for c in CASES])
def test_storage_lands_in_the_reserved_slice_and_computes_the_same(L, pkg, nso, name, qt, bs, st, asym, in_slice):
    _dev(L)
    rng = np.random.default_rng(bs)
    n, k, m = 1000, 1024, 3
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    blob = nso.quant_pack(w, bs, getattr(nso, qt), getattr(nso, st), asym, nso.CORE_AVX512_VNNI_KB if qt in ("S4", "S8", "S3") else nso.CORE_AVX512F)
    a = rng.standard_normal((m, k)).astype(np.float32)
    want = np.zeros((m, n), np.float32)
    L.bestla_f32f32_forward(nso.ptr(a), nso.ptr(blob), nso.ptr(want), m, n, k, k, n, None)
    L.ns_hip_cache_clear()
    dev = L.bestla_create_device(False)
    q = L.bestla_get_device_queue(dev)
    before = (C.c_uint64 * 6)()
    L.ns_hip_device_load_stats(before)
    size = int(np.frombuffer(blob[:8].tobytes(), np.uint64)[0])
    slice_bytes = (size + 255) // 256 * 256
    dptr = L.bestla_device_malloc(slice_bytes, q)
    stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
    hb = blob.copy()
    L.bestla_device_load_storage(nso.ptr(hb), nso.ptr(stor), dptr, q)
    hb[:] = 0xee   # the caller frees the host blob right after the call (model_files.h:1526): it must not be read again
    after = (C.c_uint64 * 6)()
    L.ns_hip_device_load_stats(after)
    assert after[0] == before[0] + 1 and after[5] >= 1          # recorded, and still waiting for the one synchronisation
    assert (after[2] > before[2]) == in_slice and (after[3] > before[3]) == (not in_slice), (name, list(after))
    da = L.bestla_device_malloc(a.nbytes, q)
    dc = L.bestla_device_malloc(want.nbytes, q)
    L.bestla_device_memcpy_sync(da, nso.ptr(a), a.nbytes, q)
    L.bestla_device_f32f32_forward(da, nso.ptr(stor), dc, m, n, k, k, n, None, q)
    out = np.zeros_like(want)
    L.bestla_device_memcpy_sync(nso.ptr(out), dc, out.nbytes, q)
    L.ns_hip_device_load_stats(after)
    assert after[5] == 0                                          # completed by the first forward
    assert np.array_equal(out.view(np.int32), want.view(np.int32)), (name, float(np.abs(out - want).max()))
    assert nso.rel_l2(out, nso.gemm_f64(a, blob)) < 1e-3
    L.ns_hip_device_storage_release(nso.ptr(stor))
    for p in (da, dc, dptr):
        L.bestla_device_free(p, q)
    L.bestla_release_device(dev)


def test_many_tensors_one_synchronisation(L, pkg, nso):
    """a model's worth of loads back to back (different sizes: the staging buffer grows), host blobs scribbled over after each
    call, every weight checked afterwards"""
    _dev(L)
    rng = np.random.default_rng(3)
    dev = L.bestla_create_device(False)
    q = L.bestla_get_device_queue(dev)
    shapes = [(256, 512), (4096, 1024), (512, 4096), (11008, 512), (64, 256), (2048, 2048)] * 2
    items = []
    for i, (n, k) in enumerate(shapes):
        w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
        blob = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
        size = int(np.frombuffer(blob[:8].tobytes(), np.uint64)[0])
        dptr = L.bestla_device_malloc((size + 255) // 256 * 256, q)
        stor = np.zeros(int(L.bestla_device_storage_size()), np.uint8)
        hb = blob.copy()
        L.bestla_device_load_storage(nso.ptr(hb), nso.ptr(stor), dptr, q)
        hb[:] = i
        items.append((n, k, blob, stor, dptr))
    st = (C.c_uint64 * 6)()
    L.ns_hip_device_load_stats(st)
    assert st[5] == len(shapes)
    for n, k, blob, stor, dptr in items:
        a = rng.standard_normal((2, k)).astype(np.float32)
        da, dc = L.bestla_device_malloc(a.nbytes, q), L.bestla_device_malloc(2 * n * 4, q)
        L.bestla_device_memcpy_sync(da, nso.ptr(a), a.nbytes, q)
        L.bestla_device_f32f32_forward(da, nso.ptr(stor), dc, 2, n, k, k, n, None, q)
        out = np.zeros((2, n), np.float32)
        L.bestla_device_memcpy_sync(nso.ptr(out), dc, out.nbytes, q)
        assert nso.rel_l2(out, nso.gemm_f64(a, blob)) < 1e-3, (n, k)
        L.ns_hip_device_storage_release(nso.ptr(stor))
        for p in (da, dc, dptr):
            L.bestla_device_free(p, q)
    L.bestla_release_device(dev)


def test_host_side_tp_shards_compute_the_unsharded_result(L, pkg, nso):
    """load-time tensor parallelism (model_files.h:145-190 rules, :1593-1640 per-rank cut): every rank cuts its shard out of the
    blob on the HOST (ns_bestla_split_weight via parallel.shard_blob) and uploads only that; a llama FFN + attention-output
    pair computed from the shards — ROW shards side by side, COLUMN shards summed like the all-reduce does — equals the unsharded
    computation"""
    import torch
    from neural_speed_amd import parallel as par
    rng = np.random.default_rng(11)
    d, ff, world, m = 1024, 2816, 4, 2
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mk = lambda n, k: nso.quant_pack((rng.standard_normal((n, k)) * k ** -0.5).astype(np.float32), 32, nso.S4, nso.BF16, False,
                                     nso.CORE_AVX512_VNNI_KB)
    b1, b3, b2 = mk(ff, d), mk(ff, d), mk(d, ff)
    x = torch.from_numpy(rng.standard_normal((m, d)).astype(np.float32)).cuda()

    def fwd(wt, a, n, k):
        c = torch.empty((a.shape[0], n), device="cuda")
        pkg.check(L.ns_hip_f32f32_forward(a.data_ptr(), wt.h, c.data_ptr(), a.shape[0], k, n, 0, None, 0, st))
        torch.cuda.synchronize()
        return c

    def ffn(blobs, n_ff):
        w1, w3, w2 = [pkg.Weight.from_host_blob(nso.ptr(b)) for b in blobs]
        h1, h3 = fwd(w1, x, n_ff, d), fwd(w3, x, n_ff, d)
        t = (torch.nn.functional.silu(h1) * h3).contiguous()
        y = fwd(w2, t, d, n_ff)
        for w in (w1, w3, w2):
            w.free()
        return t, y
    t_full, y_full = ffn((b1, b3, b2), ff)
    up_before = sum(b.size for b in (b1, b3, b2))
    y_sum = torch.zeros_like(y_full)
    uploaded = 0
    for rank in range(world):
        ctx = par.ParallelContext.__new__(par.ParallelContext)
        ctx.rank, ctx.world = rank, world
        s1 = ctx.shard_blob(b1, par.calc_split_type("layers.0.feed_forward.w1.weight"))
        s3 = ctx.shard_blob(b3, par.calc_split_type("layers.0.feed_forward.w3.weight"))
        s2 = ctx.shard_blob(b2, par.calc_split_type("layers.0.feed_forward.w2.weight"))
        uploaded += s1.size + s3.size + s2.size
        t_r, y_r = ffn((s1, s3, s2), ff // world)
        # a ROW shard's outputs are the unsharded outputs of its columns
        assert nso.rel_l2(t_r.cpu().numpy(), t_full[:, rank * ff // world:(rank + 1) * ff // world].cpu().numpy()) < 1e-6
        y_sum += y_r
    assert nso.rel_l2(y_sum.cpu().numpy(), y_full.cpu().numpy()) < 1e-5
    assert uploaded < 1.02 * up_before          # the ranks together upload the model once, not world times
    L.ns_hip_cache_clear()
