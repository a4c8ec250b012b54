"""Tensor-parallel path on CPU: world_size-2 gloo processes exercise the communication layer
(neural-speed_amd/parallel.py, the replacement of parallel_context.{h,cpp}) and the reference's 1-D split rules,
with the ORACLE standing in for the GEMMs (tests may use it).  Checks that  all_reduce( A[:, Kr] * W[Kr, :] )
over the K-split and the concatenation over the N-split both equal the unsharded result."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
    import __graft_entry__ as ge
    import nso
    pkg = ge.load_package()
    from neural_speed_amd import parallel as par
    ctx = par.init_parallel_context("gloo")
    ws, rk = ctx.get_tp_size(), ctx.get_tp_rank()
    assert ws == 2 and par.get_tp_size() == 2 and par.is_master() == (rk == 0)
    # split rules (model_files.h:145-190)
    assert par.calc_split_type("layers.0.attention.wq.weight") == par.TENSOR_1D_ROW
    assert par.calc_split_type("layers.3.feed_forward.w2.weight") == par.TENSOR_1D_COLUMN
    assert par.calc_split_type("layers.3.attention.wo.weight") == par.TENSOR_1D_COLUMN
    assert par.calc_split_type("tok_embeddings.weight") == par.TENSOR_NO_CHANGE
    # broadcast of token ids exactly as llama.cpp:182-187 does it (int32 reinterpreted as float)
    ids = torch.tensor([1, 15043, 3186] if rk == 0 else [0, 0, 0], dtype=torch.int32)
    par.broadcast(ids.view(torch.float32))
    assert ids.tolist() == [1, 15043, 3186]
    rng = np.random.default_rng(5)  # same stream on both ranks
    d, ff, m, bs = 256, 512, 3, 32
    x = rng.standard_normal((m, d)).astype(np.float32)
    w1 = (rng.standard_normal((ff, d)) * 0.05).astype(np.float32)   # ROW split (N)
    w2 = (rng.standard_normal((d, ff)) * 0.05).astype(np.float32)   # COLUMN split (K)
    full1 = nso.quant_pack(w1, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    full2 = nso.quant_pack(w2, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
    h_full = nso.gemm_f64(x, full1)
    y_full = nso.gemm_f64(h_full.astype(np.float32), full2)
    # shards: slice the CANONICAL codes (what ns_hip_weight_slice does on the device) and re-pack
    n0, n1 = ctx.shard_range(ff, 16)
    q1, s1, _ = nso.unpack_canonical(full1)
    sh1 = nso.pack_q(q1[:, n0:n1], s1[:, n0:n1], None, bs, nso.S4, nso.BF16, nso.CORE_AVX512_VNNI_KB)
    h_loc = nso.gemm_f64(x, sh1)
    assert np.array_equal(h_loc, h_full[:, n0:n1])          # N split: bit-identical columns
    k0, k1 = ctx.shard_range(ff, 128)
    q2, s2, _ = nso.unpack_canonical(full2)
    sh2 = nso.pack_q(q2[k0:k1], s2[k0 // bs:k1 // bs], None, bs, nso.S4, nso.BF16, nso.CORE_AVX512_VNNI_KB)
    part = nso.gemm_f64(h_loc.astype(np.float32), sh2).astype(np.float32)
    y = torch.from_numpy(part.copy())
    par.reduce_add(y)                                         # ne_all_reduce (ne_layers.c:5466-5476)
    err = nso.rel_l2(y.numpy(), y_full)
    assert err < 1e-6, err
    # the reference's own sharding (dequantize -> slice -> re-quantize, model_files.h:1538-1563) stays close but is
    # not bit-identical: report it
    deq = nso.unpack_fp32(full2)[k0:k1]
    sh2r = nso.quant_pack(np.ascontiguousarray(deq), bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB, is_trans=False)
    yr = torch.from_numpy(nso.gemm_f64(h_loc.astype(np.float32), sh2r).astype(np.float32))
    par.reduce_add(yr)
    assert nso.rel_l2(yr.numpy(), y_full) < 5e-2
    # the peer-memory all-reduce cannot be set up without a GPU: every rank must notice, fall back TOGETHER, and
    # reduce_add must keep working through the process group (the same agreement path a failed IPC open takes)
    if not torch.cuda.is_available():
        assert ctx.enable_p2p(1 << 16) is False and not ctx.p2p_enabled()
        assert ctx.p2p_error() is False
        z = torch.full((8,), float(rk + 1))
        par.reduce_add(z)
        assert z.tolist() == [3.0] * 8
        ctx.disable_p2p()   # no-op
    par.barrier()
    # one result file per rank: the two ranks' stdout lines can interleave
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank" + str(rk) + ".ok"), "w").write(repr(float(err)))
    print("rank", rk, "ok", err)
''') % (ROOT, ROOT)


def test_tp2_gloo(tmp_path):
    script = tmp_path / "tp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for rk in range(2):
        assert (tmp_path / ("rank%d.ok" % rk)).exists(), r.stdout + r.stderr[-2000:]
