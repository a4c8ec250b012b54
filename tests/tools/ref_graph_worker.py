"""Worker of the reference-graph tests (fresh interpreter: the bestla_* provider must be loaded before libne_ref.so).
argv[1] = "mock": build + load tests/tools/mock_bestla_provider.c, check the marshalling.
argv[1] = "product": load neural-speed_amd/libns_hip.so, run ne_mul_mat / ne_ffn_silu over BTLA tensors through the
reference's graph executor and compare with the oracle (GPU needed)."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nso  # noqa: E402


class MockCall(C.Structure):
    _fields_ = [("which", C.c_int), ("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("lda", C.c_int), ("ldo", C.c_int),
                ("a", C.c_void_p), ("w", C.c_void_p), ("c", C.c_void_p), ("ws", C.c_void_p), ("w1", C.c_void_p),
                ("w2", C.c_void_p), ("w3", C.c_void_p), ("seq", C.c_int), ("fin", C.c_int), ("fmid", C.c_int),
                ("fout", C.c_int)]


def blobs(rng, shapes, bs=32):
    out = []
    for n, k in shapes:
        w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
        out.append(nso.quant_pack(w, bs, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB))
    return out


def decoder_layer_case(ne, rng):
    """A whole decoder layer as ONE reference graph (neref_decoder_layer) against an fp64 model of the same layer built
    from the dequantized weights."""
    T, d, heads, ff, eps, base = 6, 256, 4, 704, 1e-5, 10000.0
    hs = d // heads

    def qw(n, k):
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        b = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
        return b, nso.unpack_fp32(b).astype(np.float64)   # dequantized [k][n]
    (bq, Wq), (bk, Wk), (bv, Wv), (bo, Wo) = qw(d, d), qw(d, d), qw(d, d), qw(d, d)
    (b1, W1), (b3, W3), (b2, W2) = qw(ff, d), qw(ff, d), qw(d, ff)
    g1 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    g2 = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    x = rng.standard_normal((T, d)).astype(np.float32)
    out = np.zeros_like(x)
    args = []
    for b in (bq, bk, bv, bo, b1, b2, b3):
        args += [nso.ptr(b), b.size]
    assert ne.neref_decoder_layer(nso.ptr(x), nso.ptr(out), T, d, heads, ff, eps, base, nso.ptr(g1), nso.ptr(g2), *args) == 0

    def rms(v, g):
        return v / np.sqrt((v * v).mean(-1, keepdims=True) + eps) * g

    def rope(v):   # [T][heads][hs], mode 0 (adjacent pairs), positions 0..T-1
        o = v.copy()
        ts = base ** (-2.0 / hs)
        for i in range(T):
            th = i * ts ** np.arange(hs // 2)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = v[i, :, 0::2], v[i, :, 1::2]
            o[i, :, 0::2] = x0 * c - x1 * s_
            o[i, :, 1::2] = x0 * s_ + x1 * c
        return o
    xd = x.astype(np.float64)
    h = rms(xd, g1)
    q = rope((h @ Wq).reshape(T, heads, hs))
    k = rope((h @ Wk).reshape(T, heads, hs)).astype(np.float16).astype(np.float64)   # kv tensors are fp16
    v = (h @ Wv).reshape(T, heads, hs).astype(np.float16).astype(np.float64)
    att = np.zeros((T, heads, hs))
    for hd in range(heads):
        sc = (q[:, hd] @ k[:, hd].T) / np.sqrt(hs)
        sc = np.where(np.tril(np.ones((T, T), bool)), sc, -np.inf)
        pr = np.exp(sc - sc.max(-1, keepdims=True))
        att[:, hd] = (pr / pr.sum(-1, keepdims=True)) @ v[:, hd]
    r1 = xd + att.reshape(T, d) @ Wo
    h2 = rms(r1, g2)
    gate = h2 @ W1
    ref = r1 + (gate / (1 + np.exp(-gate)) * (h2 @ W3)) @ W2
    err = nso.rel_l2(out, ref)
    assert err < 3e-3, err
    return err


def gpt2_small_greedy(ne, n_new=4):
    """BASELINE config 1 (plumbing): a GPT-2-small-SHAPED decoder (12 layers, d 768, 12 heads, d_ff 3072, vocab 50257 —
    neural-speed has no GPT-2 graph, SURVEY.md section 8d restates the case on the Llama-style layer it does have), int4
    sym g32 weights, greedy decode, every layer ONE reference graph (neref_decoder_layer), lm_head through ne_mul_mat.
    Pass = the same token ids as an independent fp64 model of the network."""
    rng = np.random.default_rng(1234)
    L_, d, heads, ff, V, eps, base = 12, 768, 12, 3072, 50257, 1e-5, 10000.0
    hs = d // heads

    def qw(n, k, std):
        w = (rng.standard_normal((n, k)) * std).astype(np.float32)
        b = nso.quant_pack(w, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)
        return b, nso.unpack_fp32(b).astype(np.float64)
    layers = []
    for _ in range(L_):
        lw = dict(q=qw(d, d, d ** -0.5), k=qw(d, d, d ** -0.5), v=qw(d, d, d ** -0.5), o=qw(d, d, 0.5 * d ** -0.5),
                  w1=qw(ff, d, d ** -0.5), w3=qw(ff, d, d ** -0.5), w2=qw(d, ff, 0.5 * ff ** -0.5),
                  g1=(1 + 0.1 * rng.standard_normal(d)).astype(np.float32), g2=(1 + 0.1 * rng.standard_normal(d)).astype(np.float32))
        layers.append(lw)
    emb = (rng.standard_normal((V, d)) * 0.5).astype(np.float32)
    gf = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    head_b, head_W = qw(V, d, d ** -0.5)

    def rms(v, g):
        return v / np.sqrt((v * v).mean(-1, keepdims=True) + eps) * g

    def rope(v):
        T = v.shape[0]
        o = v.copy()
        ts = base ** (-2.0 / hs)
        for i in range(T):
            th = i * ts ** np.arange(hs // 2)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = v[i, :, 0::2], v[i, :, 1::2]
            o[i, :, 0::2] = x0 * c - x1 * s_
            o[i, :, 1::2] = x0 * s_ + x1 * c
        return o

    def model_fp64(tokens):
        T = len(tokens)
        x = emb[tokens].astype(np.float64)
        for lw in layers:
            h = rms(x, lw["g1"])
            q = rope((h @ lw["q"][1]).reshape(T, heads, hs))
            k = rope((h @ lw["k"][1]).reshape(T, heads, hs)).astype(np.float16).astype(np.float64)
            v = (h @ lw["v"][1]).reshape(T, heads, hs).astype(np.float16).astype(np.float64)
            att = np.zeros((T, heads, hs))
            for hd in range(heads):
                sc = (q[:, hd] @ k[:, hd].T) / np.sqrt(hs)
                sc = np.where(np.tril(np.ones((T, T), bool)), sc, -np.inf)
                pr = np.exp(sc - sc.max(-1, keepdims=True))
                att[:, hd] = (pr / pr.sum(-1, keepdims=True)) @ v[:, hd]
            r1 = x + att.reshape(T, d) @ lw["o"][1]
            h2 = rms(r1, lw["g2"])
            gate = h2 @ lw["w1"][1]
            x = r1 + (gate / (1 + np.exp(-gate)) * (h2 @ lw["w3"][1])) @ lw["w2"][1]
        return rms(x[-1:], gf) @ head_W

    def model_graph(tokens):
        T = len(tokens)
        x = np.ascontiguousarray(emb[tokens])
        for lw in layers:
            out = np.zeros_like(x)
            args = []
            for key in ("q", "k", "v", "o", "w1", "w2", "w3"):
                args += [nso.ptr(lw[key][0]), lw[key][0].size]
            assert ne.neref_decoder_layer(nso.ptr(x), nso.ptr(out), T, d, heads, ff, eps, base, nso.ptr(lw["g1"]), nso.ptr(lw["g2"]), *args) == 0
            x = out
        last = np.ascontiguousarray((rms(x[-1:].astype(np.float64), gf)).astype(np.float32))
        logits = np.zeros((1, V), np.float32)
        assert ne.neref_mul_mat(nso.ptr(last), nso.ptr(head_b), head_b.size, nso.ptr(logits), 1, V, d) == 0
        return logits

    prompt = [464, 3290, 318, 257]
    t_ref, t_graph = list(prompt), list(prompt)
    margins = []
    for _ in range(n_new):
        lr, lg = model_fp64(t_ref)[0], model_graph(t_graph)[0]
        top2 = np.sort(lr)[-2:]
        margins.append(float(top2[1] - top2[0]))
        t_ref.append(int(np.argmax(lr)))
        t_graph.append(int(np.argmax(lg)))
        assert nso.rel_l2(lg[None], lr[None]) < 5e-3
    assert t_graph == t_ref, (t_graph, t_ref)
    return t_graph[len(prompt):], margins


def expert_nodes_case(rng, d, ff):
    """expert-indexed nodes (MoE): ne_mul_mat_id groups the token rows per expert and issues one forward per row
    (ne_layers.c:7783-7916); ne_mul_id_ffn_silu / _gelu run the fused FFN of the expert the first row selects (:8053-8170)"""
    def gelu(x):
        return 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))
    n_as, n_ids, me = 4, 2, 5
    experts = blobs(rng, [(ff, d)] * n_as)
    ids = rng.integers(0, n_as, size=(me, n_ids)).astype(np.int32)
    am = rng.standard_normal((me, d)).astype(np.float32)
    for id_ in range(n_ids):
        y = nso.neref_mul_mat_id(am, experts, ids, id_)
        ref = np.stack([nso.gemm_f64(am[t:t + 1], experts[ids[t, id_]])[0] for t in range(me)])
        assert nso.rel_l2(y, ref) < 1e-3, ("mul_mat_id", id_)
    gate, down, up = blobs(rng, [(ff, d)] * n_as), blobs(rng, [(d, ff)] * n_as), blobs(rng, [(ff, d)] * n_as)
    a1 = am[:1].copy()
    for use_gelu in (False, True):
        y = nso.neref_ffn_id(a1, gate, down, up, ids[:1], 1, gelu=use_gelu)
        e = ids[0, 1]
        g_ = nso.gemm_f64(a1, gate[e])
        act = gelu(g_) if use_gelu else g_ / (1 + np.exp(-g_))
        ref = nso.gemm_f64((act * nso.gemm_f64(a1, up[e])).astype(np.float32), down[e])
        assert nso.rel_l2(y, ref) < 2e-3, ("ffn_id", use_gelu)


def main(kind):
    rng = np.random.default_rng(21)
    m, d, ff = 3, 256, 512
    if kind == "oracle":
        # the CPU oracle answers the bestla_* calls: multi-node reference graphs can be checked without a GPU
        so = os.path.join(tempfile.mkdtemp(), "liboracle_bestla.so")
        nso.build()
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "tools", "oracle_bestla_provider.c"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lns_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
        ne = nso.neref(so)
        assert ne is not None, "oracle/_ref/libne_ref.so missing"
        expert_nodes_case(rng, d, ff)
        print("decoder layer through the reference graph, oracle provider: rel l2 %.2e" % decoder_layer_case(ne, rng))
        if os.environ.get("NS_REF_GRAPH_GPT2", "1") != "0":
            toks, margins = gpt2_small_greedy(ne)
            print("config 1 (GPT-2-small-shaped greedy decode through the reference graph): tokens", toks,
                  "top-1 margins %s" % ["%.3f" % m for m in margins])
        print("REF_GRAPH_ORACLE_OK")
        return
    if kind == "mock":
        so = os.path.join(tempfile.mkdtemp(), "libmock_bestla.so")
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "tools", "mock_bestla_provider.c")])
        provider = so
    else:
        provider = os.path.join(ROOT, "neural-speed_amd", "libns_hip.so")
        import torch  # noqa: F401  (torch's HIP runtime first, as neural_speed_amd.lib() does)
    ne = nso.neref(provider)
    assert ne is not None, "oracle/_ref/libne_ref.so missing"
    (bw,) = blobs(rng, [(ff, d)])
    a = rng.standard_normal((m, d)).astype(np.float32)
    c = np.zeros((m, ff), np.float32)
    assert ne.neref_mul_mat(nso.ptr(a), nso.ptr(bw), bw.size, nso.ptr(c), m, ff, d) == 0
    b1, b2, b3 = blobs(rng, [(ff, d), (d, ff), (ff, d)])
    out = np.zeros((m, d), np.float32)
    if kind == "mock":
        mock = C.CDLL(provider)
        mock.mock_last_call.restype = C.POINTER(MockCall)
        lc = mock.mock_last_call().contents
        # ne_compute_forward_mul_mat_q_f32_bestla (ne_layers.c:7312): (src1 data, BTLA blob, dst, ne1, ne0, ne10, nb11/4, nb1/4, wdata)
        assert (lc.which, lc.m, lc.n, lc.k, lc.lda, lc.ldo) == (1, m, ff, d, d, ff), (lc.which, lc.m, lc.n, lc.k, lc.lda, lc.ldo)
        assert lc.w == bw.ctypes.data and lc.ws          # the weight pointer is the caller's blob; a workspace was sized
        assert np.array_equal(c, a[:, :1] + np.arange(ff, dtype=np.float32)[None, :])
        assert ne.neref_ffn_silu(nso.ptr(a), nso.ptr(b1), b1.size, nso.ptr(b2), b2.size, nso.ptr(b3), b3.size, nso.ptr(out), m, d, ff) == 0
        lc = mock.mock_last_call().contents
        assert (lc.which, lc.seq, lc.fin, lc.fmid, lc.fout) == (2, m, d, ff, d)
        assert (lc.w1, lc.w2, lc.w3) == (b1.ctypes.data, b2.ctypes.data, b3.ctypes.data)
        assert np.all(out == 42.0)
        bq, bk, bv = blobs(rng, [(ff, d)] * 3)
        qkv = np.zeros((3, m, ff), np.float32)
        assert ne.neref_mul_qkv(nso.ptr(a), nso.ptr(bq), bq.size, nso.ptr(bk), bk.size, nso.ptr(bv), bv.size, nso.ptr(qkv), m, ff, d) == 0
        lc = mock.mock_last_call().contents
        # ne_compute_forward_mul_qkv (ne_layers.c:8050): (src, qw, kw, vw, dst, m, n, k, lda = k, ldo = n, wdata)
        assert (lc.which, lc.m, lc.n, lc.k, lc.lda, lc.ldo) == (3, m, ff, d, d, ff)
        assert (lc.w1, lc.w2, lc.w3) == (bq.ctypes.data, bk.ctypes.data, bv.ctypes.data) and np.all(qkv == 7.0)
        # fused-attention node: the tensors' nb[] strides become the args struct of include/ns_bestla.h
        class AttnArgsC(C.Structure):
            _fields_ = ([("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("dst", C.c_void_p)] +
                        [(n_, C.c_float) for n_ in ("Q_sc", "K_sc", "V_sc", "dst_sc")] + [("tmp", C.c_void_p), ("QK_scale", C.c_float),
                         ("attn_flags", C.c_uint32)] +
                        [(n_, C.c_int) for n_ in ("batch_size", "head_num", "heads_kv", "head_size", "sl_q", "sl_kv", "Q_layout",
                                                  "K_layout", "V_layout", "dst_layout", "step_q_bs", "step_q_head_num", "step_q_sl",
                                                  "step_k_bs", "step_k_head_num", "step_k_sl", "step_k_head_size", "step_v_bs",
                                                  "step_v_head_num", "step_v_sl", "step_v_head_size", "step_dst_bs",
                                                  "step_dst_head_num", "step_dst_sl")])
        bs_, hn, hkv, hs, slq, slkv = 2, 8, 2, 64, 3, 11
        qa = rng.standard_normal((bs_, slq, hn, hs)).astype(np.float32)
        ka = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float16)
        va = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float16)
        o = nso.neref_flash_attn(qa, ka, va, 0.125, 1)
        mock.mock_last_attn.restype = C.POINTER(AttnArgsC)
        g = mock.mock_last_attn().contents
        assert (g.batch_size, g.head_num, g.heads_kv, g.head_size, g.sl_q, g.sl_kv) == (bs_, hn, hkv, hs, slq, slkv)
        assert (g.step_q_bs, g.step_q_head_num, g.step_q_sl) == (slq * hn * hs, hs, hn * hs)
        assert (g.step_k_bs, g.step_k_head_num, g.step_k_sl, g.step_k_head_size) == (slkv * hkv * hs, hs, hkv * hs, 1)
        assert (g.step_v_bs, g.step_v_head_num, g.step_v_sl, g.step_v_head_size) == (slkv * hkv * hs, hs, hkv * hs, 1)
        assert (g.step_dst_bs, g.step_dst_head_num, g.step_dst_sl) == (slq * hn * hs, hs, hn * hs)
        assert abs(g.QK_scale - 0.125) < 1e-9 and g.attn_flags == 1 and g.tmp and np.all(o == 3.0)
        print("REF_GRAPH_MOCK_OK")
        return
    # the real product behind the reference's graph
    assert nso.rel_l2(c, nso.gemm_f64(a, bw)) < 1e-3
    assert ne.neref_ffn_silu(nso.ptr(a), nso.ptr(b1), b1.size, nso.ptr(b2), b2.size, nso.ptr(b3), b3.size, nso.ptr(out), m, d, ff) == 0
    g = nso.gemm_f64(a, b1)
    h = (g / (1 + np.exp(-g)) * nso.gemm_f64(a, b3)).astype(np.float32)
    assert nso.rel_l2(out, nso.gemm_f64(h, b2)) < 2e-3
    # decode-sized call as the model issues it: one token row
    c1 = np.zeros((1, ff), np.float32)
    assert ne.neref_mul_mat(nso.ptr(a[:1].copy()), nso.ptr(bw), bw.size, nso.ptr(c1), 1, ff, d) == 0
    assert nso.rel_l2(c1, nso.gemm_f64(a[:1], bw)) < 1e-3
    # fused QKV node: three products stacked along dim 0 (ip_fusion_qkv.cpp:84-86)
    bq, bk, bv = blobs(rng, [(ff, d)] * 3)
    qkv = np.zeros((3, m, ff), np.float32)
    assert ne.neref_mul_qkv(nso.ptr(a), nso.ptr(bq), bq.size, nso.ptr(bk), bk.size, nso.ptr(bv), bv.size, nso.ptr(qkv), m, ff, d) == 0
    for i, b in enumerate((bq, bk, bv)):
        assert nso.rel_l2(qkv[i], nso.gemm_f64(a, b)) < 1e-3
    # the other fused nodes of the path: bias add, FFN GeLU / Add_GeLU / GeLU_Mul
    def gelu(x):
        return 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))

    def fused(kind_, w1, w2=None, w3=None, bias1=None, bias2=None, n_out=d):
        o = np.zeros((m, n_out), np.float32)
        sz = lambda b: (nso.ptr(b), b.size) if b is not None else (None, 0)
        assert ne.neref_fused(kind_, nso.ptr(a), *sz(w1), *sz(w2), *sz(w3), nso.ptr(bias1), nso.ptr(bias2), nso.ptr(o), m, d, ff) == 0
        return o
    bi1 = rng.standard_normal(ff).astype(np.float32)
    bi2 = rng.standard_normal(d).astype(np.float32)
    g1 = nso.gemm_f64(a, b1)
    assert nso.rel_l2(fused(0, b1, bias1=bi1, n_out=ff), g1 + bi1) < 1e-3
    assert nso.rel_l2(fused(1, b1, b2), nso.gemm_f64(gelu(g1).astype(np.float32), b2)) < 2e-3
    assert nso.rel_l2(fused(2, b1, b2, bias1=bi1, bias2=bi2), nso.gemm_f64(gelu(g1 + bi1).astype(np.float32), b2) + bi2) < 2e-3
    assert nso.rel_l2(fused(3, b1, b2, b3), nso.gemm_f64((gelu(g1) * nso.gemm_f64(a, b3)).astype(np.float32), b2)) < 2e-3
    # fused-attention node (ne_flash_attn): the reference marshals tensor strides, the product's kernel answers; GQA,
    # causal with sl_q < sl_kv, batch 2.  (The reference reads the flags back as a bool, ne_layers.c:10168: causal only.)
    bs_, hn, hkv, hs, slq, slkv = 2, 8, 2, 64, 3, 11
    qa = rng.standard_normal((bs_, slq, hn, hs)).astype(np.float32)
    ka = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float16)
    va = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float16)
    for flags in (1, 0):
        o = nso.neref_flash_attn(qa, ka, va, hs ** -0.5, flags)
        assert nso.rel_l2(o, nso.attn_ref(qa, ka, va, hs ** -0.5, flags)) < 1e-3
    # library-managed kv cache (llama.cpp:496-571): NE_TYPE_BTLA cache tensors sized by batch_kv_info, fp32 K / V appended by
    # the update nodes (one call, or past rows then current rows at seq_off > 0), attention over views with the info strides
    n_ctx = 16
    kf = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float32)
    vf = rng.standard_normal((bs_, slkv, hkv, hs)).astype(np.float32)
    want = nso.attn_ref(qa, kf.astype(np.float16), vf.astype(np.float16), hs ** -0.5, 1)
    for split in (False, True):
        o = nso.neref_reordered_attn(qa, kf, vf, n_ctx, hs ** -0.5, 1, split=split)
        assert nso.rel_l2(o, want) < 1e-3, (split, nso.rel_l2(o, want))
    # ne_rms_norm / ne_norm nodes: their forwards call bestla_layernormalization unconditionally (ne_layers.c:4622)
    x = rng.standard_normal((5, 300)).astype(np.float32)
    xd = x.astype(np.float64)
    assert nso.rel_l2(nso.neref_norm(x, 1e-6, True), xd / np.sqrt((xd ** 2).mean(-1, keepdims=True) + 1e-6)) < 1e-6
    assert nso.rel_l2(nso.neref_norm(x, 1e-5, False), (xd - xd.mean(-1, keepdims=True)) / np.sqrt(xd.var(-1, keepdims=True) + 1e-5)) < 1e-5
    expert_nodes_case(rng, d, ff)
    print("decoder layer through the reference graph on libns_hip.so: rel l2 %.2e" % decoder_layer_case(ne, rng))
    if os.environ.get("NS_REF_GRAPH_GPT2_PRODUCT", "1") != "0":
        toks, margins = gpt2_small_greedy(ne)
        print("config 1 on libns_hip.so: tokens", toks, "top-1 margins %s" % ["%.3f" % m_ for m_ in margins])
    print("REF_GRAPH_PRODUCT_OK")


if __name__ == "__main__":
    main(sys.argv[1])
