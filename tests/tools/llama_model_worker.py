"""Worker of tests/test_llama_model.py / tests/test_gpu_llama_model.py (fresh interpreter: the bestla_* provider must be
loaded RTLD_GLOBAL before oracle/_ref/libne_llama_ref.so = the reference's UNCHANGED llama model code + graph executor,
see oracle/llama_ref_harness.cpp).

  llama_model_worker.py oracle  <workdir> <kv: auto|f16|f32> <heads_kv> [existing quantized file | -] [family: llama | gptj]
      the CPU oracle answers the bestla_* calls; the quantized NE file is written here (nso.quant_pack blobs) unless one
      is given (the GPU test hands over the file the product run produced)
  llama_model_worker.py product <workdir> <kv> <heads_kv> [-] [family]
      libns_hip.so answers them (GPU): an fp32 NE file goes through the reference's quantizer driver
      (model_quantize -> bestla_quantize -> BTLAGemmQuantPackB -> glue/bestla_gemm_hip.cpp -> ns_BTLAGemmQuantPackB), the
      resulting file's blobs must equal the oracle's byte for byte, then the reference's loader + llama graph generate
  llama_model_worker.py device  <workdir> f32 <heads_kv> <quantized file>
      libns_hip.so again, but the reference built with ITS device switch (-DNS_SYCL -> oracle/_ref/libne_llama_dev_ref.so):
      every layer offloaded, BTLA weights through bestla_device_load_storage, fp32 device kv cache, the device branch of
      the graph builder — the unchanged graph runs device-resident (glue/ne_bestla_hip_device.c, csrc/ns_device.hip)
All: greedy generation, token ids and logits compared with an independent fp64 model of the network built from the
dequantized weights; results saved to <workdir>/<mode>_<kv>_<heads_kv>.npz for the cross-provider comparison."""
import ctypes as C
import os

# the CPU oracle's GEMMs are OpenMP loops over small matrices: a team as wide as a GPU host (hundreds of hardware threads,
# possibly behind a CPU quota) spends its time in barriers — 40 s for one prompt step on the GPU box vs 1 s with 8 threads
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import ne_file  # noqa: E402
import nso  # noqa: E402

V, D, HEADS, FF, LAYERS, EPS, BASE = 384, 256, 4, 704, 22, 1e-5, 10000.0
N_CTX = int(os.environ.get("NS_WORKER_N_CTX", "64"))   # (long generations: test_gpu_llama_model.py's 330-token run)
PROMPT = [1, 17, 200, 3, 99, 42, 311]   # bos first (llama.cpp:80-85 warns otherwise)
if int(os.environ.get("NS_WORKER_PROMPT_LEN", "0")) > len(PROMPT):   # a prompt-SIZED prompt (more than 16 rows: the tiled GEMMs and, on the device route, the window's prefill forms)
    PROMPT = PROMPT + [int(t_) for t_ in np.random.default_rng(11).integers(3, V, int(os.environ["NS_WORKER_PROMPT_LEN"]) - len(PROMPT))]
N_NEW = int(os.environ.get("NS_WORKER_N_NEW", "6"))
KV = {"auto": 0, "f16": 1, "f32": 2}


def make_model(heads_kv, n_experts=0):
    """n_experts > 0: a Mixtral-style mixture of experts (router ffn_gate_inp + n_experts SiLU FFNs per layer, two used per
    token; llama_utils.cpp:141-152).  Such a model loads through the GGUF branch only: the NE branch of Llama::load sums
    ne_nbytes(layer.ffn[0..2]), null for it (llama_utils.cpp:228-230)."""
    rng = np.random.default_rng(77)
    hs = D // HEADS
    dkv = hs * heads_kv
    ff = 256 if n_experts else FF   # eight experts per layer: keep the file small
    t = [("tok_embeddings.weight", (rng.standard_normal((V, D)) * 0.5).astype(np.float32)),
         ("norm.weight", (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)),
         ("output.weight", (rng.standard_normal((V, D)) * D ** -0.5).astype(np.float32))]
    for i in range(LAYERS):
        p = "layers.%d." % i
        t += [(p + "attention_norm.weight", (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)),
              (p + "attention.wq.weight", (rng.standard_normal((D, D)) * D ** -0.5).astype(np.float32)),
              (p + "attention.wk.weight", (rng.standard_normal((dkv, D)) * D ** -0.5).astype(np.float32)),
              (p + "attention.wv.weight", (rng.standard_normal((dkv, D)) * D ** -0.5).astype(np.float32)),
              (p + "attention.wo.weight", (rng.standard_normal((D, D)) * 0.5 * D ** -0.5).astype(np.float32)),
              (p + "ffn_norm.weight", (1 + 0.1 * rng.standard_normal(D)).astype(np.float32))]
        if n_experts:
            # a wide router (logit gaps of order one: the choice of experts must not hang on the last bits of a softmax).  Eight
            # experts: with fewer the reference's quantizer driver overruns its own output buffer on the router weight (it sizes
            # the buffer as 4 bytes per element, quant_utils.cpp:411-412; a blob pads N to the 48-column tile: 8000 bytes here)
            t += [(p + "ffn_gate_inp.weight", (rng.standard_normal((n_experts, D)) * 4.0 * D ** -0.5).astype(np.float32))]
            for x in range(n_experts):
                t += [(p + "ffn_gate.%d.weight" % x, (rng.standard_normal((ff, D)) * D ** -0.5).astype(np.float32)),
                      (p + "ffn_down.%d.weight" % x, (rng.standard_normal((D, ff)) * 0.5 * ff ** -0.5).astype(np.float32)),
                      (p + "ffn_up.%d.weight" % x, (rng.standard_normal((ff, D)) * D ** -0.5).astype(np.float32))]
        else:
            t += [(p + "feed_forward.w1.weight", (rng.standard_normal((FF, D)) * D ** -0.5).astype(np.float32)),
                  (p + "feed_forward.w2.weight", (rng.standard_normal((D, FF)) * 0.5 * FF ** -0.5).astype(np.float32)),
                  (p + "feed_forward.w3.weight", (rng.standard_normal((FF, D)) * D ** -0.5).astype(np.float32))]
    hp = dict(n_vocab=V, n_embd=D, n_mult=256, n_head=HEADS, n_head_kv=heads_kv, n_layer=LAYERS, n_rot=hs, ftype=0,
              max_seq_len=N_CTX, ffn_hidden_size=ff, norm_eps=EPS, freq_base=BASE, freq_scale=1.0, rope_scaling_factor=0.0,
              n_experts=n_experts, n_experts_used=2 if n_experts else 0)
    return hp, t


GJ_LAYERS, GJ_ROT, GJ_FF = 28, 32, 4 * D   # gptj_mem_req knows 28 layers only (gptj.h:29-41); n_ff = 4 n_embd (gptj_utils.cpp:66)


def make_model_gptj():
    """GPT-J: LayerNorm with bias, rotary on the first n_rot dims of every head, attention and FFN both fed by ln_1(x) and
    added to x (parallel residual), gelu(x W_in + b_in) W_out + b_out, lm_head with bias (tensor names of gptj_utils.cpp:113-144)"""
    rng = np.random.default_rng(78)
    vec = lambda n, s=0.1, m=0.0: (m + s * rng.standard_normal(n)).astype(np.float32)
    t = [("transformer.wte.weight", (rng.standard_normal((V, D)) * 0.5).astype(np.float32)),
         ("transformer.ln_f.weight", vec(D, 0.1, 1.0)), ("transformer.ln_f.bias", vec(D)),
         ("lm_head.weight", (rng.standard_normal((V, D)) * D ** -0.5).astype(np.float32)), ("lm_head.bias", vec(V))]
    for i in range(GJ_LAYERS):
        p = "transformer.h.%d." % i
        t += [(p + "ln_1.weight", vec(D, 0.1, 1.0)), (p + "ln_1.bias", vec(D)),
              (p + "attn.q_proj.weight", (rng.standard_normal((D, D)) * D ** -0.5).astype(np.float32)),
              (p + "attn.k_proj.weight", (rng.standard_normal((D, D)) * D ** -0.5).astype(np.float32)),
              (p + "attn.v_proj.weight", (rng.standard_normal((D, D)) * D ** -0.5).astype(np.float32)),
              (p + "attn.out_proj.weight", (rng.standard_normal((D, D)) * 0.4 * D ** -0.5).astype(np.float32)),
              (p + "mlp.fc_in.weight", (rng.standard_normal((GJ_FF, D)) * D ** -0.5).astype(np.float32)), (p + "mlp.fc_in.bias", vec(GJ_FF)),
              (p + "mlp.fc_out.weight", (rng.standard_normal((D, GJ_FF)) * 0.4 * GJ_FF ** -0.5).astype(np.float32)),
              (p + "mlp.fc_out.bias", vec(D))]
    hp = dict(n_vocab=V, n_embd=D, n_mult=256, n_head=HEADS, n_head_kv=HEADS, n_layer=GJ_LAYERS, n_rot=GJ_ROT, ftype=0,
              max_seq_len=N_CTX, norm_eps=EPS, freq_base=BASE, freq_scale=1.0, rope_scaling_factor=0.0)
    return hp, t


def model_fp64_gptj(deq, tokens, kv_fp16):
    hs, T = D // HEADS, len(tokens)

    def ln(v, g, b):
        mu = v.mean(-1, keepdims=True)
        return (v - mu) / np.sqrt(((v - mu) ** 2).mean(-1, keepdims=True) + EPS) * g + b

    def rope(v):   # [T][h][hs]: adjacent pairs of the first GJ_ROT dims only, positions 0..T-1
        o = v.copy()
        ts = BASE ** (-2.0 / GJ_ROT)
        for i in range(T):
            th = i * ts ** np.arange(GJ_ROT // 2)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = v[i, :, 0:GJ_ROT:2], v[i, :, 1:GJ_ROT:2]
            o[i, :, 0:GJ_ROT:2] = x0 * c - x1 * s_
            o[i, :, 1:GJ_ROT:2] = x0 * s_ + x1 * c
        return o
    gelu = lambda x: 0.5 * x * (1 + np.tanh(0.7978845834732056 * (x + 0.044714998453855515 * x ** 3)))
    x = deq["transformer.wte.weight"][tokens]
    for i in range(GJ_LAYERS):
        p = "transformer.h.%d." % i
        h = ln(x, deq[p + "ln_1.weight"], deq[p + "ln_1.bias"])
        q = rope((h @ deq[p + "attn.q_proj.weight"]).reshape(T, HEADS, hs))
        k = rope((h @ deq[p + "attn.k_proj.weight"]).reshape(T, HEADS, hs))
        v = (h @ deq[p + "attn.v_proj.weight"]).reshape(T, HEADS, hs)
        if kv_fp16:
            k, v = k.astype(np.float16).astype(np.float64), v.astype(np.float16).astype(np.float64)
        att = np.zeros((T, HEADS, hs))
        for hd in range(HEADS):
            sc = (q[:, hd] @ k[:, hd].T) / np.sqrt(hs)
            sc = np.where(np.tril(np.ones((T, T), bool)), sc, -np.inf)
            pr = np.exp(sc - sc.max(-1, keepdims=True))
            att[:, hd] = (pr / pr.sum(-1, keepdims=True)) @ v[:, hd]
        ffn = gelu(h @ deq[p + "mlp.fc_in.weight"] + deq[p + "mlp.fc_in.bias"]) @ deq[p + "mlp.fc_out.weight"] + deq[p + "mlp.fc_out.bias"]
        x = x + att.reshape(T, D) @ deq[p + "attn.out_proj.weight"] + ffn
    return ln(x[-1:], deq["transformer.ln_f.weight"], deq["transformer.ln_f.bias"]) @ deq["lm_head.weight"] + deq["lm_head.bias"]


def quantized(name, a):
    """the llama quant-layer rule (llama_utils.cpp:259-295): 2-D '*weight' tensors except the token embedding"""
    return a.ndim == 2 and name.endswith("weight") and name not in ("tok_embeddings.weight", "transformer.wte.weight")


def quantize_tensors(tensors):
    out = []
    for name, a in tensors:
        if quantized(name, a):
            blob = nso.quant_pack(a, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB)  # int4 sym g32, bf16 scales, int8 compute
            out.append((name, (blob, a.shape[0], a.shape[1])))
        else:
            out.append((name, a))
    return out


def same_blob(data, a):
    """the blob of the file equals the oracle's in every byte a packer WRITES.  The alignment gaps between sections and the
    tail of the reduce section are written by neither side (bestla_storage.h:85-109), and the quantizer driver packs into an
    uninitialised buffer (quant_utils.cpp:411-413), so those hold whatever the allocator left: the written bytes are the
    ones two oracle packs over differently pre-filled buffers agree on."""
    b0 = nso.quant_pack(a, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB, fill=0x00)
    b1 = nso.quant_pack(a, 32, nso.S4, nso.BF16, False, nso.CORE_AVX512_VNNI_KB, fill=0xFF)
    got = np.frombuffer(data, np.uint8)
    if got.size != b0.size:
        return "size %d != %d" % (got.size, b0.size)
    written = b0 == b1
    bad = written & (got != b0)
    return None if not bad.any() else "%d of %d written bytes differ, first at %d" % (int(bad.sum()), int(written.sum()), int(np.argmax(bad)))


def weights_from_file(path):
    """fp64 weights of the model the reference's loader will see: 2-D GEMM weights as [K][N], the embedding as [V][D]"""
    _, tensors = ne_file.read(path)
    if "token_embd.weight" in tensors:   # a file quantized from GGUF keeps the GGUF tensor names: back to the converter's
        import gguf_file
        tensors = {gguf_file.ne_name(k_): v_ for k_, v_ in tensors.items()}
    deq = {}
    for name, (typ, ne, data) in tensors.items():
        if typ == ne_file.NE_TYPE_BTLA:
            blob = nso.aligned_bytes(len(data))
            blob[:] = np.frombuffer(data, np.uint8)
            deq[name] = nso.unpack_fp32(blob).astype(np.float64)
        elif typ == ne_file.NE_TYPE_Q4_0:
            deq[name] = ne_file.dequant_q4_0(data, ne)
        else:
            deq[name] = np.frombuffer(data, np.float32).reshape(tuple(reversed(ne))).astype(np.float64)
    return deq


def model_fp64(deq, heads_kv, tokens, kv_fp16, n_experts=0, gaps=None, all_positions=False):
    """logits of the LAST position (all_positions: of every position — the model is causal, row t is what it says after tokens[: t + 1]).
    gaps (list): receives, per layer and position, the relative gap between the second and
    the third router probability (how safely the two experts were chosen)"""
    hs, T, grp = D // HEADS, len(tokens), HEADS // heads_kv

    def rms(v, g):
        return v / np.sqrt((v * v).mean(-1, keepdims=True) + EPS) * g

    def rope(v):   # [T][h][hs], mode 0: adjacent pairs, positions 0..T-1
        o = v.copy()
        ts = BASE ** (-2.0 / hs)
        for i in range(T):
            th = i * ts ** np.arange(hs // 2)
            c, s_ = np.cos(th), np.sin(th)
            x0, x1 = v[i, :, 0::2], v[i, :, 1::2]
            o[i, :, 0::2] = x0 * c - x1 * s_
            o[i, :, 1::2] = x0 * s_ + x1 * c
        return o
    x = deq["tok_embeddings.weight"][tokens]
    for i in range(LAYERS):
        p = "layers.%d." % i
        h = rms(x, deq[p + "attention_norm.weight"])
        q = rope((h @ deq[p + "attention.wq.weight"]).reshape(T, HEADS, hs))
        k = rope((h @ deq[p + "attention.wk.weight"]).reshape(T, heads_kv, hs))
        v = (h @ deq[p + "attention.wv.weight"]).reshape(T, heads_kv, hs)
        if kv_fp16:
            k, v = k.astype(np.float16).astype(np.float64), v.astype(np.float16).astype(np.float64)
        att = np.zeros((T, HEADS, hs))
        for hd in range(HEADS):
            sc = (q[:, hd] @ k[:, hd // grp].T) / np.sqrt(hs)
            sc = np.where(np.tril(np.ones((T, T), bool)), sc, -np.inf)
            pr = np.exp(sc - sc.max(-1, keepdims=True))
            att[:, hd] = (pr / pr.sum(-1, keepdims=True)) @ v[:, hd // grp]
        x = x + att.reshape(T, D) @ deq[p + "attention.wo.weight"]
        h2 = rms(x, deq[p + "ffn_norm.weight"])
        if n_experts:   # llama.cpp:619-683: softmax over the router logits, top 2, weights renormalised
            lg = h2 @ deq[p + "ffn_gate_inp.weight"]
            pr = np.exp(lg - lg.max(-1, keepdims=True))
            pr /= pr.sum(-1, keepdims=True)
            order = np.argsort(-pr, axis=-1)
            moe = np.zeros_like(x)
            for ti in range(T):
                sel = order[ti, :2]
                wsel = pr[ti, sel] / pr[ti, sel].sum()
                if gaps is not None and n_experts > 2:
                    gaps.append(float((pr[ti, order[ti, 1]] - pr[ti, order[ti, 2]]) / pr[ti, order[ti, 1]]))
                for e_, w_ in zip(sel, wsel):
                    g = h2[ti] @ deq[p + "ffn_gate.%d.weight" % e_]
                    moe[ti] += w_ * ((g / (1 + np.exp(-g)) * (h2[ti] @ deq[p + "ffn_up.%d.weight" % e_])) @ deq[p + "ffn_down.%d.weight" % e_])
            x = x + moe
        else:
            g = h2 @ deq[p + "feed_forward.w1.weight"]
            x = x + (g / (1 + np.exp(-g)) * (h2 @ deq[p + "feed_forward.w3.weight"])) @ deq[p + "feed_forward.w2.weight"]
    return rms(x if all_positions else x[-1:], deq["norm.weight"]) @ deq["output.weight"]


def device_turns(ref, qpath, heads_kv):
    """A conversation through the reference's unchanged model_eval on the device route (oracle/llama_ref_harness.cpp nellama_generate_dev_turns):
    prompt, 12 tokens; a second chunk of 5 tokens that FOLLOWS the cache, 12 tokens; a third chunk of 9 tokens that starts over at position 0 of the
    same cache, 12 tokens.  Every generated token's logits against the fp64 model of the sequence the cache holds at that point."""
    rng = np.random.default_rng(3)
    chunks = [list(PROMPT), [int(t) for t in rng.integers(3, V, 5)], [1] + [int(t) for t in rng.integers(3, V, 8)]]
    n_new, rewind = [12, 12, 12], [0, 0, 1]
    flat = [t for c in chunks for t in c]
    total = sum(n_new)
    toks = (C.c_int * total)()
    logits = np.zeros((total, V), np.float32)
    ia = lambda v: (C.c_int * len(v))(*v)
    ref.nellama_generate_dev_turns.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = ref.nellama_generate_dev_turns(qpath.encode(), 3, ia(flat), ia([len(c) for c in chunks]), ia(n_new), ia(rewind), N_CTX, LAYERS, toks, logits.ctypes.data)
    assert n == total, n
    toks = list(toks)
    hipl = C.CDLL(os.path.join(ROOT, "neural-speed_amd", "libns_hip.so"))
    rs = (C.c_uint64 * 8)()
    hipl.ns_hip_route_stats(rs)
    print("device route replay: tokens_replayed=%d tokens_eager=%d plans=%d fallbacks=%d launches_per_token=%d captured_launches=%d capture_failures=%d"
          % (rs[0], rs[1], rs[2], rs[3], rs[4], rs[5], rs[6]))
    deq = weights_from_file(qpath)
    kv16_model = os.environ.get("NS_DEVICE_KV", "f16") not in ("f32", "fp32", "0")
    errs, checked = [], 0
    seq, made = [], 0
    for t, chunk in enumerate(chunks):
        if rewind[t]:
            seq = []
        # the cache holds: what it held, this chunk, and the turn's generated tokens but the last (which is never evaluated)
        gen = toks[made:made + n_new[t]]
        full = seq + chunk + gen[:-1]
        want_all = model_fp64(deq, heads_kv, full, kv16_model, all_positions=True)
        for i in range(n_new[t]):
            want = want_all[len(seq) + len(chunk) - 1 + i]
            errs.append(nso.rel_l2(logits[made + i], want))
            top = np.sort(want)[-2:]
            if top[1] - top[0] > 0.05:
                assert gen[i] == int(np.argmax(want)), (t, i, gen[i], int(np.argmax(want)))
                checked += 1
        seq = full
        made += n_new[t]
    print("device route, three turns (follow-up chunk, then a new sequence at position 0 of the same cache): tokens %s, logits rel l2 vs the fp64 model max %.2e, "
          "%d of %d tokens had a clear margin and are the model's" % (toks, max(errs), checked, total))
    assert max(errs) < 1e-2, errs
    print("LLAMA_MODEL_DEVICE_OK")


def main(mode, workdir, kv, heads_kv, given=None, family="llama"):
    heads_kv = int(heads_kv)
    given = None if given in (None, "-") else given
    os.makedirs(workdir, exist_ok=True)
    n_experts = int(os.environ.get("NS_WORKER_EXPERTS", "0")) if family == "llama" else 0
    hp, tensors = make_model(heads_kv, n_experts) if family == "llama" else make_model_gptj()
    qt = quantize_tensors(tensors)
    tag = ("%d%s" % (heads_kv, "_moe%d" % n_experts if n_experts else "")) if family == "llama" else family
    qpath = given or os.path.join(workdir, "%s_q_%s_%s.bin" % (family, mode, tag))
    device = mode == "device"
    if device:
        assert given and kv == "f32" and family == "llama" and not n_experts, "device mode: an existing quantized file, the fp32 cache"
    if mode == "oracle":
        so = os.path.join(tempfile.mkdtemp(), "liboracle_bestla.so")
        nso.build()
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "tools", "oracle_bestla_provider.c"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lns_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
        provider = so
    else:
        import torch  # noqa: F401  (torch's HIP runtime first, as neural_speed_amd.lib() does)
        provider = os.path.join(ROOT, "neural-speed_amd", "libns_hip.so")
    C.CDLL(provider, mode=C.RTLD_GLOBAL)
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libne_llama_dev_ref.so" if device else "libne_%s_ref.so" % family)
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "nellamadev" if device else "nellama"], stdout=subprocess.DEVNULL)
    ref = C.CDLL(lib_path)
    use_gguf = os.environ.get("NS_WORKER_GGUF") == "1"
    if use_gguf and not given:
        # GGUF route: an fp32 GGUF file -> the reference's GGUF reader (model_files.h:643-986) -> its quantizer driver on this
        # provider's quantizer -> an NE-container file with GGUF tensor names, which the llama loader takes by its GGUF branch
        # (llama_utils.cpp:113-152)
        import gguf_file
        gpath = os.path.join(workdir, "%s_f32_%s.gguf" % (family, tag))
        gguf_file.write_llama(gpath, hp, tensors)
        ref.nellama_quantize.argtypes = [C.c_char_p] * 4 + [C.c_int] + [C.c_char_p] * 2
        assert ref.nellama_quantize(gpath.encode(), qpath.encode(), b"int4", b"sym", 32, b"bf16", b"int8") == 0
        _, got = ne_file.read(qpath)
        for name, t in qt:
            typ, ne, data = got[gguf_file.gguf_name(name)]
            if isinstance(t, tuple):
                why = same_blob(data, dict(tensors)[name])
                assert typ == ne_file.NE_TYPE_BTLA and why is None, (name, why)
        print("GGUF route: the reference's GGUF reader + quantizer driver on the %s provider wrote the oracle's blobs" % mode)
    elif mode == "oracle":
        if not given:
            ne_file.write(qpath, dict(hp, ftype=ne_file.NE_FTYPE_MOSTLY_Q_BTLA), qt)
    elif device:
        pass   # the file the product run wrote
    else:
        # the reference's quantizer driver on the product's quantizer
        fpath = os.path.join(workdir, "%s_f32_%s.bin" % (family, tag))
        ne_file.write(fpath, hp, tensors)
        ref.nellama_quantize.argtypes = [C.c_char_p] * 4 + [C.c_int] + [C.c_char_p] * 2
        assert ref.nellama_quantize(fpath.encode(), qpath.encode(), b"int4", b"sym", 32, b"bf16", b"int8") == 0
        hq, got = ne_file.read(qpath)
        # (the saver writes the INPUT file's ftype: write_hparams ignores its new_ftype argument, model_files.h:1248-1257)
        assert len(got) == len(qt), (len(got), len(qt))
        n_blobs = 0
        for name, t in qt:
            typ, ne, data = got[name]
            if isinstance(t, tuple):
                assert typ == ne_file.NE_TYPE_BTLA and ne == (t[2], t[1]), name
                why = same_blob(data, dict(tensors)[name])
                assert why is None, "blob of %s differs from the oracle's: %s" % (name, why)
                n_blobs += 1
            elif name in ("tok_embeddings.weight", "transformer.wte.weight"):
                assert typ == ne_file.NE_TYPE_Q4_0   # the reference's own ggml quantizer (llama_utils.cpp:261-265)
            else:
                assert typ == ne_file.NE_TYPE_F32 and data == np.ascontiguousarray(t, np.float32).tobytes(), name
        print("reference quantizer driver on libns_hip.so: %d BTLA blobs equal to the oracle's in every written byte" % n_blobs)
    if device and os.environ.get("NS_WORKER_TURNS") == "1":
        return device_turns(ref, qpath, heads_kv)
    toks = (C.c_int * N_NEW)()
    logits = np.zeros((N_NEW, V), np.float32)
    prompt = (C.c_int * len(PROMPT))(*PROMPT)
    ref.nellama_generate.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sys.stdout.flush()
    if device:
        us = C.c_double(0)
        ref.nellama_generate_dev.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        n = ref.nellama_generate_dev(qpath.encode(), prompt, len(PROMPT), N_NEW, N_CTX, LAYERS, toks, logits.ctypes.data, C.byref(us))
        print("device-resident graph: %.1f us per single-token eval (%d layers, d %d)" % (us.value, LAYERS, D))
        # ns_route.cpp: how the single-token evals were issued (replayed from the plan / launched one by one)
        hipl = C.CDLL(os.path.join(ROOT, "neural-speed_amd", "libns_hip.so"))
        rs = (C.c_uint64 * 8)()
        hipl.ns_hip_route_stats(rs)
        print("device route replay: tokens_replayed=%d tokens_eager=%d plans=%d fallbacks=%d launches_per_token=%d captured_launches=%d capture_failures=%d"
              % (rs[0], rs[1], rs[2], rs[3], rs[4], rs[5], rs[6]))
    else:
        n = ref.nellama_generate(qpath.encode(), prompt, len(PROMPT), N_NEW, N_CTX, KV[kv], toks, logits.ctypes.data)
    assert n == N_NEW, n
    toks = list(toks)
    deq = weights_from_file(qpath)
    # independent fp64 model: same greedy tokens wherever its own top-1 margin is clear of the path's tolerance
    seq, errs, margins, gaps = list(PROMPT), [], [], []
    # the device route's attention reads an fp16 mirror of its fp32 cache (csrc/ns_route.h) unless NS_DEVICE_KV=f32 keeps the fp32 kernels
    kv16_model = kv != "f32" or (device and os.environ.get("NS_DEVICE_KV", "f16") not in ("f32", "fp32", "0"))
    long_run = family == "llama" and not n_experts and N_NEW > 16
    if long_run:   # one pass over the whole generated sequence instead of one per token (the model is causal)
        want_all = model_fp64(deq, heads_kv, list(PROMPT) + toks[:-1], kv16_model, all_positions=True)
    for i in range(N_NEW):
        want = (want_all[len(PROMPT) - 1 + i] if long_run else
                (model_fp64(deq, heads_kv, seq, kv16_model, n_experts, gaps) if family == "llama" else model_fp64_gptj(deq, seq, kv != "f32"))[0])
        errs.append(nso.rel_l2(logits[i], want))
        top = np.sort(want)[-2:]
        margins.append(float(top[1] - top[0]))
        if margins[-1] > 0.05:
            assert toks[i] == int(np.argmax(want)), (i, toks[i], int(np.argmax(want)), margins[-1])
        seq.append(toks[i])
    print("%s (%d layers, heads %d/%d, kv %s) through the reference's model code, %s provider: tokens %s, logits rel l2 vs fp64 "
          "model max %.2e, top-1 margins %s" % (family, LAYERS if family == "llama" else GJ_LAYERS, HEADS, heads_kv, kv, mode, toks,
                                                max(errs), ["%.2f" % m for m in margins]) +
          (", %d experts (2 used), smallest router gap %.3f" % (n_experts, min(gaps)) if gaps else ""))
    assert max(errs) < 1e-2, errs
    np.savez(os.path.join(workdir, "%s_%s_%s.npz" % (mode, kv, tag)), tokens=np.array(toks), logits=logits)
    if family == "llama" and not n_experts and not device and os.environ.get("NS_WORKER_CONT_BATCH", "1") != "0":
        # continuous batching: two requests per eval (concatenated, no padding; llama.cpp:66-70, :330-350, :496-571) must
        # reproduce what each request generates alone — prompts of different lengths (two attention groups of one request)
        # and of equal length (one group of two: the kv update / attention entries see batch 2)
        ref.nellama_generate2.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p]

        def alone(p_):
            tk = (C.c_int * N_NEW)()
            lg = np.zeros((N_NEW, V), np.float32)
            assert ref.nellama_generate(qpath.encode(), (C.c_int * len(p_))(*p_), len(p_), N_NEW, N_CTX, KV[kv], tk, lg.ctypes.data) == N_NEW
            return list(tk), lg
        for pb in ([1, 5, 77, 130, 9], [1, 300, 12, 250, 7, 64, 199]):
            tk2 = (C.c_int * (2 * N_NEW))()
            lg2 = np.zeros((2, N_NEW, V), np.float32)
            assert ref.nellama_generate2(qpath.encode(), prompt, len(PROMPT), (C.c_int * len(pb))(*pb), len(pb), N_NEW, N_CTX, KV[kv],
                                         tk2, lg2.ctypes.data) == N_NEW
            tb, lb = alone(pb)
            assert list(tk2)[:N_NEW] == toks and list(tk2)[N_NEW:] == tb, (list(tk2), toks, tb)
            e0, e1 = nso.rel_l2(lg2[0], logits), nso.rel_l2(lg2[1], lb)
            assert max(e0, e1) < 2e-3, (e0, e1)
            print("continuous batching, prompts of %d and %d tokens: both requests generate what they generate alone "
                  "(logits rel l2 %.1e / %.1e)" % (len(PROMPT), len(pb), e0, e1))
    print("LLAMA_MODEL_%s_OK" % mode.upper())


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("NS_WORKER_WATCHDOG_S", "150")), exit=True)   # a hang must not eat the box
    main(*sys.argv[1:7])
