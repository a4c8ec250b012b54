"""Llama-2-7B-shaped synthetic model through the reference's UNCHANGED model code (loader, graph builder, graph executor:
oracle/_ref/libne_llama_dev_ref.so = the reference built with its own device switch -DNS_SYCL) on libns_hip.so's
bestla_device_* set: writes a BTLA-quantized NE file (int4 sym g32 bf16 scales; the product's GPU quantizer makes the
blobs), lets the reference load it with every layer offloaded and generate greedily; prints tokens/s of the single-token
evals.  `host` as first argument runs the host-pointer route (libne_llama_ref.so) on the same file for comparison.
Usage (GPU box): python scripts/dev_llama7b.py [device|host] [n_new] [n_ctx]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import resource

resource.setrlimit(resource.RLIMIT_CORE, (0, 0))   # an abort of a 10 GB process must not write a core file on the GPU box
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import ne_file  # noqa: E402

V, D, HEADS, FF, LAYERS = 32000, 4096, 32, 11008, 32


def build_file(path):
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    L = pkg.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def blob(n, k, seed, scale):
        g = torch.Generator(device="cuda").manual_seed(seed)
        w = torch.randn((n, k), generator=g, device="cuda") * scale
        size = L.ns_BTLAGemmPackBSize(n, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, None)
        d = torch.zeros(size, dtype=torch.uint8, device="cuda")
        pkg.check(L.ns_hip_quant_pack_device(d.data_ptr(), w.data_ptr(), n, k, k, 32, pkg.S4, pkg.BF16, False, pkg.COMP_INT8, True, st))
        torch.cuda.synchronize()
        return (d.cpu().numpy(), n, k)
    rng = np.random.default_rng(5)
    # the token embedding as Q4_0, as the reference's own quantizer driver leaves it (llama_utils.cpp:261-265) — its loader
    # sizes the host pool for exactly that (llama_utils.cpp:100-101)
    t = [("tok_embeddings.weight", ("q4_0", rng.standard_normal((V, D), dtype=np.float32) * 0.5)),
         ("norm.weight", np.ones(D, np.float32)), ("output.weight", blob(V, D, 1, D ** -0.5))]
    for i in range(LAYERS):
        p = "layers.%d." % i
        s = 10 + i * 8
        t += [(p + "attention_norm.weight", np.ones(D, np.float32)),
              (p + "attention.wq.weight", blob(D, D, s + 0, D ** -0.5)), (p + "attention.wk.weight", blob(D, D, s + 1, D ** -0.5)),
              (p + "attention.wv.weight", blob(D, D, s + 2, D ** -0.5)), (p + "attention.wo.weight", blob(D, D, s + 3, 0.5 * D ** -0.5)),
              (p + "ffn_norm.weight", np.ones(D, np.float32)),
              (p + "feed_forward.w1.weight", blob(FF, D, s + 4, D ** -0.5)), (p + "feed_forward.w2.weight", blob(D, FF, s + 5, 0.5 * FF ** -0.5)),
              (p + "feed_forward.w3.weight", blob(FF, D, s + 6, D ** -0.5))]
    hp = dict(n_vocab=V, n_embd=D, n_mult=256, n_head=HEADS, n_head_kv=HEADS, n_layer=LAYERS, n_rot=D // HEADS,
              ftype=ne_file.NE_FTYPE_MOSTLY_Q_BTLA, max_seq_len=2048, ffn_hidden_size=FF, norm_eps=1e-5, freq_base=10000.0, freq_scale=1.0,
              rope_scaling_factor=0.0)
    ne_file.write(path, hp, t)
    L.ns_hip_cache_clear()


def main(mode="device", n_new="24", n_ctx="512"):
    n_new, n_ctx = int(n_new), int(n_ctx)
    path = "/tmp/ns_llama7b_q.bin"
    t0 = time.time()
    if not os.path.exists(path):
        build_file(path)
    print("quantized NE file: %.2f GB, written in %.0f s" % (os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
    if os.environ.get("NS_DEV7B_SCHED"):  # experiment: the runtime's wait mode (1 spin, 2 yield, 4 blocking sync) before the context exists
        hip = C.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(C.c_uint(int(os.environ["NS_DEV7B_SCHED"]))), flush=True)
    import torch  # noqa: F401
    C.CDLL(os.path.join(ROOT, "neural-speed_amd", "libns_hip.so"), mode=C.RTLD_GLOBAL)
    prompt = [1, 17, 200, 3, 99, 42, 311, 2048]
    n_prompt = int(os.environ.get("NS_DEV7B_PROMPT", "8"))  # longer prompts: the decode steps attend over that many cached positions
    if n_prompt > len(prompt):
        prompt = prompt + [int(t) for t in np.random.default_rng(1).integers(3, V, n_prompt - len(prompt))]
    toks = (C.c_int * n_new)()
    pr = (C.c_int * len(prompt))(*prompt)
    t0 = time.time()
    if mode == "device":
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libne_llama_dev_ref.so"))
        us = C.c_double(0)
        ref.nellama_generate_dev.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        n = ref.nellama_generate_dev(path.encode(), pr, len(prompt), n_new, n_ctx, LAYERS, toks, None, C.byref(us))
        assert n == n_new, n
        hip = C.CDLL(os.path.join(ROOT, "neural-speed_amd", "libns_hip.so"))
        stats = (C.c_uint64 * 6)()
        hip.ns_hip_device_load_stats(stats)
        print('{"load": {"btla_tensors": %d, "blob_MB": %.1f, "streaming_layout_in_graph_slices_MB": %.1f, "own_allocations_MB": %.1f, '
              '"seconds_in_load_storage_calls_and_the_one_sync": %.3f, "hbm_ratio_vs_blobs": %.3f}}' % (
                  stats[0], stats[1] / 1e6, stats[2] / 1e6, stats[3] / 1e6, stats[4] / 1e6, (stats[1] + stats[3]) / max(1, stats[1])), flush=True)
        times = (C.c_double * 4096)()
        ref.nellama_eval_times.argtypes = [C.c_void_p, C.c_int]
        nt = min(4096, ref.nellama_eval_times(times, 4096))
        ev = sorted(times[i] for i in range(nt))
        tail = [times[i] for i in range(nt // 2, nt)]
        if hasattr(ref, "nellama_prompt_us"):
            ref.nellama_prompt_us.restype = C.c_double
            pus = ref.nellama_prompt_us()
            print('{"prompt_eval": {"tokens": %d, "ms": %.2f, "tokens_per_s": %.0f}}' % (len(prompt), pus / 1e3, len(prompt) * 1e6 / max(1.0, pus)), flush=True)
        rs = (C.c_uint64 * 8)()
        hip.ns_hip_route_stats(rs)
        print('{"replay": {"tokens_replayed": %d, "tokens_eager": %d, "plans": %d, "fallbacks": %d, "launches_per_token": %d, "captured_launches": %d, '
              '"capture_failures": %d}, "single_token_evals": %d, "us_median": %.1f, "tokens_per_s_median": %.1f, "us_mean_second_half": %.1f, '
              '"tokens_per_s_second_half": %.1f, "us_max": %.1f}' % (rs[0], rs[1], rs[2], rs[3], rs[4], rs[5], rs[6], nt, ev[nt // 2] if nt else 0.0,
                                                                   1e6 / ev[nt // 2] if nt else 0.0, sum(tail) / max(1, len(tail)),
                                                                   1e6 * len(tail) / max(1e-9, sum(tail)), ev[-1] if nt else 0.0), flush=True)
        print('{"route": "device-resident (reference built with -DNS_SYCL on bestla_device_*)", "model": "llama-2-7b-shaped synthetic, Q4_0 g32 bf16", '
              '"us_per_token": %.1f, "tokens_per_s": %.1f, "n_ctx": %d, "tokens": %s, "wall_s": %.1f}' % (us.value, 1e6 / us.value, n_ctx, list(toks)[:8], time.time() - t0))
    else:
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libne_llama_ref.so"))
        ref.nellama_generate.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        t1 = time.time()
        kv = int(os.environ.get("NS_DEV7B_KV", "0"))  # 0: the library-managed cache (bestla_reordered_attn_*), 1 fp16, 2 fp32 tensors
        n = ref.nellama_generate(path.encode(), pr, len(prompt), n_new, n_ctx, kv, toks, None)
        assert n == n_new, n
        ref.nellama_last_us_per_token.restype = C.c_double
        us = ref.nellama_last_us_per_token()
        print('{"route": "host-pointer entries (activations cross PCIe per call)", "kv_type": %d, "n_prompt": %d, "us_per_token": %.1f, '
              '"tokens_per_s": %.1f, "wall_s_incl_load": %.1f, "tokens": %s}' % (kv, len(prompt), us, 1e6 / us if us else 0.0, time.time() - t1, list(toks)[:8]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
