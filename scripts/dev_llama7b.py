"""Llama-2-7B-shaped synthetic model through the reference's UNCHANGED model code (loader, graph builder, graph executor:
oracle/_ref/libne_llama_dev_ref.so = the reference built with its own device switch -DNS_SYCL) on libns_hip.so's
bestla_device_* set: writes a BTLA-quantized NE file (int4 sym g32 bf16 scales; the product's GPU quantizer makes the
blobs), lets the reference load it with every layer offloaded and generate greedily; prints tokens/s of the single-token
evals.  `host` as first argument runs the host-pointer route (libne_llama_ref.so) on the same file for comparison.
Usage (GPU box): python scripts/dev_llama7b.py [device|host] [n_new] [n_ctx]
                 python scripts/dev_llama7b.py leg <device|host> <n_prompt> <n_new> <n_ctx> [model file]
                     one JSON line for bench.py's `reference_route` leg (round 6): prompt eval ms, per-eval times, replay statistics, tokens"""
import ctypes as C
import json
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
DEV_LIB = os.path.join(ROOT, "oracle", "_ref", "libne_llama_dev_ref.so")
HOST_LIB = os.path.join(ROOT, "oracle", "_ref", "libne_llama_ref.so")
HIP_LIB = os.path.join(ROOT, "neural-speed_amd", "libns_hip.so")


def default_path():
    """the 3.9 GB file: in memory (/dev/shm) where there is room for it, /tmp otherwise"""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 8e9:
            return "/dev/shm/ns_llama7b_q.bin"
    except OSError:
        pass
    return "/tmp/ns_llama7b_q.bin"


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
    ne_file.write(path + ".part", hp, t)
    os.replace(path + ".part", path)
    L.ns_hip_cache_clear()


def make_prompt(n_prompt):
    prompt = [1, 17, 200, 3, 99, 42, 311, 2048]
    if n_prompt > len(prompt):
        prompt = prompt + [int(t) for t in np.random.default_rng(1).integers(3, V, n_prompt - len(prompt))]
    return prompt[:max(1, n_prompt)]


def run_device(path, prompt, n_new, n_ctx):
    """-> dict: the reference's device build generating greedily on libns_hip.so (call after libns_hip.so is loaded RTLD_GLOBAL)"""
    ref = C.CDLL(DEV_LIB)
    if os.environ.get("NS_TUNE"):   # diagnostics: NS_TUNE="attn_stream_wg_target=512,attn_stream_min_keys=64" -> ns_hip_set_tuning before the run
        hip0 = C.CDLL(HIP_LIB)
        hip0.ns_hip_set_tuning.argtypes = [C.c_char_p, C.c_int]
        for kv in os.environ["NS_TUNE"].split(","):
            k, v = kv.split("=")
            assert hip0.ns_hip_set_tuning(k.strip().encode(), int(v)) == 0, kv
    toks = (C.c_int * n_new)()
    pr = (C.c_int * len(prompt))(*prompt)
    us = C.c_double(0)
    ref.nellama_generate_dev.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    t0 = time.time()
    n = ref.nellama_generate_dev(path.encode(), pr, len(prompt), n_new, n_ctx, LAYERS, toks, None, C.byref(us))
    assert n == n_new, n
    wall = time.time() - t0
    hip = C.CDLL(HIP_LIB)
    stats = (C.c_uint64 * 6)()
    hip.ns_hip_device_load_stats(stats)
    times = (C.c_double * 4096)()
    ref.nellama_eval_times.argtypes = [C.c_void_p, C.c_int]
    nt = min(4096, ref.nellama_eval_times(times, 4096))
    ev = sorted(times[i] for i in range(nt))
    tail = [times[i] for i in range(nt // 2, nt)]
    ref.nellama_prompt_us.restype = C.c_double
    pus = ref.nellama_prompt_us() if len(prompt) > 1 else 0.0
    ref.nellama_prompt_warm_us.restype = C.c_double
    pwarm = ref.nellama_prompt_warm_us()   # NS_HARNESS_PROMPT_REPEAT=1: the prompt evaluated a second time in the same context
    rs = (C.c_uint64 * 8)()
    hip.ns_hip_route_stats(rs)
    return {"n_prompt": len(prompt), "n_new": n_new, "n_ctx": n_ctx, "tokens": list(toks),
            "prompt_ms": round(pus / 1e3, 3), "prompt_tokens_per_s": round(len(prompt) * 1e6 / max(1.0, pus), 1),
            "prompt_ms_second_evaluation": round(pwarm / 1e3, 3) if pwarm else None,
            "single_token_evals": nt, "us_median": round(ev[nt // 2], 1) if nt else None,
            "tokens_per_s_median": round(1e6 / ev[nt // 2], 1) if nt else None,
            "tokens_per_s_second_half": round(1e6 * len(tail) / max(1e-9, sum(tail)), 1) if tail else None,
            "us_max": round(ev[-1], 1) if nt else None, "us_mean_all_but_first_two": round(us.value, 1),
            "replay": {"tokens_replayed": rs[0], "evaluations_not_replayed": rs[1], "plans": rs[2], "fallbacks": rs[3], "launches_per_token": rs[4],
                       "captured_launches": rs[5], "capture_failures": rs[6]},
            "load": {"btla_tensors": stats[0], "blob_MB": round(stats[1] / 1e6, 1), "streaming_layout_in_graph_slices_MB": round(stats[2] / 1e6, 1),
                     "own_allocations_MB": round(stats[3] / 1e6, 1), "seconds_in_load_storage_calls_and_the_one_sync": round(stats[4] / 1e6, 3)},
            "wall_s_incl_load": round(wall, 2)}


def run_host(path, prompt, n_new, n_ctx, kv=0):
    ref = C.CDLL(HOST_LIB)
    toks = (C.c_int * n_new)()
    pr = (C.c_int * len(prompt))(*prompt)
    ref.nellama_generate.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    t1 = time.time()
    n = ref.nellama_generate(path.encode(), pr, len(prompt), n_new, n_ctx, kv, toks, None)
    assert n == n_new, n
    ref.nellama_last_us_per_token.restype = C.c_double
    us = ref.nellama_last_us_per_token()
    return {"kv_type": kv, "n_prompt": len(prompt), "n_new": n_new, "tokens": list(toks), "us_per_token": round(us, 1),
            "tokens_per_s": round(1e6 / us, 1) if us else None, "wall_s_incl_load": round(time.time() - t1, 2)}


def leg(mode, n_prompt, n_new, n_ctx, path=None):
    """one run for bench.py: ONE JSON line on stdout (everything else goes to stderr)"""
    path = path or default_path()
    out_fd = os.dup(1)
    os.dup2(2, 1)   # the reference's loader prints to stdout
    t0 = time.time()
    built = False
    if not os.path.exists(path):
        build_file(path)
        built = True
    import torch  # noqa: F401
    C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
    prompt = make_prompt(int(n_prompt))
    res = run_device(path, prompt, int(n_new), int(n_ctx)) if mode == "device" else run_host(path, prompt, int(n_new), int(n_ctx), int(os.environ.get("NS_DEV7B_KV", "0")))
    res["model_file"] = {"path": path, "GB": round(os.path.getsize(path) / 1e9, 2), "built_here_s": round(time.time() - t0, 1) if built else None}
    sys.stdout.flush()
    os.write(out_fd, (json.dumps(res) + "\n").encode())


def main(mode="device", n_new="24", n_ctx="512"):
    n_new, n_ctx = int(n_new), int(n_ctx)
    path = os.environ.get("NS_DEV7B_FILE", "/tmp/ns_llama7b_q.bin")
    t0 = time.time()
    if not os.path.exists(path):
        build_file(path)
    print("quantized NE file: %.2f GB, written in %.0f s" % (os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
    if os.environ.get("NS_DEV7B_SCHED"):  # experiment: the runtime's wait mode (1 spin, 2 yield, 4 blocking sync) before the context exists
        hip = C.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(C.c_uint(int(os.environ["NS_DEV7B_SCHED"]))), flush=True)
    import torch  # noqa: F401
    C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
    prompt = make_prompt(int(os.environ.get("NS_DEV7B_PROMPT", "8")))  # longer prompts: the decode steps attend over that many cached positions
    t0 = time.time()
    if mode == "device":
        r = run_device(path, prompt, n_new, n_ctx)
        print(json.dumps({"load": r["load"]}), flush=True)
        if len(prompt) > 1:
            print(json.dumps({"prompt_eval": {"tokens": len(prompt), "ms": r["prompt_ms"], "tokens_per_s": r["prompt_tokens_per_s"],
                                              "ms_second_evaluation": r["prompt_ms_second_evaluation"]}}), flush=True)
        print(json.dumps({k: r[k] for k in ("replay", "single_token_evals", "us_median", "tokens_per_s_median", "tokens_per_s_second_half", "us_max")}), flush=True)
        print('{"route": "device-resident (reference built with -DNS_SYCL on bestla_device_*)", "model": "llama-2-7b-shaped synthetic, Q4_0 g32 bf16", '
              '"us_per_token": %.1f, "tokens_per_s": %.1f, "n_ctx": %d, "tokens": %s, "wall_s": %.1f}' % (
                  r["us_mean_all_but_first_two"], 1e6 / max(1.0, r["us_mean_all_but_first_two"]), n_ctx, r["tokens"][:8], time.time() - t0))
    else:
        r = run_host(path, prompt, n_new, n_ctx, int(os.environ.get("NS_DEV7B_KV", "0")))  # 0: the library-managed cache (bestla_reordered_attn_*), 1 fp16, 2 fp32 tensors
        print('{"route": "host-pointer entries (activations cross PCIe per call)", "kv_type": %d, "n_prompt": %d, "us_per_token": %.1f, '
              '"tokens_per_s": %.1f, "wall_s_incl_load": %.1f, "tokens": %s}' % (r["kv_type"], len(prompt), r["us_per_token"], r["tokens_per_s"] or 0.0,
                                                                                r["wall_s_incl_load"], r["tokens"][:8]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "leg":
        leg(*sys.argv[2:7])
    else:
        main(*sys.argv[1:4])
