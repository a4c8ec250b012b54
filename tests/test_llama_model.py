"""The reference's UNCHANGED llama model code — graph builder models/llama/llama.cpp, loader llama_utils.cpp +
model_utils/model_files.h (NE file reader, BTLA tensors), context / kv-cache set-up model_utils.cpp, graph executor
core/ne_layers.c — compiled from where it lies into oracle/_ref/libne_llama_ref.so with glue/shim in front of the include
path (oracle/Makefile target nellama) and run on a synthetic 22-layer int4 llama.  Here (no GPU) the CPU oracle answers the
bestla_* calls; tests/test_gpu_llama_model.py runs the same code on libns_hip.so.  Pass = greedy tokens and logits of an
independent fp64 model of the network."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(mode, workdir, kv, heads_kv, given=None, family="llama", gguf=False, experts=0, timeout=900, env=None):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libne_llama_ref.so")) and not os.path.exists(
            "/root/reference/neural_speed/models/llama/llama.cpp"):
        pytest.skip("oracle/_ref/libne_llama_ref.so not built (reference tree absent)")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "llama_model_worker.py"), mode, str(workdir), kv,
                        str(heads_kv), str(given) if given else "-", family], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, NS_WORKER_GGUF="1" if gguf else "0", NS_WORKER_EXPERTS=str(experts), **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "LLAMA_MODEL_%s_OK" % mode.upper() in r.stdout
    return r.stdout


@pytest.mark.parametrize("kv,heads_kv", [("f32", 4), ("f16", 2)])
def test_reference_llama_on_the_oracle_provider(tmp_path, kv, heads_kv):
    """heads_kv == heads takes the fused QKV node (ne_mul_qkv), heads_kv < heads three ne_mul_mat; both the fused FFN node;
    lm_head through ne_mul_mat over a BTLA tensor; the model's own fp32 / fp16 kv cache and unfused attention"""
    run_worker("oracle", tmp_path, kv, heads_kv)


def test_reference_gguf_route_on_the_oracle_provider(tmp_path):
    """an fp32 GGUF file (tests/tools/gguf_file.py) through the reference's GGUF reader, its quantizer driver (here on the
    oracle's packer: the blobs it writes are checked), the NE-container output with GGUF tensor names, the loader's GGUF
    branch and the llama graph"""
    out = run_worker("oracle", tmp_path, "f32", 4, gguf=True)
    assert "GGUF route:" in out


def test_reference_mixture_of_experts_llama_on_the_oracle_provider(tmp_path):
    """a Mixtral-style llama (8 experts, 2 per token) through models/llama/llama.cpp:619-683: router ne_mul_mat over a BTLA
    weight 8 columns wide, soft_max, ne_top_k (the reference's argsort.cpp), renormalised weights, ne_mul_mat_id per expert for
    the prompt and the fused ne_mul_id_ffn_silu for single tokens — against an fp64 model of the same routing.  Through the
    GGUF route: the NE branch of the reference's loader cannot load such a model (llama_utils.cpp:228-230)."""
    out = run_worker("oracle", tmp_path, "f32", 4, gguf=True, experts=8)
    assert "8 experts (2 used)" in out


@pytest.mark.parametrize("kv", ["f32", "f16"])
def test_reference_gptj_on_the_oracle_provider(tmp_path, kv):
    """a second family through its own unchanged graph builder (models/gptj/gptj.cpp): LayerNorm with bias, partial rotary,
    parallel residual, the fused gelu(x W + b) W + b node (ne_ffn_add_gelu), lm_head through ne_mul_mat_with_bias"""
    run_worker("oracle", tmp_path, kv, 4, family="gptj")


def test_glue_shim_headers_are_all_the_model_code_needs(tmp_path):
    """models/llama/llama.cpp compiles against glue/shim + the reference's own headers (no xbyak, no JIT headers)"""
    src = "/root/reference/neural_speed/models/llama/llama.cpp"
    if not os.path.exists(src) or shutil.which("g++") is None:
        pytest.skip("reference tree absent")
    ref = "/root/reference"
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + os.path.join(ROOT, "glue", "shim"), "-I" + ref,
                           "-I" + ref + "/neural_speed", "-I" + ref + "/neural_speed/core", "-I" + ref + "/bestla",
                           "-I" + ref + "/bestla/bestla", "-H", src], stderr=open(tmp_path / "inc.txt", "w"))
    included = open(tmp_path / "inc.txt").read()
    assert "glue/shim/core/layers/bestla_common.hpp" in included and "glue/shim/bestla/bestla_parallel.h" in included
    for banned in ("xbyak", "bestla_jit", "bestla_device", "bestla_prologue_b"):
        assert banned not in included, banned


def test_every_model_family_compiles_against_the_shim():
    """all of models/*/*.cpp (graph builders and loaders of every family the reference ships: baichuan, bloom, chatglm,
    falcon, gemma, gptj, gptneox, grok, llama, mpt, opt, phi, qwen, stablelm, starcoder, whisper) pass the compiler front end
    with glue/shim in place of the JIT headers — the include-path change of INTEGRATION.md section 2 is all they need"""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    ref = "/root/reference"
    srcs = sorted(glob.glob(ref + "/neural_speed/models/*/*.cpp"))
    if not srcs or shutil.which("g++") is None:
        pytest.skip("reference tree absent")
    inc = ["-I" + os.path.join(ROOT, "glue", "shim"), "-I" + ref, "-I" + ref + "/neural_speed", "-I" + ref + "/neural_speed/core",
           "-I" + ref + "/bestla", "-I" + ref + "/bestla/bestla"]

    def check(src):
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-w"] + inc + [src], capture_output=True, text=True)
        return src, r.returncode, r.stderr[-500:]
    with ThreadPoolExecutor(8) as ex:
        results = list(ex.map(check, srcs))
    failed = [(s_, e) for s_, rc, e in results if rc]
    assert len(srcs) >= 40 and not failed, failed[:3]


def test_every_bestla_symbol_the_model_families_reference_is_provided(tmp_path, pkg):
    """object files of all model sources (compiled against glue/shim): every `bestla_*` / `BTLAGemm*` symbol they leave
    undefined is either exported by libns_hip.so or defined by glue/bestla_gemm_hip.cpp (the three C++-linkage quantizer
    functions) — nothing a model family calls is missing from the drop-in"""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    ref = "/root/reference"
    srcs = sorted(glob.glob(ref + "/neural_speed/models/*/*.cpp"))
    if not srcs or shutil.which("g++") is None or shutil.which("nm") is None:
        pytest.skip("reference tree absent")
    inc = ["-I" + os.path.join(ROOT, "glue", "shim"), "-I" + ref, "-I" + ref + "/neural_speed", "-I" + ref + "/neural_speed/core",
           "-I" + ref + "/bestla", "-I" + ref + "/bestla/bestla"]

    def undefined(src):
        obj = str(tmp_path / (src.replace("/", "_") + ".o"))
        subprocess.check_call(["g++", "-std=c++17", "-O0", "-fPIC", "-w", '-DMODEL_NAME="x"'] + inc + ["-c", src, "-o", obj])
        out = subprocess.check_output(["nm", "-u", obj], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ("bestla_" in ln or "BTLA" in ln)}
    with ThreadPoolExecutor(8) as ex:
        need = set().union(*ex.map(undefined, srcs))
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH], text=True).splitlines() if ln.split()}
    missing = sorted(need - exported)
    assert len(need) >= 12
    # what is left is C++-mangled: BTLAGemmPackBSize / BTLAGemmQuantPackB / BTLAGemmPackB, the forwarders of glue/bestla_gemm_hip.cpp
    assert all(m.startswith("_Z") and "BTLAGemm" in m for m in missing) and len(missing) <= 4, missing
