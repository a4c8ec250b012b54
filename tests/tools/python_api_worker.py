"""Worker of tests/test_python_api.py / tests/test_gpu_python_api.py: the reference's Python-facing layers on the library.

  python_api_worker.py oracle  <workdir>   (needs /root/reference: the reference's OWN Python package `neural_speed`,
      imported from where it lies, with oracle/_ref/llama_cpp.so — its pybind module built from its unchanged sources,
      oracle/Makefile nepy — registered as neural_speed.llama_cpp; the CPU oracle answers bestla_*)
      ->  neural_speed.Model().init_from_bin("llama", file, ...) ; Model.generate(input_ids, max_new_tokens=N)
  python_api_worker.py product <workdir>   (GPU box, no reference tree: the same pybind module driven directly, libns_hip.so
      answering bestla_*)
      ->  llama_cpp.Model().init_model(file, ...) ; .generate(input_ids=[[...]]) per token  (what Model.generate() calls,
          neural_speed/__init__.py:231-331, :380-384)
Both compare the token ids with the flat C harness run (tests/tools/llama_model_worker.py has the fp64 model)."""
import ctypes as C
import importlib.util
import os

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
import llama_model_worker as lw  # noqa: E402
import ne_file  # noqa: E402
import nso  # noqa: E402


def beam(mode, workdir):
    """beam search (2 beams) through the pybind Model: model_utils.cpp's beam_search drives batched evals of both beams, the
    kv-cache reorder between beams (bestla_fusion_attn_fp32_batch_cpy_k / _v when the cache is the library's) and the batch-2
    attention.  Prints BEAM_TOKENS; the GPU test compares the product's sequence with the oracle provider's."""
    os.makedirs(workdir, exist_ok=True)
    hp, tensors = lw.make_model(4)
    qpath = os.path.join(workdir, "llama_q_beam.bin")
    ne_file.write(qpath, dict(hp, ftype=ne_file.NE_FTYPE_MOSTLY_Q_BTLA), lw.quantize_tensors(tensors))
    if mode == "oracle":
        so = os.path.join(tempfile.mkdtemp(), "liboracle_bestla.so")
        nso.build()
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "tools", "oracle_bestla_provider.c"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lns_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
        provider = so
    else:
        import torch  # noqa: F401
        provider = os.path.join(ROOT, "neural-speed_amd", "libns_hip.so")
    C.CDLL(provider, mode=C.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("neural_speed.llama_cpp", os.path.join(ROOT, "oracle", "_ref", "llama_cpp.so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m = mod.Model()
    m.init_model(qpath, max_new_tokens=lw.N_NEW, ctx_size=lw.N_CTX, threads=1, num_beams=2, do_sample=False, scratch_size_ratio=0.125,
                 early_stopping=True)
    r = m.generate(input_ids=[list(lw.PROMPT)])
    print("BEAM_TOKENS %s" % list(r[0]))
    print("PYTHON_API_%s_OK beam" % mode.upper())


def main(mode, workdir):
    os.makedirs(workdir, exist_ok=True)
    heads_kv = 4
    hp, tensors = lw.make_model(heads_kv)
    qpath = os.path.join(workdir, "llama_q_py.bin")
    ne_file.write(qpath, dict(hp, ftype=ne_file.NE_FTYPE_MOSTLY_Q_BTLA), lw.quantize_tensors(tensors))
    if mode == "oracle":
        so = os.path.join(tempfile.mkdtemp(), "liboracle_bestla.so")
        nso.build()
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "tools", "oracle_bestla_provider.c"),
                               "-L" + os.path.join(ROOT, "oracle"), "-lns_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
        provider = so
    else:
        import torch  # noqa: F401
        provider = os.path.join(ROOT, "neural-speed_amd", "libns_hip.so")
    # the flat-harness answer on the same file and provider, from its own process (both libraries carry the model code's
    # static quant-layer registry, which refuses a second registration: quant_config.h:206-211); that worker checks the
    # tokens and logits against the fp64 model
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "llama_model_worker.py"), mode, workdir, "auto",
                        str(heads_kv), qpath, "llama"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    want = [int(t) for t in np.load(os.path.join(workdir, "%s_auto_%d.npz" % (mode, heads_kv)))["tokens"]]
    C.CDLL(provider, mode=C.RTLD_GLOBAL)
    # the reference's pybind module
    spec = importlib.util.spec_from_file_location("neural_speed.llama_cpp", os.path.join(ROOT, "oracle", "_ref", "llama_cpp.so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mode == "oracle":
        sys.path.insert(0, "/root/reference")
        sys.modules["neural_speed.llama_cpp"] = mod
        import torch
        import neural_speed
        neural_speed.llama_cpp = mod
        m = neural_speed.Model()
        m.init_from_bin("llama", qpath, max_new_tokens=lw.N_NEW, ctx_size=lw.N_CTX, threads=1, do_sample=False, scratch_size_ratio=0.125)
        out = m.generate(torch.tensor([lw.PROMPT]), max_new_tokens=lw.N_NEW, do_sample=False)
        got = out[0][len(lw.PROMPT):]
        print("neural_speed.Model.generate():", out)
    else:
        m = mod.Model()
        m.init_model(qpath, max_new_tokens=lw.N_NEW, ctx_size=lw.N_CTX, threads=1, do_sample=False, scratch_size_ratio=0.125)
        got, inp = [], [list(lw.PROMPT)]
        for _ in range(lw.N_NEW):
            r = m.generate(input_ids=inp)
            inp = []
            if not r:
                break
            got.extend(r[0])
        print("llama_cpp.Model.generate():", got)
    assert list(got)[:lw.N_NEW] == want, (got, want)
    print("PYTHON_API_%s_OK tokens %s" % (mode.upper(), want))


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("NS_WORKER_WATCHDOG_S", "150")), exit=True)
    (beam if len(sys.argv) > 3 and sys.argv[3] == "beam" else main)(*sys.argv[1:3])
