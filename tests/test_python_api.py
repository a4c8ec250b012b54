"""north_star: "... so the existing model graphs and Python Model.generate() API keep working unchanged".
The reference's OWN Python package (`neural_speed`, imported from /root/reference) on top of its pybind module built from
its unchanged sources with glue/shim (oracle/_ref/llama_cpp.so, oracle/Makefile nepy): Model().init_from_bin("llama", file)
and Model.generate(input_ids, max_new_tokens=6) return the prompt followed by the tokens of the flat C harness run, which
tests/tools/llama_model_worker.py checks against an fp64 model.  Here the CPU oracle answers the bestla_* calls;
tests/test_gpu_python_api.py drives the same pybind module on libns_hip.so."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(mode, workdir, extra=(), timeout=600):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "llama_cpp.so")) and not os.path.exists(
            "/root/reference/neural_speed/application/main_pybind.cpp"):
        pytest.skip("oracle/_ref/llama_cpp.so not built (reference tree absent)")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "llama_cpp.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "nellama", "nepy"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "python_api_worker.py"), mode, str(workdir)] + list(extra),
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "PYTHON_API_%s_OK" % mode.upper() in r.stdout
    return r.stdout


def test_reference_python_model_generate_on_the_oracle_provider(tmp_path):
    if not os.path.exists("/root/reference/neural_speed/__init__.py"):
        pytest.skip("the reference's Python package is not on this box")
    out = run_worker("oracle", tmp_path)
    assert "neural_speed.Model.generate(): [[1, 17, 200, 3, 99, 42, 311," in out


def test_reference_beam_search_on_the_oracle_provider(tmp_path):
    """num_beams = 2 through the pybind Model (model_utils.cpp beam_search: batched evals, kv reorder between beams)"""
    out = run_worker("oracle", tmp_path, extra=("beam",))
    assert "BEAM_TOKENS [" in out
