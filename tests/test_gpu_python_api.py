"""The reference's pybind `Model` class (application/main_pybind.cpp, built unchanged into oracle/_ref/llama_cpp.so) — the
object neural_speed.Model.generate() calls into — on libns_hip.so: init_model + generate per token, same tokens as the flat
harness run that is checked against the fp64 model.  (The Python package itself is exercised where the reference tree is:
tests/test_python_api.py.)"""
import pytest

from test_python_api import run_worker

pytestmark = pytest.mark.gpu


def test_reference_pybind_model_generates_on_the_hip_library(tmp_path):
    out = run_worker("product", tmp_path)
    assert "llama_cpp.Model.generate():" in out


def test_reference_beam_search_on_the_hip_library(tmp_path):
    """beam search with 2 beams on libns_hip.so: both beams evaluated as a batch of 2 over the library-managed kv cache, rows
    copied between the beams' caches through bestla_fusion_attn_fp32_batch_cpy_k / _v — the same sequence as the same code on
    the CPU oracle provider (its own fp16 cache)"""
    def toks(out):
        line = [ln for ln in out.splitlines() if ln.startswith("BEAM_TOKENS ")][-1]
        return line[len("BEAM_TOKENS "):]
    assert toks(run_worker("product", tmp_path, extra=("beam",))) == toks(run_worker("oracle", tmp_path, extra=("beam",)))
