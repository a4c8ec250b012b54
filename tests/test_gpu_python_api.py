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
