"""The drop-in boundary seen from the reference's side.  oracle/_ref/libne_ref.so is the reference's OWN graph code
(/root/reference/neural_speed/core/ne_layers.c, compiled from where it lies) plus the three-function glue a maintainer
adds (oracle/ne_ref_harness.c = INTEGRATION.md section 2).  Its `ne_mul_mat` / `ne_ffn_silu` nodes over BTLA weight
tensors reach `bestla_f32f32_forward` / `bestla_fusion_FFN_SiLu_f32f32_forward` of whichever library is loaded:
 * CPU: a recording mock — the reference's argument marshalling (ne_layers.c:7312, :8037-8051) and workspace sizing
   arrive as include/ns_bestla.h declares them;
 * GPU (tests/test_gpu_reference_graph.py): libns_hip.so itself — unchanged reference graph code, HIP kernels underneath.
Each case runs in a fresh interpreter: the provider has to be loaded (RTLD_GLOBAL) before libne_ref.so."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_worker(kind, timeout=600):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libne_ref.so")) and not os.path.exists(
            "/root/reference/neural_speed/core/ne_layers.c"):
        pytest.skip("oracle/_ref/libne_ref.so not built (reference tree absent)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "ref_graph_worker.py"), kind],
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_reference_graph_marshalling_with_mock_provider():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    assert "REF_GRAPH_MOCK_OK" in run_worker("mock")



def test_reference_graph_decoder_layer_with_oracle_provider():
    """A whole Llama-style decoder layer built the way the reference's model code builds it (fused QKV, views, RoPE,
    fp16 K / V copies, permuted views, flash_attn, fused FFN, residuals) and executed by the reference's graph executor as
    ONE graph; here the CPU oracle answers the bestla_* calls (tests/tools/oracle_bestla_provider.c), on the GPU
    libns_hip.so does (tests/test_gpu_reference_graph.py runs the same graph)."""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    out = run_worker("oracle")
    assert "REF_GRAPH_ORACLE_OK" in out
    # BASELINE config 1 (plumbing, no GPU): a GPT-2-small-shaped 12-layer decoder, int4 g32, greedy decode, every layer one
    # reference graph, lm_head through ne_mul_mat — identical token ids with an fp64 model of the network
    assert "config 1 (GPT-2-small-shaped greedy decode through the reference graph): tokens" in out
