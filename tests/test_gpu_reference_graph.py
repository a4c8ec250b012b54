"""The reference's own graph executor (ne_layers.c, compiled into oracle/_ref/libne_ref.so) running its BTLA matmul and
fused FFN nodes on libns_hip.so: see tests/test_reference_graph.py."""
import pytest

from test_reference_graph import run_worker

pytestmark = pytest.mark.gpu


def test_reference_graph_runs_on_the_hip_library():
    assert "REF_GRAPH_PRODUCT_OK" in run_worker("product")
