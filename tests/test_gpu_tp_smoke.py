"""The tensor-parallel bench path on a GPU: two ranks share cuda:0 and all-reduce over gloo (RCCL refuses two ranks per
device), so sharding by ns_hip_weight_slice, the per-GEMM-run graph capture and the all-reduce placement are exercised
end to end on real kernels.  The line it prints is marked INVALID by bench.py (not RCCL, reduced layer count)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, port):
    env = dict(os.environ, NS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--layers", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "tp2"
    assert d["scaling"] == "strong" and d["value"] > 0
    assert "INVALID" in d["config"]  # smoke run: gloo + 2 layers
    return d


def test_tp2_bench_path_on_one_gpu():
    """all-reduces through the process group: one graph per GEMM run, the collectives eager in between"""
    d = _run({"NS_P2P": "0"}, 29541)
    assert d["config"]["launch"].startswith("hipGraph per GEMM run")
    assert d["config"]["all_reduce"].startswith("torch.distributed")


def test_tp2_bench_path_peer_memory_all_reduce():
    """default: the one-shot peer-memory all-reduce (csrc/ns_p2p.hip) — the whole token, all-reduces included, is ONE graph"""
    d = _run({}, 29542)
    assert d["config"]["all_reduce"].startswith("one-shot kernel")
    assert d["config"]["launch"] == "hipGraph replay"
