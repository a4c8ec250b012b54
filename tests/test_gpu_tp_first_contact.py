"""scripts/tp_first_contact.sh: what a driver runs on first contact with a multi-GPU node — the cross-device litmus
of ns_p2p.hip's hand-off, ns_tp_* over the real RCCL at 2 / 4 / 8 ranks, bench.py --gpus {1,2,4,8} with all_reduce_us / comm_fraction.
On the one-GPU test box every multi-GPU step must SKIP cleanly; the litmus worker itself is exercised with two ranks sharing cuda:0
(same kernels, same flags and payload reads, HIP IPC between the processes)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_first_contact_script_skips_cleanly_on_one_gpu(tmp_path):
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("a multi-GPU node runs the script for real: scripts/tp_first_contact.sh")
    env = dict(os.environ, NS_FC_SKIP_BENCH="1", NS_FC_OUT=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "tp_first_contact.sh"), "1000"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for n in (2, 4, 8):
        assert "(a) litmus, %d ranks: SKIP" % n in out and "(b) ns_tp over RCCL, %d ranks: SKIP" % n in out, out
    assert "TP_FIRST_CONTACT_DONE gpus=1" in out


@pytest.mark.parametrize("world", [2])
def test_litmus_worker_with_ranks_sharing_one_gpu(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NS_P2P_TIMEOUT_MS="20000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29581 + world), os.path.join(ROOT, "scripts", "tp", "p2p_litmus_worker.py"), "3000", "4096"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "P2P_LITMUS_OK world=%d" % world in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
