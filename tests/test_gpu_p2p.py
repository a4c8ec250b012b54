"""One-shot all-reduce over peer-mapped HBM (csrc/ns_p2p.hip, the MI355X form of the reference's shm_all_reduce,
shared_memory_ccl.hpp:100-139): ranks are separate processes that map each other's segment through HIP IPC.  On the
one-GPU test box the ranks share cuda:0 (their kernels run concurrently on different queues); the handles, flags and
payload reads take the same code path as across xGMI."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_all_reduce_matches_process_group(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", NS_P2P_TIMEOUT_MS="20000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29551 + world), os.path.join(ROOT, "tests", "p2p_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "P2P_OK world=%d" % world in r.stdout
