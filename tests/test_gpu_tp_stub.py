"""libns_hip.so's tensor-parallel layer with world = 2 and 4 on ONE GPU.  RCCL refuses two ranks per device, so the
collective library ns_tp.cpp binds is replaced — for these tests only, through NS_TP_RCCL_LIB — by the shared-memory
stand-in of tests/tools/stub_rccl.cpp (same ABI slice, sums in rank order).  Everything around the wire is the product's:
unique-id hand-over, ns_tp_init on every rank, reduce_add / broadcast / alltoall / barrier on device and host pointers,
HIP-graph capture and replay of a collective, the routing between the one-shot peer-memory kernel (csrc/ns_p2p.hip, real HIP
IPC between the processes) and the collective library, glue/parallel_context_hip.cpp's id-file bootstrap with the
reference's eight functions (parallel_context.h:40-47; ne_compute_forward_all_reduce's in-place reduce_add,
ne_layers.c:5466-5476), and bench.py's tp2 path with the NATIVE collectives (its line is marked INVALID: not RCCL)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "oracle", "_ref", "libpc_glue.so")  # built from glue/ against the reference header (oracle/Makefile pcglue)


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("g++ / ROCm headers absent")
    so = str(tmp_path_factory.mktemp("stub_rccl") / "libns_stub_rccl.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "tools", "stub_rccl.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", so], check=True)
    return so


def _env(stub, **extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", NS_TP_RCCL_LIB=stub, NS_TP_LOCAL_RANK="0",
               NS_STUB_RCCL_TIMEOUT_S="60", NS_P2P_TIMEOUT_MS="20000")
    env.update(extra)
    env.pop("NS_TP_ID_FILE", None)
    return env


@pytest.mark.parametrize("world", [2, 4])
def test_native_tp_layer_and_reference_glue_with_several_ranks(stub, world):
    if not os.path.exists(GLUE):
        pytest.skip("oracle/_ref/libpc_glue.so not built (needs the reference header)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29561 + world), os.path.join(ROOT, "tests", "tools", "tp_stub_worker.py"), ROOT, GLUE]
    r = subprocess.run(cmd, env=_env(stub), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        assert "TP_STUB_OK rank %d of %d" % (rank, world) in r.stdout, r.stdout[-2000:]


def test_a_missing_rank_fails_every_other_rank_instead_of_hanging(stub, tmp_path):
    """failure agreement: rank 1 never calls ns_tp_init; rank 0's ns_tp_init returns an error within the time-out"""
    code = r'''
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); L = pkg.lib()
L.ns_tp_init.restype = C.c_void_p
idb = C.create_string_buffer(128)
assert L.ns_tp_unique_id(idb) == 0
tp = L.ns_tp_init(0, 2, idb.raw, 0)
assert not tp and "ncclCommInitRank" in pkg.last_error(), pkg.last_error()
print("TP_STUB_TIMEOUT_OK")
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT], env=_env(stub, NS_STUB_RCCL_TIMEOUT_S="3"), capture_output=True, text=True,
                       timeout=300)
    assert "TP_STUB_TIMEOUT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("p2p", ["0", "1"])
def test_bench_tp2_with_the_native_collectives(stub, p2p):
    """bench.py --gpus 2: sharded weights, ns_tp_reduce_add between the GEMM runs — through the collective library alone
    (NS_P2P=0) and with the peer-memory kernel attached"""
    env = _env(stub, NS_DIST_BACKEND="gloo", NS_P2P=p2p)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29571 + int(p2p)), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--layers", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "tp2" and d["value"] > 0
    assert "INVALID" in d["config"]
    assert "ns_tp_reduce_add (native C ABI)" in d["config"]["all_reduce"], d["config"]["all_reduce"]
