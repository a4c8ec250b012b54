#!/usr/bin/env python3
"""Litmus for ns_p2p.hip's fence-free hand-off ACROSS DEVICES (VERDICT r04 #12 / next #6a): one rank per GPU (torchrun), N back-to-back
peer-memory all-reduces whose payload is a function of (sequence number, element, rank) — integers below 2^16, so the sum over
<= 16 ranks is exact in fp32 and every element of every call has ONE right answer.  A stale payload (a flag that overtook its data:
the hazard the drained-sc1 hand-off must exclude), a torn vector or a lost update shows up as a mismatch; mismatches are counted on
the device (no host synchronisation inside the run) and the sticky error word is read at the end.  Half-way through, the ODD ranks
start a background streaming load on a side stream (uneven load is where a missing release shows, MI355X_MICROARCH.md).
argv: [calls (default 200000)] [elements (default 4096)].  Prints P2P_LITMUS_OK / P2P_LITMUS_FAIL."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    ge.load_package()
    from neural_speed_amd import parallel as par
    ctx = par.init_parallel_context("gloo")  # side channel only (IPC handles); the data path is the peer-memory kernel
    rank, world = ctx.get_tp_rank(), ctx.get_tp_size()
    if not ctx.enable_p2p(max(64 * 1024, n * 4)):
        print("P2P_LITMUS_FAIL rank %d: peer-memory segments could not be mapped (%s)" % (rank, ge.load_package().last_error()))
        sys.exit(2)
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    ranks = torch.arange(world, device="cuda", dtype=torch.int64)
    bad = torch.zeros((), device="cuda", dtype=torch.int64)
    first_bad = torch.full((), -1, device="cuda", dtype=torch.int64)
    side = torch.cuda.Stream()
    load_a = torch.randn((4096, 4096), device="cuda", dtype=torch.float16) if rank % 2 == 1 else None
    dist.barrier()
    for seq in range(calls):
        base = seq * 7 + idx * 3
        x = ((base + rank * 11) & 0xFFFF).to(torch.float32)
        want = ((base[None, :] + ranks[:, None] * 11) & 0xFFFF).sum(0).to(torch.float32)
        ctx.reduce_add(x)
        miss = (x != want).sum()
        first_bad = torch.where((first_bad < 0) & (miss > 0), torch.full_like(first_bad, seq), first_bad)
        bad += miss
        if load_a is not None and seq >= calls // 2 and seq % 8 == 0:
            with torch.cuda.stream(side):
                torch.mm(load_a, load_a)
    torch.cuda.synchronize()
    err = ctx.p2p_error()
    t = torch.tensor([int(bad.item()), int(err), int(first_bad.item())], dtype=torch.int64)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    if rank == 0:
        devs = "one GPU per rank" if ndev >= world else "%d ranks on %d GPU(s)" % (world, ndev)
        if all(int(v[0]) == 0 and int(v[1]) == 0 for v in allt):
            print("P2P_LITMUS_OK world=%d calls=%d elements=%d (%s): every sum exact on every rank, no flag time-out" % (world, calls, n, devs))
        else:
            print("P2P_LITMUS_FAIL world=%d calls=%d (%s): per rank (wrong elements, error word, first bad call) = %s"
                  % (world, calls, devs, [tuple(int(x) for x in v) for v in allt]))
            sys.exit(1)


if __name__ == "__main__":
    main()
