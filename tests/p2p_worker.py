"""Worker of tests/test_gpu_p2p.py: one rank of a torch.distributed job whose ranks may share ONE GPU (process group over
gloo).  Checks the one-shot peer-memory all-reduce (csrc/ns_p2p.hip) against the process group's own all-reduce."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    ge.load_package()
    from neural_speed_amd import parallel as par
    ctx = par.init_parallel_context("gloo")
    rank, world = ctx.get_tp_rank(), ctx.get_tp_size()
    assert ctx.enable_p2p(64 * 1024), "peer-memory all-reduce could not be set up"
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    # eager calls: sizes around the float4 / block boundaries, many back-to-back calls (slot parity, flag reuse)
    for n in (1, 3, 4, 5, 1023, 1024, 4096, 4097, 16384):
        for it in range(5):
            x = torch.randn(n, generator=g, device="cuda")
            ref = x.clone()
            dist.all_reduce(ref)          # gloo: sum of the ranks' vectors (exact for two ranks, order-free)
            y = x.clone()
            ctx.reduce_add(y)
            torch.cuda.synchronize()
            if world == 2:
                assert torch.equal(y, ref), (n, it)
            else:
                ref64 = x.double()
                dist.all_reduce(ref64)
                assert float((y.double() - ref64).abs().max()) <= 1e-6 * max(1e-30, float(ref64.abs().max())), (n, it)
    # results are bit-identical on every rank (same summation order everywhere)
    x = torch.randn(4096, generator=g, device="cuda")
    ctx.reduce_add(x)
    gathered = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(gathered, x)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    # larger than the slot: falls through to the process group
    big = torch.ones(32 * 1024, device="cuda")
    ctx.reduce_add(big)
    assert float(big[0]) == world
    # inside a HIP graph, replayed: the sequence number lives in device memory
    a = torch.randn(4096, generator=g, device="cuda")
    buf = torch.empty_like(a)
    ref = a.clone()
    dist.all_reduce(ref)
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr.capture_begin()
        buf.copy_(a)
        ctx.reduce_add(buf)
        buf.mul_(0.5)
        ctx.reduce_add(buf)   # two dependent all-reduces per replay, like the two per layer of the decode step
        gr.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(50):
        gr.replay()
    torch.cuda.synchronize()
    expect = a.double()           # fp64 reference: the process group's fp32 ring sums in another order
    dist.all_reduce(expect)
    expect *= 0.5
    dist.all_reduce(expect)
    assert float((buf.double() - expect).abs().max()) <= 1e-6 * float(expect.abs().max())
    assert not ctx.p2p_error()
    if os.environ.get("NS_P2P_LATENCY"):  # diagnostics (scripts/final_check.sh): launch-to-launch latency inside a graph
        v = torch.randn(4096, generator=g, device="cuda") * 1e-3
        gl = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gl.capture_begin()
            for _ in range(64):
                ctx.reduce_add(v)
            gl.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        gl.replay()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gl.replay()
        e1.record()
        torch.cuda.synchronize()
        if rank == 0:
            print("P2P_LATENCY world=%d ranks on %d GPU(s): %.2f us per 16-KB all-reduce (64 per graph, 20 replays)"
                  % (world, min(world, torch.cuda.device_count()), e0.elapsed_time(e1) * 1e3 / (64 * 20)))
        assert not ctx.p2p_error()
    ctx.disable_p2p()
    dist.barrier()
    if rank == 0:
        print("P2P_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
