"""One rank of the peer-memory all-reduce test (started by test_gpu_xgmi_allreduce.py, one
process per rank, all on cuda:0: HIP IPC maps a region of the same device into another process
exactly as it maps a peer GPU's, so the handshake, the barrier protocol, the double buffering
and graph replay are exercised; only the xGMI transport itself is not)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd.distributed.xgmi_all_reduce import XgmiAllReduce  # noqa: E402


def rank_input(rank, numel, dtype, call):
    g = torch.Generator().manual_seed(1000 * call + rank)
    return torch.empty(numel, dtype=torch.float32).uniform_(-1, 1, generator=g).to(dtype)


def expected(world, numel, dtype, call):
    """fp32 sum in rank order, rounded once: the bits the kernel must produce."""
    acc = torch.zeros(numel, dtype=torch.float32)
    for r in range(world):
        acc += rank_input(r, numel, dtype, call).float()
    return acc.to(dtype)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = XgmiAllReduce(dist.group.WORLD, dev, max_size=4 << 20)
    assert not comm.disabled
    call = 0
    # ---- sizes and dtypes, out of place and in place, back to back (double buffering)
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        for numel in (8, 64, 4096 + 8, 256 * 4096, 1000 * 8):
            for inplace in (False, True):
                call += 1
                x = rank_input(rank, numel, dtype, call).to(dev)
                assert comm.should_custom_ar(x)
                out = comm.all_reduce(x, out=x if inplace else None)
                want = expected(world, numel, dtype, call)
                got = out.cpu()
                assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (
                    dtype, numel, inplace, got[:8].tolist(), want[:8].tolist())
    # ---- unreduced split-K slabs as the input: each rank's slabs are summed (fp32, slab order)
    # and rounded once, then the ranks' values are summed as above
    from vllm_legacy_amd._custom_ops import SplitKSlabs
    for dtype in (torch.float16, torch.bfloat16):
        for m, n, ns in ((3, 64, 2), (256, 4096, 8), (17, 1000, 5)):
            call += 1
            g = torch.Generator().manual_seed(77 * call)
            per_rank = [torch.empty((ns, m, n), dtype=torch.float32).uniform_(-1, 1, generator=g)
                        for _ in range(world)]
            acc = torch.zeros((m, n), dtype=torch.float32)
            for r in range(world):
                part = per_rank[r][0].clone()
                for p in range(1, ns):
                    part += per_rank[r][p]
                acc += part.to(dtype).float()
            want = acc.to(dtype)
            ws = per_rank[rank].to(dev).view(-1).view(torch.uint8)
            slabs = SplitKSlabs(ws, ns, m, n)
            assert comm.takes_slabs(m, n, dtype)
            got = comm.all_reduce_slabs(slabs, dtype).cpu()
            assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), ("slabs", dtype, m, n, ns)
            # and the same bits as reducing first and all-reducing the finished tensor
            two = comm.all_reduce(slabs.reduce(dtype).contiguous()).cpu()
            assert torch.equal(two.view(torch.uint8), got.view(torch.uint8))
    assert not comm.takes_slabs(1024, 4096, torch.float16)        # 8 MiB > this communicator's 4
    # ---- all-reduce + residual add + RMSNorm in one launch, tensor and slab inputs: the residual
    # gets the bits of the separate calls, the normalised rows agree to the last place
    from vllm_legacy_amd import _custom_ops as ops
    for dtype in (torch.float16, torch.bfloat16):
        for m, n, ns in ((1, 64 * world, 0), (37, 1024 * world, 3), (256, 4096, 0), (256, 4096, 8),
                         (5, 8192, 2)):
            if (n // 8) % world:
                assert not comm.takes_add_rms_norm(m, n, dtype)
                continue
            call += 1
            g = torch.Generator().manual_seed(91 * call)
            res0 = torch.empty((m, n), dtype=torch.float32).uniform_(-1, 1, generator=g).to(dtype)
            weight = torch.empty(n, dtype=torch.float32).uniform_(0.5, 1.5, generator=g).to(dtype)
            per_rank = [torch.empty((max(ns, 1), m, n), dtype=torch.float32).uniform_(-1, 1, generator=g)
                        for _ in range(world)]
            if ns:
                x = SplitKSlabs(per_rank[rank].to(dev).view(-1).view(torch.uint8), ns, m, n)
                plain = comm.all_reduce_slabs(x, dtype)
            else:
                x = per_rank[rank][0].to(dtype).to(dev)
                plain = comm.all_reduce(x)
            res_a, res_b = res0.to(dev), res0.to(dev)
            ops.fused_add_rms_norm(plain, res_a, weight.to(dev), 1e-5)      # plain <- normed
            got = comm.all_reduce_add_rms_norm(x, res_b, weight.to(dev), 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(res_a.view(torch.int16), res_b.view(torch.int16)), ("residual", dtype, m, n, ns)
            # a last-place step in x * inv, multiplied by the weight and rounded again: two units
            ulp = 2.0**-10 if dtype == torch.float16 else 2.0**-7
            torch.testing.assert_close(got.float(), plain.float(), rtol=2 * ulp, atol=1e-6)
            frac = (got != plain).float().mean().item()
            assert frac < 0.02, ("too many last-place differences", frac)
    # messages the communicator declines
    assert comm.custom_all_reduce(torch.zeros(7, dtype=torch.float16, device=dev)) is None
    assert comm.custom_all_reduce(torch.zeros((5 << 20), dtype=torch.float16, device=dev)) is None
    # ---- graph capture and replay with fresh inputs: barrier counts advance on the device
    numel = 256 * 4096
    xs = [torch.zeros(numel, dtype=torch.float16, device=dev) for _ in range(3)]
    outs = [torch.empty_like(x) for x in xs]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for x, o in zip(xs, outs):      # warm-up on the side stream
            comm.all_reduce(x, out=o)
        call += 3
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for x, o in zip(xs, outs):
                comm.all_reduce(x, out=o)
    for _ in range(4):
        base = call
        for i, x in enumerate(xs):
            x.copy_(rank_input(rank, numel, torch.float16, base + i + 1))
        call += 3
        graph.replay()
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            want = expected(world, numel, torch.float16, base + i + 1)
            assert torch.equal(o.cpu().view(torch.uint8), want.view(torch.uint8)), ("graph", i)
    # ---- latency of the decode message (2 MiB), informational
    x = torch.zeros(numel, dtype=torch.float16, device=dev)
    for _ in range(5):
        comm.all_reduce(x, out=x)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        comm.all_reduce(x, out=x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    # the same message followed by add + RMSNorm: two launches against the fused one
    us_two = us_fused = None
    if comm.takes_add_rms_norm(256, 4096, torch.float16):
        xm = torch.zeros((256, 4096), dtype=torch.float16, device=dev)
        res = torch.zeros_like(xm)
        wgt = torch.ones(4096, dtype=torch.float16, device=dev)

        def two():
            ops.fused_add_rms_norm(comm.all_reduce(xm), res, wgt, 1e-5)

        def fused():
            comm.all_reduce_add_rms_norm(xm, res, wgt, 1e-5)

        times = []
        for fn in (two, fused):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times.append(round(e0.elapsed_time(e1) * 1000 / 50, 1))
        us_two, us_fused = times
    errors = comm.error_count()
    comm.close()
    print(json.dumps({"rank": rank, "world": world, "ok": errors == 0, "timeouts": errors,
                      "us_per_2MiB_call_same_gpu": round(us, 1), "us_with_norm_two_launches": us_two,
                      "us_with_norm_fused": us_fused}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if errors == 0 else 3)


if __name__ == "__main__":
    main()
