#!/usr/bin/env python3
"""Probe: does a paged-attention launch (HBM-bound) overlap with the int4 GEMMs of a layer
(MFMA-bound) when they run on two streams?  Times, from captured graphs of `iters` layers:
attention alone (batch B/2), the four GEMMs alone (M = B/2), both on two streams, and for
comparison the serial full-batch layer (attention at B, GEMMs at M = B).

Finding on MI355X: no overlap.  Two half-batch layers take 649 us one after the other and 654 us
on two streams (graph or eager); the full batch takes 547 us.  A workgroup of the GEMM holds the
whole register file of every SIMD it runs on (8 waves x 256 VGPRs), so it cannot share a CU with
attention waves, and splitting the batch only re-reads the weights."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402

SHAPES = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]


def main():
    dev = torch.device("cuda", 0)
    B, S, BS, D, hq, hkv = 256, 4096, 16, 128, 32, 8
    iters, nsets = 8, 4
    g = torch.Generator(device=dev).manual_seed(0)
    nb = B * (S // BS)
    caches = []
    for _ in range(2):
        kc = torch.randint(0, 120, (nb, hkv, D // 16, BS, 16), dtype=torch.uint8, device=dev, generator=g)
        vc = torch.randint(0, 120, (nb, hkv, D, BS), dtype=torch.uint8, device=dev, generator=g)
        caches.append((kc, vc))
    bt = torch.randperm(nb, generator=g, device=dev).to(torch.int32).view(B, S // BS).contiguous()
    sl = torch.full((B, ), S, dtype=torch.int32, device=dev)
    q = torch.empty((B, hq, D), dtype=torch.float16, device=dev).uniform_(-0.1, 0.1, generator=g)
    out = torch.empty_like(q)
    sets = []
    for _ in range(nsets):
        layer = []
        for k, n in SHAPES:
            qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
            qz = torch.randint(-2**31, 2**31 - 1, (k // 128, n // 8), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
            sc = (torch.empty((k // 128, n), device=dev).uniform_(0.5, 1.5, generator=g) * 1e-2).half()
            pw, pz = ops.awq_repack(qw, qz)
            layer.append((pw, pz, sc))
        sets.append(layer)
    xs = {m: [torch.empty((m, k), device=dev).uniform_(-1, 1, generator=g).half() for k, _ in SHAPES]
          for m in (B // 2, B)}

    def attn(i, lo, hi):
        kc, vc = caches[i % 2]
        ops.paged_attention_v1(out[lo:hi], q[lo:hi], kc, vc, hkv, D**-0.5, bt[lo:hi], sl[lo:hi], BS, S, None, "fp8", 1.0)

    def gemms(i, m):
        for (pw, pz, sc), x in zip(sets[i % nsets], xs[m]):
            ops.q4_gemm_packed(x, pw, pz, sc, 0)

    def timed(build):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            build(stream)               # warm-up (allocations, lazy inits)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                build(stream)
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                graph.replay()
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        return round(sorted(ts)[2], 1)

    side = torch.cuda.Stream()

    def attn_half(stream):
        for i in range(iters):
            attn(i, 0, B // 2)

    def gemm_half(stream):
        for i in range(iters):
            gemms(i, B // 2)

    def serial_full(stream):
        for i in range(iters):
            gemms(i, B)
            attn(i, 0, B)

    def serial_halves(stream):
        for i in range(iters):
            gemms(i, B // 2)
            attn(i, 0, B // 2)
            gemms(i, B // 2)
            attn(i, B // 2, B)

    def overlapped(stream):
        # per layer: [attention of half 1 || GEMMs of half 2] then [GEMMs of half 1 || attention of half 2]
        for i in range(iters):
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                gemms(i, B // 2)
            attn(i, 0, B // 2)
            stream.wait_stream(side)
            side.wait_stream(stream)
            with torch.cuda.stream(side):
                attn(i, B // 2, B)
            gemms(i, B // 2)
            stream.wait_stream(side)

    def timed_eager(build):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            build(stream)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                build(stream)
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        return round(sorted(ts)[2], 1)

    eager = {"eager_serial_two_halves_us": timed_eager(serial_halves),
             "eager_overlapped_two_halves_us": timed_eager(overlapped)}
    res = {"attention_half_us": timed(attn_half), "gemms_half_us": timed(gemm_half),
           "serial_full_layer_us": timed(serial_full), "serial_two_halves_us": timed(serial_halves),
           "overlapped_two_halves_us": timed(overlapped)}
    res.update(eager)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
