#!/usr/bin/env python3
"""Kernel study: time the int4 dequant-GEMM on the Llama-3-8B layer shapes (HIP events), print JSON.

    python3 tools/gemm_bench.py [--m 256] [--iters 30] [--check]

Weights are random AWQ tensors repacked once (the linear method's steady state); 8 distinct weight
sets are cycled so that no launch finds its weights in the Infinity Cache.  --check compares each
shape against an fp32 matmul of the library's own dequantised weights.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate_up": (4096, 28672), "down": (14336, 4096)}
SHAPES_70B = {"qkv": (8192, 10240), "o": (8192, 8192), "gate_up": (8192, 57344), "down": (28672, 8192)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=256)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--tp", type=int, default=1, help="per-GPU shapes at this tensor-parallel degree")
    ap.add_argument("--model", default="8b", choices=["8b", "70b"], help="Llama-3 layer shapes")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}
    for name, (k, n) in (SHAPES if a.model == "8b" else SHAPES_70B).items():
        if a.only and name != a.only:
            continue
        if name in ("qkv", "gate_up"):
            n //= a.tp                    # column-parallel
        else:
            k //= a.tp                    # row-parallel
        sets = []
        for _ in range(a.sets):
            qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int64, device=dev,
                               generator=g).to(torch.int32)
            qz = torch.randint(-2**31, 2**31 - 1, (k // 128, n // 8), dtype=torch.int64, device=dev,
                               generator=g).to(torch.int32)
            sc = (torch.empty((k // 128, n), device=dev).uniform_(0.5, 1.5, generator=g) * 1e-2).half()
            pw, pz = ops.awq_repack(qw, qz)
            sets.append((qw, qz, sc, pw, pz))
        x = torch.empty((a.m, k), device=dev).uniform_(-1, 1, generator=g).half()
        if a.check:
            qw, qz, sc, pw, pz = sets[0]
            got = ops.q4_gemm_packed(x, pw, pz, sc, 0).float()
            want = x.float() @ ops.awq_dequantize(qw, sc, qz).float()
            err = (got - want).abs().max().item()
            ref = want.abs().max().item()
            res[name + "_max_abs_err"] = err
            res[name + "_max_abs_ref"] = ref
            assert err <= 2e-3 * ref + 2e-3, (name, err, ref)
        for i in range(3):
            ops.q4_gemm_packed(x, *sets[i % a.sets][3:], sets[i % a.sets][2], 0)
        torch.cuda.synchronize()
        # the calls are replayed from a graph so that host launch cost (two launches and a torch
        # allocation per call) does not hide the kernels' time
        stream = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(graph, stream=stream):
                for i in range(a.iters):
                    _, _, sc, pw, pz = sets[i % a.sets]
                    ops.q4_gemm_packed(x, pw, pz, sc, 0)
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                graph.replay()
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.iters)
        ts.sort()
        med = ts[len(ts) // 2]
        res[name] = {"k": k, "n": n, "us_median": round(med * 1e3, 1), "us_min": round(ts[0] * 1e3, 1),
                     "TFLOPs": round(2 * a.m * k * n / (med * 1e-3) / 1e12, 1)}
    tot = sum(v["us_median"] for v in res.values() if isinstance(v, dict))
    res["layer_us"] = round(tot, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
