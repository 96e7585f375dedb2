#!/usr/bin/env python3
"""Kernel study: quantized Mixtral MoE block at Mixtral-8x7B shapes, routed vs dense execution."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="16,64,256,1024,4096")
    ap.add_argument("--grouped-only", action="store_true", help="profiling runs: the graph path alone")
    a = ap.parse_args()
    moe = QuantMixtralMoE(4096, 14336, 8, 2, "awq", 128, "cuda", seed=0)
    res = []
    for tokens in [int(t) for t in a.tokens.split(",")]:
        x = torch.randn((tokens, 4096), device="cuda").half()
        with torch.inference_mode():
            r = d = g = 0.0
            if not a.grouped_only:
                r = timed(lambda: moe.forward_routed(x))
                d = timed(lambda: moe.forward_dense(x))
                g = timed(lambda: moe.forward_grouped(x))
            # the grouped path has no host synchronisation: time it from a graph as decode runs it
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                moe.forward_grouped(x)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    for _ in range(4):
                        moe.forward_grouped(x)
            gg = timed(graph.replay) / 4
        res.append({"tokens": tokens, "routed_ms": round(r, 3), "dense_loop_ms": round(d, 3),
                    "grouped_ms": round(g, 3), "grouped_graph_ms": round(gg, 3)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
