#!/usr/bin/env python3
"""Kernel study: the two forms of the prompt attention kernel (ops.set_prefill_form 1 / 2) over
prompt lengths and head layouts, TFLOP/s of causal attention (HIP events, median of 9)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def timed(fn, iters=9):
    for _ in range(3):
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
    dev = torch.device("cuda", 0)
    res = {}
    for hq, hkv, d in ((32, 8, 128), (32, 32, 128), (64, 8, 128), (32, 8, 64)):
        for nseq, s in ((128, 256), (64, 512), (32, 1024), (21, 1536), (16, 2048), (10, 3072), (8, 4096), (4, 8192)):
            t = nseq * s
            qkv = torch.empty((t, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1)
            q, k, v = [x.view(t, -1, d) for x in qkv.split([hq * d, hkv * d, hkv * d], dim=-1)]
            out = torch.empty((t, hq, d), dtype=torch.float16, device=dev)
            cu = torch.arange(0, t + 1, s, dtype=torch.int32, device=dev)
            flops = 4.0 * nseq * hq * d * s * (s + 1) / 2
            row = {}
            for form in (1, 2):
                ops.set_prefill_form(form)
                ms = timed(lambda: ops.prefill_attention(out, q, k, v, cu, s, d**-0.5))
                row["wg256" if form == 1 else "wg512"] = round(flops / ms / 1e9, 1)
            ops.set_prefill_form(0)
            res[f"h{hq}/{hkv} d{d} {nseq}x{s}"] = row
            print(f"h{hq}/{hkv} d{d} {nseq}x{s}", row, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
