#!/usr/bin/env python3
"""Timing study: one launch of the prompt attention kernel from a library built with
-DVLA_PF_STAMP (VLA_LIB_PATH names it); the kernel prints per-wave cycle sums per phase.
    VLA_LIB_PATH=$PWD/tools/exp/abl/libvla_pfstamp.so python3 tools/exp/prefill_stamp.py [nseq seqlen]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda", 0)
    hq, hkv, d = 32, 8, 128
    t = nseq * s
    qkv = torch.empty((t, (hq + 2 * hkv) * d), dtype=torch.float16, device=dev).uniform_(-1, 1)
    q, k, v = [x.view(t, -1, d) for x in qkv.split([hq * d, hkv * d, hkv * d], dim=-1)]
    out = torch.empty((t, hq, d), dtype=torch.float16, device=dev)
    cu = torch.arange(0, t + 1, s, dtype=torch.int32, device=dev)
    n = int(os.environ.get("PF_LAUNCHES", "3"))
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.prefill_attention(out, q, k, v, cu, s, d**-0.5)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
        if n <= 3:
            print("launch", i, "ms", ts[-1], flush=True)
    if n > 3:
        ms = sorted(ts[n // 2:])[(n - n // 2) // 2]
        flops = 4.0 * nseq * hq * d * s * (s + 1) / 2
        print(f"median of last {n - n // 2} launches: {ms:.4f} ms = {flops / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
