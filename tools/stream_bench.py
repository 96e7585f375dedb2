#!/usr/bin/env python3
"""What this MI355X delivers on plain streaming access (HIP events): device-to-device copy of
4 GiB (read + write), and a read-only pass (sum) over 4 GiB.  The attention roofline fraction can
be read against these as well as against the 8 TB/s data-sheet figure."""
import json

import torch


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
    return sorted(ts)[len(ts) // 2] * 1e-3


def main():
    n = 1 << 30  # float32 elements = 4 GiB
    a = torch.empty(n, dtype=torch.float32, device="cuda").uniform_()
    b = torch.empty_like(a)
    t_copy = timed(lambda: b.copy_(a))
    t_read = timed(lambda: a.sum())
    c = a.view(torch.int32)
    t_read_i = timed(lambda: torch.bitwise_xor(c[:n // 2], c[n // 2:], out=b.view(torch.int32)[:n // 2]))
    print(json.dumps({
        "copy_GBps_read_plus_write": round(2 * 4 * n / t_copy / 1e9, 1),
        "sum_GBps_read_only": round(4 * n / t_read / 1e9, 1),
        "xor_GBps_2reads_1write": round(3 * 2 * n / t_read_i / 1e9, 1),
        "props": {"name": torch.cuda.get_device_properties(0).name,
                  "cus": torch.cuda.get_device_properties(0).multi_processor_count,
                  "hbm_GiB": round(torch.cuda.get_device_properties(0).total_memory / 2**30, 1)},
    }))


if __name__ == "__main__":
    main()
