#!/usr/bin/env python3
"""Probe: paged attention of one half batch (HBM-bound) beside the int4 GEMMs of the other half
batch (MFMA-bound) on two streams that own DISJOINT sets of CUs (hipExtStreamCreateWithCUMask).

Plain two-stream overlap gives nothing (tools/exp/overlap_probe.py): a GEMM workgroup fills the
register file of the SIMDs it runs on, so whichever kernel starts first owns the chip.  With CU
masks each kernel keeps its share.  Timed eagerly (a captured graph does not keep a stream's CU
mask), per layer: the serial full-batch layer, two half-batch layers one after the other, and
the two-stream schedule for several splits of the 256 CUs.

    python3 tools/exp/overlap_cumask_probe.py [--iters 8]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_legacy_amd import _custom_ops as ops  # noqa: E402

SHAPES = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
NUM_CUS = 256


def masked_stream(hip, bits):
    """A torch stream over a HIP stream restricted to the CUs whose bit is set."""
    words = (C.c_uint32 * (NUM_CUS // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), NUM_CUS // 32, words)
    if rc != 0:        # the runtime grants a process only a few masked queues
        raise RuntimeError(f"hipExtStreamCreateWithCUMask: hipError_t {rc}")
    return torch.cuda.ExternalStream(s.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    B, S, BS, D, hq, hkv = 256, 4096, 16, 128, 32, 8
    iters, nsets = a.iters, 4
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
        ops.paged_attention_v1(out[lo:hi], q[lo:hi], kc, vc, hkv, D**-0.5, bt[lo:hi], sl[lo:hi], BS, S, None,
                               "fp8", 1.0)

    def gemms(i, m):
        for (pw, pz, sc), x in zip(sets[i % nsets], xs[m]):
            ops.q4_gemm_packed(x, pw, pz, sc, 0)

    def timed(fn, stream):
        with torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                fn()
                e1.record(stream)
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / iters)
        return round(sorted(ts)[2], 1)

    main_s = torch.cuda.Stream()

    def serial_full():
        for i in range(iters):
            gemms(i, B)
            attn(i, 0, B)

    def serial_halves():
        for i in range(iters):
            gemms(i, B // 2)
            attn(i, 0, B // 2)
            gemms(i, B // 2)
            attn(i, B // 2, B)

    res = {"serial_full_layer_us": timed(serial_full, main_s),
           "serial_two_halves_us": timed(serial_halves, main_s), "splits": []}
    print(json.dumps(res), flush=True)

    def schedule(attn_s, gemm_s):
        """per layer: [attention of half 1 || GEMMs of half 2], then the halves swapped; the two
        masked streams hand over to each other through events, the timing stream waits for both"""
        def run():
            cur = torch.cuda.current_stream()
            attn_s.wait_stream(cur)
            gemm_s.wait_stream(cur)
            for i in range(iters):
                for lo in (0, B // 2):
                    with torch.cuda.stream(gemm_s):
                        gemms(i, B // 2)
                    with torch.cuda.stream(attn_s):
                        attn(i, lo, lo + B // 2)
                    # the next phase of either stream depends on both (the other half's results)
                    attn_s.wait_stream(gemm_s)
                    gemm_s.wait_stream(attn_s)
            cur.wait_stream(attn_s)
            cur.wait_stream(gemm_s)
        return run

    def alone(stream, fn):
        def run():
            cur = torch.cuda.current_stream()
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                for i in range(iters):
                    fn(i)
            cur.wait_stream(stream)
        return run

    for layout in ("contiguous", "interleaved"):
        for attn_cus in ((112, 128, 144) if layout == "contiguous" else (128, )):
            if layout == "contiguous":
                a_bits = list(range(attn_cus))
            else:   # spread the attention CUs evenly over the mask's bit positions
                a_bits = sorted({(j * NUM_CUS) // attn_cus for j in range(attn_cus)})
            g_bits = [b for b in range(NUM_CUS) if b not in set(a_bits)]
            try:
                sa, sg = masked_stream(hip, a_bits), masked_stream(hip, g_bits)
            except RuntimeError as e:
                print(json.dumps({"layout": layout, "attention_cus": attn_cus, "error": str(e)}), flush=True)
                continue
            rec = {"layout": layout, "attention_cus": len(a_bits), "gemm_cus": len(g_bits),
                   "attention_half_alone_us": timed(alone(sa, lambda i: attn(i, 0, B // 2)), main_s),
                   "gemms_half_alone_us": timed(alone(sg, lambda i: gemms(i, B // 2)), main_s),
                   "two_halves_two_masked_streams_us": timed(schedule(sa, sg), main_s)}
            res["splits"].append(rec)
            print(json.dumps(rec), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
