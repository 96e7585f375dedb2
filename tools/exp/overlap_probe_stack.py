"""Study: does a decode step gain from running two half batches on two streams, so that one half's
paged attention (HBM-bound) shares the chip with the other half's int4 GEMMs (issue-bound)?

Builds the Llama-3-8B decode stack (AWQ, fp8 KV, context 4096) with a few layers, captures
  (a) the whole batch on one stream,
  (b) two half batches on two streams inside one graph, the second delayed by `--lag` layers'
      worth of sleep so that its GEMMs meet the first half's attention,
and prints the replay times.  Timing only; results of (b) equal those of running the halves one
after the other (each row's computation does not depend on the other rows).

usage: python tools/exp/overlap_probe.py [--layers 8] [--batch 256] [--lag-us 170]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata  # noqa: E402
from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel  # noqa: E402


def make_meta(block_tables, seq_lens, slot_mapping, S):
    return Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=block_tables.shape[0],
        slot_mapping=slot_mapping, seq_lens=None, seq_lens_tensor=seq_lens, max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=S, query_start_loc=None, seq_start_loc=None,
        use_cuda_graph=True, context_lens_tensor=None, block_tables=block_tables)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--lag-us", type=float, default=170.0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--parts", type=int, default=2)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    cfg = LlamaConfig.llama3_8b()
    cfg.num_hidden_layers = args.layers
    B, S, BS = args.batch, args.seq, 16
    torch.manual_seed(0)
    model = LlamaDecodeModel(cfg, quant="awq", kv_cache_dtype="fp8", device=device, dtype=torch.float16)
    num_blocks = B * (S // BS) + 8
    kv_caches = bench.fill_kv_caches(model, num_blocks, BS, "fp8", device)
    g = torch.Generator(device=device).manual_seed(7)
    block_tables = torch.randperm(num_blocks, generator=g, device=device)[:B * (S // BS)]
    block_tables = block_tables.to(torch.int32).view(B, S // BS).contiguous()
    seq_lens = torch.full((B, ), S, dtype=torch.int32, device=device)
    positions = torch.full((B, ), S - 1, dtype=torch.int64, device=device)
    slot_mapping = block_tables[:, (S - 1) // BS].to(torch.int64) * BS + (S - 1) % BS
    input_ids = torch.randint(0, cfg.vocab_size, (B, ), generator=g, device=device)

    def part(lo, hi):
        return (input_ids[lo:hi].contiguous(), positions[lo:hi].contiguous(),
                make_meta(block_tables[lo:hi].contiguous(), seq_lens[lo:hi].contiguous(),
                          slot_mapping[lo:hi].contiguous(), S))

    whole = part(0, B)
    P = args.parts
    halves = [part(i * B // P, (i + 1) * B // P) for i in range(P)]

    @torch.inference_mode()
    def run(p):
        return model(p[0], p[1], kv_caches, p[2])

    def time_graph(graph, stream):
        with torch.cuda.stream(stream):
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                graph.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.reps * 1e3

    s1 = torch.cuda.Stream(device=device)
    others = [torch.cuda.Stream(device=device) for _ in range(P - 1)]
    with torch.cuda.stream(s1):
        run(whole)
        for h in halves:
            run(h)
        torch.cuda.synchronize()

    # (a) whole batch
    g_whole = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_whole, stream=s1):
        out_w = run(whole)
    t_whole = time_graph(g_whole, s1)

    # (a') halves one after the other on one stream
    g_seq = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_seq, stream=s1):
        outs_seq = [run(h) for h in halves]
    t_seq = time_graph(g_seq, s1)

    # (b) halves on separate streams, later ones delayed
    # sleep cycles: torch's sleep kernel spins on the shader clock (~100 MHz wall clock counter on
    # this part is not what it uses); calibrate
    with torch.cuda.stream(s1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        torch.cuda.synchronize()
        a.record(s1)
        torch.cuda._sleep(1000000)
        b.record(s1)
        torch.cuda.synchronize()
        us_per_mcycle = a.elapsed_time(b) * 1e3
    results = {}
    for lag in (0.0, args.lag_us):
        cycles = int(lag / us_per_mcycle * 1e6)
        g_par = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_par, stream=s1):
            fork = torch.cuda.Event()
            fork.record(s1)
            outs = [None] * P
            for i, s in enumerate(others):
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    if cycles > 0:
                        torch.cuda._sleep(cycles * (i + 1))
                    outs[i + 1] = run(halves[i + 1])
            outs[0] = run(halves[0])
            for s in others:
                j = torch.cuda.Event()
                j.record(s)
                s1.wait_event(j)
        results[lag] = time_graph(g_par, s1)
        torch.cuda.synchronize()
        same = all(torch.equal(o, q) for o, q in zip(outs, outs_seq))
        print(f"lag {lag:6.1f} us: {results[lag]:.3f} ms  (outputs equal to sequential halves: {same})",
              flush=True)
    print(f"layers {args.layers} batch {B}: whole {t_whole:.3f} ms, {P} parts in sequence {t_seq:.3f} ms, "
          f"sleep calibration {us_per_mcycle:.1f} us per 1e6 cycles", flush=True)


if __name__ == "__main__":
    main()
