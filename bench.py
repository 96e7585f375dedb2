#!/usr/bin/env python3
"""Decode-step benchmark of the gfx950 quantized paged-attention backend.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): one decode step
of Llama-3-8B with AWQ int4 weights and an fp8 KV cache at batch 256, context 4096 -- 32 layers
of {int4 qkv GEMM, rotary, KV-cache write, paged attention over 4096 cached tokens, int4 o_proj,
int4 MLP}, lm_head and greedy sampling -- on synthetic random weights and caches.  With N > 1
the model is tensor-parallel over N GPUs, one process per GPU, with RCCL all-reduces, the same
batch: strong scaling.  The N ranks are either started by the caller (`python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK / WORLD_SIZE are in the
environment) or, when `--gpus N` is given to a plain `python bench.py`, by this script itself: the
parent process touches no GPU, starts torch.distributed.run as a child, relays its output and
exits with its status.

One JSON line is printed by rank 0.  `value` is batch / step time with everything resident in
HBM.  `roofline` describes the dominant kernel (paged attention): algorithmic bytes per launch /
its launch duration measured with HIP events on the launch stream.  `cpu_baseline` times the
reference's own CPU attention kernel (oracle/_ref, built from the reference's csrc/cpu) -- or the
oracle port if the host lacks AVX-512 -- on a bounded sample; it is a baseline only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
HBM_MEASURED_READ_GBS = 7184.0  # linear read-only stream with nontemporal loads measured on the part (6409 with default-policy loads; tools/exp/read_bw.hip)
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak of the part


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--quant", default="awq", choices=["awq", "gptq", "none"])
    ap.add_argument("--kv", default="fp8", choices=["fp8", "auto"])
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"],
                    help="model dtype (the int4 linears need f16; bf16 goes with --quant none)")
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b", "mixtral-8x7b", "tiny"])
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rank-of", type=int, default=1,
                    help="kernel study: one process with the per-GPU shapes of an N-way tensor-"
                         "parallel rank (heads, FFN and vocabulary divided by N, no collectives)")
    ap.add_argument("--attn-only", action="store_true",
                    help="time only the paged-attention launches (kernel study)")
    return ap.parse_args()


def fill_kv_caches(model, num_blocks, block_size, kv, device):
    """Layer 0 gets U(-scale, scale) values (quantised with convert_fp8 for an fp8 cache, the
    reference's recipe, vllm/utils.py:324-341, :398-446); the other layers are block-rolled
    copies of it, which keeps start-up short without making any layer's data cache-resident."""
    from vllm_legacy_amd import _custom_ops as ops
    shape = model.kv_cache_shape(num_blocks, block_size)
    dtype = torch.uint8 if kv == "fp8" else model.dtype
    scale = model.cfg.head_dim**-0.5
    g = torch.Generator(device=device).manual_seed(1234)
    first = torch.empty(shape, dtype=dtype, device=device)
    flat = first.view(-1)
    chunk = 1 << 28
    for i in range(0, flat.numel(), chunk):
        n = min(chunk, flat.numel() - i)
        tmp = torch.empty(n, dtype=torch.float16, device=device).uniform_(-scale, scale, generator=g)
        if kv == "fp8":
            ops.convert_fp8(flat[i:i + n], tmp)
        else:
            flat[i:i + n] = tmp.to(dtype)
    caches = [first]
    for layer in range(1, len(model.layers)):
        # block-rolled copy built from contiguous slice copies (torch.roll's gather kernel
        # faults on tensors of more than 2^31 elements)
        c = torch.empty_like(first)
        s = (17 * layer) % num_blocks
        for half in range(2):
            c[half, s:].copy_(first[half, :num_blocks - s])
            c[half, :s].copy_(first[half, num_blocks - s:])
        caches.append(c)
    return caches


def load_profile_record(name):
    """A small JSON record under profiles/ (written by tools/profile_summary.py from rocprofv3
    output of a run on an MI355X), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def cpu_baseline(args):
    """Reference CPU paged_attention_v1 (bf16, block 16: it has no fp8 and no int4 path,
    csrc/cpu/attention.cpp:241, :429) on a bounded sample of the same attention shape."""
    import numpy as np
    from oracle import build_ref
    from oracle import oracle as O
    hq, hkv, d, bs, S = 32, 8, 128, 16, args.seq
    # 32 sequences x 4096 tokens of bf16 K and V = 0.5 GiB per call: well past any last-level
    # cache, so the rate is a DRAM rate (4 sequences, 67 MB, would be partly cache-resident)
    b = 32
    nb = b * S // bs
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    scale = d**-0.5
    g = torch.Generator().manual_seed(0)
    q = (torch.rand(b, hq, d, generator=g) * 2 - 1).mul(scale).bfloat16()
    kc = (torch.rand(nb, hkv, d // 8, bs, 8, generator=g) * 2 - 1).mul(scale).bfloat16()
    vc = (torch.rand(nb, hkv, d, bs, generator=g) * 2 - 1).mul(scale).bfloat16()
    bt = torch.randperm(nb, generator=g).to(torch.int32).view(b, S // bs)
    sl = torch.full((b, ), S, dtype=torch.int32)
    ref = build_ref.load_ref() if build_ref.ref_usable() else None
    if ref is not None:
        kind = "reference"
        out = torch.empty_like(q)

        def run():
            ref.paged_attention_v1(out, q, kc, vc, hkv, scale, bt, sl, bs, S, None, "auto", 1.0, 0,
                                   0, 0, 64, 0)
    else:
        kind = "port"
        from tests import pa_testlib as T
        qb, kb, vb = T.bits(q), T.bits(kc), T.bits(vc)
        btn, sln = bt.numpy(), sl.numpy()

        def run():
            O.paged_attention_v1(qb, kb, vb, btn, sln, scale, O.DT_BF16)
    run()
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < 12.0 and iters < 400:     # bounded: <= 12 s, <= 400 calls
        run()
        iters += 1
    dt = (time.perf_counter() - t0) / iters
    layer_calls_per_step = args.layers
    tok_s = b / (dt * layer_calls_per_step)
    gbs = 2 * hkv * d * S * b * 2 / dt / 1e9
    return {
        "value": round(tok_s, 3), "unit": "tokens/s", "cores": threads, "kind": kind,
        "sample": (f"attention only: csrc/cpu paged_attention_v1, bf16 KV (the CPU path has no fp8/int4), "
                   f"{b} sequences x {S} tokens, Hq=32 Hkv=8 D=128 block 16, {iters} timed calls, "
                   f"{dt * 1e3:.1f} ms per layer call = {gbs:.1f} GB/s; tokens/s = {b} / ({layer_calls_per_step} layer calls)"),
    }


def select_all_reduce(device, world, rank, hidden, tokens, dtype):
    """Tensor-parallel runs: try the peer-memory all-reduce on this node.  It is used for the
    timed steps only if set-up succeeds on every rank, its results equal the fp32 rank-order sum
    of the gathered inputs bit for bit on every rank over repeated calls, no barrier wait timed
    out, and it is faster than the RCCL call on the decode message; otherwise RCCL stays.
    VLA_XGMI_ALLREDUCE=0 skips the trial.  Returns a record for the result line."""
    import torch.distributed as dist
    from vllm_legacy_amd.distributed import communication_op as co
    from vllm_legacy_amd.distributed import parallel_state as ps
    rec = {"impl": "rccl", "xgmi_us": None, "rccl_us": None, "xgmi_verified": False,
           "fused_norm": False, "note": ""}
    if os.environ.get("VLA_XGMI_ALLREDUCE", "auto") == "0":
        rec["note"] = "trial skipped"
        return rec
    cpu_group = ps.get_tensor_model_parallel_cpu_group()
    comm = None
    try:
        from vllm_legacy_amd.distributed.xgmi_all_reduce import XgmiAllReduce
        comm = XgmiAllReduce(cpu_group, device)
        if comm.disabled:
            rec["note"] = comm.reason
            return rec
        def agree(flag: bool) -> bool:
            """The same answer on every rank: the calls below are collective, so whether they
            are made must not depend on what one rank saw."""
            t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=cpu_group)
            return t.item() == 1.0

        ok = True
        g = torch.Generator(device=device).manual_seed(1234 + rank)
        for it in range(24):
            n = tokens * hidden if it % 3 else 8 * (1 + it)
            x = torch.empty(n, dtype=dtype, device=device).uniform_(-1, 1, generator=g)
            parts = co.tensor_model_parallel_all_gather(x.view(1, n), dim=0)      # [world, n]
            want = torch.zeros(n, dtype=torch.float32, device=device)
            for r in range(world):
                want += parts[r].float()
            got = comm.all_reduce(x)
            ok = ok and torch.equal(got, want.to(dtype))
            # a barrier that cannot see its peers only gives up after seconds of polling: stop at
            # the first sign instead of sitting through every remaining call
            if it in (0, 1, 7) and not agree(comm.error_count() == 0):
                ok = False
                break
        ok = agree(ok and comm.error_count() == 0)
        if not ok:
            rec["note"] = "results differed or a barrier timed out"
            return rec
        # the form the row-parallel linears use: unreduced split-K slabs as the input
        slabs_ok = dtype in (torch.float16, torch.bfloat16) and comm.takes_slabs(tokens, hidden, dtype)
        if slabs_ok:
            from vllm_legacy_amd._custom_ops import SplitKSlabs
            for ns in (2, 8):
                ws = torch.empty(ns * tokens * hidden, dtype=torch.float32,
                                 device=device).uniform_(-1, 1, generator=g)
                slabs = SplitKSlabs(ws.view(torch.uint8), ns, tokens, hidden)
                got = comm.all_reduce_slabs(slabs, dtype)
                slabs_ok = slabs_ok and torch.equal(
                    got, comm.all_reduce(slabs.reduce(dtype).contiguous()))
        ok = agree(slabs_ok and comm.error_count() == 0)
        if not ok:
            rec["note"] = "the slab form differed or a barrier timed out"
            return rec
        # and the form that also does the add + RMSNorm after the all-reduce: residual bit for
        # bit, normalised rows to the last place, against the separate launches; it is switched
        # off on its own if it does not hold
        norm_ok = co.FUSE_NORM and comm.takes_add_rms_norm(tokens, hidden, dtype)
        if norm_ok:
            from vllm_legacy_amd import _custom_ops as ops
            for it in range(4):
                x = torch.empty((tokens, hidden), dtype=dtype, device=device).uniform_(-1, 1, generator=g)
                res0 = torch.empty_like(x).uniform_(-1, 1, generator=g)
                w = torch.empty(hidden, dtype=dtype, device=device).uniform_(0.5, 1.5, generator=g)
                plain, ra, rb = comm.all_reduce(x), res0.clone(), res0.clone()
                ops.fused_add_rms_norm(plain, ra, w, 1e-5)
                got = comm.all_reduce_add_rms_norm(x, rb, w, 1e-5)
                ulp = 2.0**-10 if dtype == torch.float16 else 2.0**-7
                norm_ok = norm_ok and torch.equal(ra, rb) and torch.allclose(
                    got.float(), plain.float(), rtol=2 * ulp, atol=1e-6)
        norm_ok = agree(norm_ok)
        ok = agree(comm.error_count() == 0)

        def timed(fn):
            x = torch.zeros(tokens * hidden, dtype=dtype, device=device)
            for _ in range(5):
                fn(x)
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn(x)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1000 / 50

        t_x = timed(lambda x: comm.all_reduce(x, out=x))
        t_r = timed(co.tensor_model_parallel_all_reduce)
        t_two = t_fused = 0.0
        if norm_ok:      # the fused form has to earn its place against the two launches
            from vllm_legacy_amd import _custom_ops as ops
            res = torch.zeros((tokens, hidden), dtype=dtype, device=device)
            w = torch.ones(hidden, dtype=dtype, device=device)
            t_two = timed(lambda x: ops.fused_add_rms_norm(
                comm.all_reduce(x.view(tokens, hidden)), res, w, 1e-5))
            t_fused = timed(lambda x: comm.all_reduce_add_rms_norm(x.view(tokens, hidden), res, w, 1e-5))
        ok = ok and comm.error_count() == 0
        # one decision for the whole group: everybody verified, and the slowest rank's times
        flags = torch.tensor([1.0 if ok else 0.0, -t_x, -t_r, 1.0 if norm_ok else 0.0, -t_two,
                              -t_fused], dtype=torch.float64)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=cpu_group)
        ok, t_x, t_r = flags[0].item() == 1.0, -flags[1].item(), -flags[2].item()
        t_two, t_fused = -flags[4].item(), -flags[5].item()
        norm_ok = ok and flags[3].item() == 1.0
        co.FUSE_NORM = co.FUSE_NORM and norm_ok and t_fused < t_two
        rec.update(xgmi_verified=ok, xgmi_us=round(t_x, 1), rccl_us=round(t_r, 1),
                   fused_norm=co.FUSE_NORM, fused_norm_verified=norm_ok,
                   norm_two_launches_us=round(t_two, 1), norm_fused_us=round(t_fused, 1))
        if ok and t_x < t_r:
            ps.set_tp_xgmi_all_reduce(comm)
            rec["impl"] = "xgmi"
            comm = None
        elif not ok:
            rec["note"] = "results differed or a barrier timed out"
        if rec["impl"] != "xgmi":
            rec["fused_norm"] = False
    except Exception as e:  # noqa: BLE001
        rec["note"] = f"{type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}"
    finally:
        if comm is not None and not comm.disabled:
            try:
                comm.close()
            except Exception:  # noqa: BLE001
                pass
    return rec


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, as the
    reference's multiprocessing executor does, vllm/executor/multiproc_gpu_executor.py) with
    torch.distributed.run as a child process and hand back its exit status.  This process never
    initialises the GPU (device_count() does not) and replaces nothing that has."""
    import socket
    import subprocess
    rehearsal = os.environ.get("VLA_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not rehearsal:
        print(f"[bench] --gpus {args.gpus} needs {args.gpus} GPUs, this machine shows {have}. "
              "(VLA_BENCH_REHEARSAL=1 runs the tensor-parallel code path with every rank on GPU 0 "
              "over Gloo; its numbers mean nothing.)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if os.environ.get("VLA_BENCH_WATCHDOG"):
        # debugging aid: dump every thread's stack and exit if the run is still going after N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VLA_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    stdout_fd = None
    if world > 1:
        # the result line must be the only thing on stdout: the Gloo library announces every
        # group it connects on file descriptor 1, so until the line is printed that descriptor
        # points at stderr
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # VLA_BENCH_REHEARSAL=1: run the tensor-parallel code path with every rank on GPU 0 and Gloo
    # collectives (a one-GPU box cannot host an RCCL group); numbers from it mean nothing.
    rehearsal = os.environ.get("VLA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from vllm_legacy_amd.distributed import parallel_state as ps
        ps.init_distributed_environment(local_rank=local_rank,
                                        backend="gloo" if rehearsal else "nccl")
        ps.initialize_model_parallel(world)
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): the "
                         "result would be labelled with a GPU count it was not measured on")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.float32, device=device)
        from vllm_legacy_amd.distributed import communication_op as _co
        ranks_seen = int(_co.tensor_model_parallel_all_reduce(ones).item())
        if ranks_seen != args.gpus:
            raise SystemExit(f"[bench] all-reduce of ones over the group gave {ranks_seen}, "
                             f"expected {args.gpus}")

    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel

    cfg = {"llama3-8b": LlamaConfig.llama3_8b, "llama3-70b": LlamaConfig.llama3_70b,
           "mixtral-8x7b": LlamaConfig.mixtral_8x7b, "tiny": LlamaConfig.tiny}[args.model]()
    cfg.num_hidden_layers = args.layers
    if args.rank_of > 1:
        assert world == 1
        cfg.num_attention_heads //= args.rank_of
        cfg.num_key_value_heads = max(1, cfg.num_key_value_heads // args.rank_of)
        cfg.intermediate_size //= args.rank_of
        cfg.vocab_size //= args.rank_of
    B, S, BS = args.batch, args.seq, 16
    torch.manual_seed(0)
    model_dtype = torch.float16 if args.dtype == "f16" else torch.bfloat16
    assert args.quant == "none" or args.dtype == "f16", "AWQ/GPTQ linears are fp16-only"
    model = LlamaDecodeModel(cfg, quant=args.quant, kv_cache_dtype=args.kv, device=device,
                             dtype=model_dtype)
    num_blocks = B * (S // BS) + 8
    kv_caches = fill_kv_caches(model, num_blocks, BS, args.kv, device)
    ar_rec = None
    if world > 1:
        ar_rec = select_all_reduce(device, world, rank, cfg.hidden_size, B, model_dtype)

    g = torch.Generator(device=device).manual_seed(7)
    block_tables = torch.randperm(num_blocks, generator=g, device=device)[:B * (S // BS)]
    block_tables = block_tables.to(torch.int32).view(B, S // BS).contiguous()
    seq_lens = torch.full((B, ), S, dtype=torch.int32, device=device)
    positions = torch.full((B, ), S - 1, dtype=torch.int64, device=device)
    slot_mapping = block_tables[:, (S - 1) // BS].to(torch.int64) * BS + (S - 1) % BS
    input_ids = torch.randint(0, cfg.vocab_size, (B, ), generator=g, device=device)
    meta = Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B, slot_mapping=slot_mapping,
        seq_lens=None, seq_lens_tensor=seq_lens, max_query_len=None, max_prefill_seq_len=0,
        max_decode_seq_len=S, query_start_loc=None, seq_start_loc=None, use_cuda_graph=True,
        context_lens_tensor=None, block_tables=block_tables)
    next_ids = torch.empty((B, ), dtype=torch.int64, device=device)

    hidden_buf = [None]

    @torch.inference_mode()
    def forward_part():
        hidden_buf[0] = model(input_ids, positions, kv_caches, meta)

    @torch.inference_mode()
    def logits_part():
        logits = model.compute_logits(hidden_buf[0])
        if logits is not None:
            next_ids.copy_(torch.argmax(logits, dim=-1))
        if world > 1:
            # once per step, before the point where a serving loop reads the sampled tokens: the
            # peer-memory all-reduce's 4-byte error read (a no-op on RCCL)
            from vllm_legacy_amd.distributed import communication_op as _co
            _co.tensor_model_parallel_poll_all_reduce_error(device)

    def step():
        forward_part()
        logits_part()

    # One GPU: the whole step is captured.  Tensor parallel: the graph holds the layer stack (with
    # its all-reduces), and the vocabulary-parallel logits with their gather to rank 0 run eagerly
    # after each replay -- the split the reference uses (its graphs cover the model forward only,
    # vllm/worker/model_runner.py:753-768), and a gather that cannot be captured then costs three
    # launches instead of the whole graph.
    captured = step if world == 1 else forward_part

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    graph_note = {"reason": None}

    def measure():
        """Capture (when possible), warm up and time args.steps steps; returns (seconds, graph,
        stream)."""
        # ---- optional graph capture (decode is replayed from a graph in the reference too,
        # vllm/worker/model_runner.py:512-515)
        # a rehearsal's Gloo collectives cannot be captured; with the peer-memory all-reduce in
        # place the layer stack holds none, and the rehearsal covers the graph path as well
        use_graph = not args.no_graph and (
            not rehearsal or (ar_rec is not None and ar_rec["impl"] == "xgmi"))
        if args.no_graph:
            graph_note["reason"] = "--no-graph"
        elif not use_graph:
            graph_note["reason"] = "rehearsal on Gloo collectives, which cannot be captured"
        graph = None
        stream = torch.cuda.Stream(device=device)
        with torch.cuda.stream(stream):
            step()  # first call: lazy inits (the AWQ repack, the GPTQ shuffle)
            torch.cuda.synchronize()
            if use_graph:
                try:
                    graph = torch.cuda.CUDAGraph()
                    # with collectives in the step, other threads (the process group's watchdog) make
                    # HIP calls of their own; they must not invalidate this thread's capture
                    mode = "thread_local" if world > 1 else "global"
                    with torch.cuda.graph(graph, stream=stream, capture_error_mode=mode):
                        captured()
                except Exception as e:  # e.g. a collective that cannot be captured
                    graph_note["reason"] = (f"capture failed: {type(e).__name__}: "
                                            f"{str(e).splitlines()[0] if str(e) else ''}")
                    if rank == 0:
                        print(f"[bench] graph {graph_note['reason']}; running eagerly "
                              "(hundreds of launches per step: expect a much slower step)",
                              file=sys.stderr)
                    graph = None
        if use_graph and graph is None:
            # a failed capture leaves its stream unusable: drain the device and move to a fresh one
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            stream = torch.cuda.Stream(device=device)
        if world > 1:
            graph = verify_ranks(graph, stream)
        with torch.cuda.stream(stream):
            if graph is None:
                run = step
            elif world == 1:
                run = graph.replay
            else:
                def run():
                    graph.replay()
                    logits_part()
            for _ in range(args.warmup):
                run()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            torch.cuda.synchronize()
            barrier()
            elapsed = time.perf_counter() - t0
        return elapsed, graph, stream

    verify = {"ranks_agree": None, "graph_matches_eager": None, "graph_eager_max_abs_diff": None}

    def hidden_fingerprint():
        """(64-bit position-weighted checksum of the final hidden state's bits, the state itself)."""
        h = hidden_buf[0]
        bits = h.contiguous().view(torch.int16).to(torch.int64).view(-1)
        wts = torch.arange(bits.numel(), device=device, dtype=torch.int64) % 65521 + 1
        return int((bits * wts).sum().item()), h.float().clone()

    def all_ranks_equal(value: int) -> bool:
        """Whether every rank holds the same 64-bit value (one all-gather over the Gloo group)."""
        from vllm_legacy_amd.distributed import parallel_state as ps
        mine = torch.tensor([value], dtype=torch.int64)
        got = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(got, mine, group=ps.get_tensor_model_parallel_cpu_group())
        return all(int(g.item()) == int(got[0].item()) for g in got)

    def verify_ranks(graph, stream):
        """Tensor-parallel runs, before anything is timed: one eager step and one graph replay from
        the same inputs (the step rewrites the same cache slots with the same values, so it can
        be repeated).  After the layer stack's last all-reduce and the final norm every rank must
        hold the same hidden state bit for bit -- eagerly, and again from the graph -- and the
        replayed step must reproduce the eager one: identical sampled tokens, hidden state equal
        (the same collective on the same ranks) or within fp16 accumulation noise when the captured
        all-reduce is another communicator's.  A graph whose collectives did not run would differ
        grossly.  Ranks that disagree end the run; a graph that does not reproduce the eager step
        is dropped and the steps are timed eagerly, with the reason in the result line."""
        with torch.cuda.stream(stream):
            step()
            torch.cuda.synchronize()
            sum_e, hid_e = hidden_fingerprint()
            ids_e = next_ids.clone()
            agree = all_ranks_equal(sum_e)
            if graph is not None:
                next_ids.fill_(-1)
                graph.replay()
                logits_part()
                torch.cuda.synchronize()
                sum_g, hid_g = hidden_fingerprint()
                agree = all_ranks_equal(sum_g) and agree
                diff = float((hid_g - hid_e).abs().max().item())
                scale_ = float(hid_e.abs().max().item())
                same_ids = bool(torch.equal(ids_e, next_ids)) if rank == 0 else True
                ok = same_ids and (sum_g == sum_e or diff <= 2e-2 * max(scale_, 1e-6))
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
                from vllm_legacy_amd.distributed import parallel_state as ps
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN,
                                             group=ps.get_tensor_model_parallel_cpu_group())
                verify["graph_matches_eager"] = bool(flag.item())
                verify["graph_eager_bit_identical"] = sum_g == sum_e
                verify["graph_eager_max_abs_diff"] = diff
                if not verify["graph_matches_eager"]:
                    graph_note["reason"] = ("the graph replay did not reproduce the eager step "
                                            f"(sampled tokens equal: {same_ids}, max |hidden diff| {diff:.3e} "
                                            f"on a scale of {scale_:.3e})")
                    if rank == 0:
                        print(f"[bench] {graph_note['reason']}; timing eager steps instead", file=sys.stderr)
                    graph = None
            verify["ranks_agree"] = agree
            if not agree:
                if rank == 0:
                    print("[bench] the ranks hold different hidden states after the same step: the "
                          "tensor-parallel all-reduce is broken on this node", file=sys.stderr)
                torch.distributed.barrier()
                sys.exit(3)
        return graph

    elapsed, graph, stream = measure()
    if ar_rec is not None and ar_rec["impl"] == "xgmi":
        # a barrier wait that timed out during the run voids it (the outputs since are NaN on every
        # rank): the collective verdict on the counter every step polled; on failure the
        # communicator is closed on all ranks and the steps are measured again on RCCL
        from vllm_legacy_amd.distributed import communication_op as _co3
        if not _co3.tensor_model_parallel_check_all_reduce(device):
            ar_rec.update(impl="rccl", note="barrier timeouts during the xgmi run; re-measured on RCCL")
            elapsed, graph, stream = measure()
    all_reduce_impl = "none"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        from vllm_legacy_amd.distributed import communication_op as _co2
        from vllm_legacy_amd.distributed import parallel_state as _ps2
        # what the layer stack's all-reduces ran on in the timed steps (inside the graph when
        # there is one), and why the ctypes communicator is absent if it is
        all_reduce_impl = _co2.all_reduce_implementation(device, capturing=graph is not None)
        if _ps2._TP_RCCL is not None and not _ps2._TP_RCCL.available:
            all_reduce_impl += f" (rccl-ctypes unavailable: {_ps2._TP_RCCL.reason})"
    ms_per_step = elapsed / args.steps * 1e3

    # ---- dominant kernel: paged attention, timed per launch with HIP events on its stream
    layer0 = model.layers[0]
    hq, hkv, d = layer0.hq, layer0.hkv, cfg.head_dim
    e_kv = 1 if args.kv == "fp8" else 2
    algo_bytes = (2 * hkv * d * S * B * e_kv          # every K and V byte once per KV head
                  + 4 * B * (S // BS)                  # block tables
                  + 2 * B * hq * d * 2 + 8 * B)        # q in, out, seq_lens
    q = torch.empty((B, hq, d), dtype=model.dtype, device=device).uniform_(-d**-0.5, d**-0.5)
    out = torch.empty_like(q)
    from vllm_legacy_amd.attention.ops.paged_attn import PagedAttention
    caches = [PagedAttention.split_kv_cache(c, hkv, d) for c in kv_caches]
    kvd = "fp8" if args.kv == "fp8" else "auto"
    with torch.cuda.stream(stream):
        def attn(layer):
            kc, vc = caches[layer]
            ops.paged_attention_v1(out, q, kc, vc, hkv, d**-0.5, block_tables, seq_lens, BS, S,
                                   None, kvd, 1.0)
        for layer in range(len(caches)):
            attn(layer)
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 5))
        evs = []
        for _ in range(reps):
            for layer in range(len(caches)):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                attn(layer)
                b_.record(stream)
                evs.append((a, b_))
        torch.cuda.synchronize()
        attn_ms = sum(a.elapsed_time(b_) for a, b_ in evs) / len(evs)
    achieved = algo_bytes / (attn_ms * 1e-3) / 1e9

    # ---- second kernel family: the int4 dequant-GEMMs of one layer at M = batch, timed the same
    # way (HIP events on their stream, every layer's own weights so nothing is cache-resident)
    gemm = None
    moe = None
    if args.quant != "none" and cfg.num_local_experts > 0:
        # the sparse-MoE block of one layer (router, pair sort, two grouped int4 launches around
        # SiLU * mul, weighted sum), every layer its own experts, replayed from one graph; what
        # bounds it at decode sizes is the stream of ALL local experts' weights
        with torch.cuda.stream(stream), torch.inference_mode():
            xm = torch.empty((B, cfg.hidden_size), dtype=model.dtype, device=device).normal_(0, 1)
            for lyr in model.layers:
                lyr.block_sparse_moe(xm)
            torch.cuda.synchronize()
            moe_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(moe_graph, stream=stream):
                for lyr in model.layers:
                    lyr.block_sparse_moe(xm)
            moe_graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                moe_graph.replay()
                b_.record(stream)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b_) / len(model.layers))
            moe_ms = sorted(ts)[len(ts) // 2]
        blk = layer0.block_sparse_moe
        n_loc = len(blk.expert_indicies)
        w_bytes = n_loc * 3 * cfg.hidden_size * cfg.intermediate_size // 2
        w_bytes += n_loc * 3 * (cfg.hidden_size * cfg.intermediate_size // 128) * 5 // 2   # zeros, scales
        moe = {
            "kernel": "quantized sparse-MoE block: grouped int4 launches (w13, w2: q4_gemm_skinny_kernel up to 48 tokens, q4_gemm_mid_grouped_kernel up to ~200 rows per expert, q4_gemm_m32_kernel above) + routing glue",
            "bound": "hbm", "achieved": round(w_bytes / (moe_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(w_bytes / (moe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": None, "algorithmic_bytes_per_block": w_bytes, "block_ms": round(moe_ms, 4),
            "local_experts": n_loc, "rows_per_expert_mean": B * cfg.num_experts_per_tok / cfg.num_local_experts,
            "timing": f"graph replay of {len(model.layers)} blocks, every block its own experts, HIP events / blocks",
        }
    if args.quant != "none" and cfg.num_local_experts == 0:
        names = ("qkv_proj", "o_proj", "gate_up_proj", "down_proj")
        with torch.cuda.stream(stream), torch.inference_mode():
            xs = {}
            flops = gemm_algo_bytes = 0
            for nme in names:
                lin = getattr(layer0, nme)
                k_in = lin.qweight.shape[0] * (8 if args.quant == "gptq" else 1)
                n_out = lin.scales.shape[1]
                xs[nme] = torch.empty((B, k_in), dtype=model.dtype, device=device).uniform_(-1, 1)
                flops += 2 * B * k_in * n_out
                groups = lin.scales.shape[0]
                # int4 weights + zeros (0.5 B) and scales (2 B) per group and column + x in, out
                gemm_algo_bytes += k_in * n_out // 2 + groups * n_out * 5 // 2 + 2 * B * (k_in + n_out)
            for lyr in model.layers:                      # warm-up: every layer once
                for nme in names:
                    getattr(lyr, nme)(xs[nme])
            torch.cuda.synchronize()

            def all_layers():
                for lyr in model.layers:
                    for nme in names:
                        getattr(lyr, nme)(xs[nme])

            # all layers' GEMMs (each layer its own weights) as one graph: no host launch gaps
            gemm_graph, gemm_timing = None, "eager launches (graph capture failed): includes launch gaps"
            try:
                gemm_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gemm_graph, stream=stream):
                    all_layers()
                gemm_timing = (f"graph replay of {len(model.layers)} layers x 4 GEMMs, every layer "
                               "its own weights, HIP events around the replay / layers")
            except Exception:  # noqa: BLE001
                gemm_graph = None
                torch.cuda.synchronize()
            run_gemms = gemm_graph.replay if gemm_graph is not None else all_layers
            run_gemms()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                run_gemms()
                b_.record(stream)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b_) / len(model.layers))
            layer_ms = sorted(ts)[len(ts) // 2]
        gemm_traffic = None
        rec = load_profile_record("gemm_traffic.json")
        if rec is not None and rec.get("workload") == {"m": B, "model": args.model, "quant": "int4",
                                                       "rank_of": args.rank_of, "world": world}:
            gemm_traffic = rec.get("bytes_per_layer")
        gemm = {
            "kernel": "q4_gemm_dma2_kernel (gate_up) + q4_gemm_dma_kp_kernel (qkv, o, down) + split-K reduce of one layer",
            "bound": "mfma",
            "achieved": round(flops / (layer_ms * 1e-3) / 1e12, 1),
            "peak": MFMA_F16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(flops / (layer_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            "traffic": gemm_traffic,
            "algorithmic_flops_per_layer": flops,
            "algorithmic_bytes_per_layer": gemm_algo_bytes,
            "layer_ms": round(layer_ms, 4),
            "timing": gemm_timing,
        }
        krec = load_profile_record("kernel_times.json")
        if (krec is not None and "q4_gemm_layer_us" in krec and world == 1 and
                krec.get("workload") == {"batch": B, "seq": S, "kv": args.kv, "model": args.model,
                                         "quant": args.quant}):
            gemm["rocprof_layer_us"] = krec["q4_gemm_layer_us"]
            gemm["rocprof_source"] = krec.get("source")

    # HBM bytes per launch from the committed PMC passes, when they were taken on this workload
    traffic = traffic_src = None
    attn_kernel = ops.paged_attention_kernel_name(B, hkv, d, BS, S, model.dtype, kvd)
    rec = load_profile_record("attn_traffic.json")
    try:
        w = rec["workload"]
        # only a record taken on this workload AND on the kernel instantiation this run launches
        if (w["batch"], w["seq"], w["kv"], w["num_heads"], w["num_kv_heads"], w["head_size"],
                w["block_size"]) == (B, S, args.kv, hq, hkv, d, BS) and rec["kernel"] == attn_kernel:
            traffic = rec["fetch_bytes"] + rec["write_bytes"]
            traffic_src = rec.get("command")
        elif rank == 0:
            print(f"[bench] profiles/attn_traffic.json is for {rec.get('kernel')!r} on {w}; this run "
                  f"launches {attn_kernel!r}: roofline.traffic left null", file=sys.stderr)
    except (TypeError, KeyError, ValueError):
        pass
    # the same kernel's average duration in the committed rocprofv3 kernel trace of this command
    rocprof_attn_us = rocprof_src = None
    krec = load_profile_record("kernel_times.json")
    # (a single-GPU record: the per-rank shapes of a tensor-parallel run are different kernels)
    if world == 1 and krec is not None and krec.get("workload") == {
            "batch": B, "seq": S, "kv": args.kv, "model": args.model, "quant": args.quant}:
        rocprof_attn_us, rocprof_src = krec.get("paged_attention_us"), krec.get("source")

    # ---- the whole step against its HBM floor: every cached K/V byte once per KV head, every
    # weight byte once (int4 words, zeros, scales; the 16-bit lm_head), per rank
    step_bytes = algo_bytes * len(caches)
    for lyr in model.layers:
        for prm in lyr.parameters():
            step_bytes += prm.numel() * prm.element_size()
        for bname, buf in lyr.named_buffers():
            step_bytes += buf.numel() * buf.element_size()
    step_bytes += model.lm_head.numel() * model.lm_head.element_size()
    step_bytes += B * cfg.hidden_size * model.embed_tokens.element_size()
    step_floor_ms = step_bytes / (HBM_PEAK_GBS * 1e9) * 1e3

    if rank == 0:
        result = {
            "metric": ("decode tokens/s, Llama-3-8B AWQ-int4 fp8-KV, bs=256 seq=4k (+ achieved HBM GB/s of paged attention in `roofline`)"
                       if args.model == "llama3-8b" else
                       f"decode tokens/s, {args.model} {args.quant} {args.kv}-KV, bs={B} seq={S} (+ achieved HBM GB/s of paged attention in `roofline`)"),
            "value": round(B / (ms_per_step * 1e-3), 2),
            "unit": "tokens/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} decode step, "
                            + (f"{args.quant} int4 weights" if args.quant != "none"
                               else f"{args.dtype} weights")
                            + f", {args.kv if args.kv == 'fp8' else args.dtype} KV cache, "
                            f"batch {B}, context {S}, block 16, {args.layers} layers",
                "parallelism": (f"tp{world}" if args.rank_of == 1 else
                                f"compute of one rank of tp{args.rank_of}, no collectives (study)"),
                "graph": graph is not None,
                "graph_fallback_reason": None if graph is not None else graph_note["reason"],
                "ranks_seen": ranks_seen,
                "ranks_agree": verify["ranks_agree"],
                "graph_matches_eager": verify["graph_matches_eager"],
                "graph_eager_bit_identical": verify.get("graph_eager_bit_identical"),
                "graph_eager_max_abs_diff": verify["graph_eager_max_abs_diff"],
                "all_reduce_impl": all_reduce_impl,
                "all_reduce": ar_rec,
                "global_batch": B,
                "seq_len": S,
            },
            "roofline": {
                "kernel": f"paged_attention_v1: {attn_kernel}",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # linear read-only stream measured on an MI355X with tools/exp/read_bw.hip
                "peak_measured_read": HBM_MEASURED_READ_GBS,
                "frac_of_measured_read": round(achieved / HBM_MEASURED_READ_GBS, 4),
                "frac_source": "launch_ms: HIP events around each launch on its stream, this run",
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algo_bytes,
                "launch_ms": round(attn_ms, 4),
                # rocprofv3 --kernel-trace --stats of this command on the committed build
                # (profiles/): the profiled process runs every kernel a few percent slower
                "rocprof_avg_us": rocprof_attn_us,
                "frac_rocprof": (None if not rocprof_attn_us else
                                 round(algo_bytes / (rocprof_attn_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)),
                "rocprof_source": rocprof_src,
                "attention_share_of_step": round(attn_ms * len(caches) / ms_per_step, 3),
            },
        }
        result["roofline_step"] = {
            "what": "whole decode step of one rank against the time its algorithmic HBM bytes take at peak",
            "bound": "hbm", "bytes": step_bytes, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "frac": round(step_floor_ms / ms_per_step, 4),
            "floor_ms": round(step_floor_ms, 3),
            "bytes_kv": algo_bytes * len(caches), "bytes_weights": step_bytes - algo_bytes * len(caches),
        }
        if gemm is not None:
            result["roofline_gemm"] = gemm
        if moe is not None:
            result["roofline_moe"] = moe
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0,
                                          "kind": "port", "sample": f"failed: {e}"}
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(result), flush=True)
        if stdout_fd is not None:
            os.dup2(2, 1)
    if world > 1:
        torch.distributed.barrier()
        from vllm_legacy_amd.distributed import parallel_state as ps
        ps.destroy_model_parallel()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
