"""One tensor-parallel rank of the Llama decode-stack test (started by test_gpu_fused.py; every
rank on cuda:0, Gloo process groups).  Runs one decode step and saves rank 0's logits; with
VLA_XGMI_ALLREDUCE=1 the row-parallel linears hand their unreduced split-K slabs to the
peer-memory all-reduce (which, unless VLA_XGMI_FUSE_NORM=0, also does the add + RMSNorm after
it), otherwise they reduce first and the all-reduce is staged through Gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    from vllm_legacy_amd.distributed import communication_op as comm_op
    from vllm_legacy_amd.distributed import parallel_state as ps
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
    ps.initialize_model_parallel(world, backend="gloo")
    cfg = LlamaConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2,
                      num_attention_heads=8, num_key_value_heads=2, head_dim=128, vocab_size=1024,
                      max_position_embeddings=512)
    model = LlamaDecodeModel(cfg, "awq", "fp8", device=dev, seed=3)
    for layer in model.layers:
        for lin in (layer.qkv_proj, layer.o_proj, layer.gate_up_proj, layer.down_proj):
            lin.scales.data = (lin.scales.data.float() * 0.1).half()
    # count how the all-reduce is fed
    fed_live = {"slabs": 0, "tensors": 0, "norm": 0}
    from vllm_legacy_amd.distributed.xgmi_all_reduce import XgmiAllReduce

    def counted(name, key):
        inner = getattr(XgmiAllReduce, name)

        def wrapper(self, x, *a, **k):
            fed_live[key(x)] += 1
            return inner(self, x, *a, **k)
        setattr(XgmiAllReduce, name, wrapper)

    counted("all_reduce_slabs", lambda x: "slabs")
    counted("all_reduce", lambda x: "tensors")
    counted("all_reduce_add_rms_norm", lambda x: "norm")

    bs = 16
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64] * 8, dtype=np.int32)       # 64 sequences
    B = len(seq_lens)
    max_blocks = (int(seq_lens.max()) + bs - 1) // bs
    num_blocks = B * max_blocks + 3
    rng = np.random.default_rng(11)
    block_tables = rng.permutation(num_blocks)[:B * max_blocks].reshape(B, max_blocks).astype(np.int32)
    positions = (seq_lens - 1).astype(np.int64)
    slot_mapping = np.array([block_tables[b, p // bs] * bs + p % bs for b, p in enumerate(positions)],
                            dtype=np.int64)
    input_ids = rng.integers(0, cfg.vocab_size, B).astype(np.int64)
    shape = model.kv_cache_shape(num_blocks, bs)
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    kv_caches = []
    for _ in range(cfg.num_hidden_layers):
        src = torch.empty(shape, dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
        c = torch.empty(shape, dtype=torch.uint8, device=dev)
        ops.convert_fp8(c, src)
        kv_caches.append(c)
    meta = Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B,
        slot_mapping=torch.from_numpy(slot_mapping).to(dev), seq_lens=None,
        seq_lens_tensor=torch.from_numpy(seq_lens).to(dev), max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=int(seq_lens.max()), query_start_loc=None,
        seq_start_loc=None, use_cuda_graph=False, context_lens_tensor=None,
        block_tables=torch.from_numpy(block_tables).to(dev))
    ids_t, pos_t = torch.from_numpy(input_ids).to(dev), torch.from_numpy(positions).to(dev)
    with torch.inference_mode():
        hidden = model(ids_t, pos_t, kv_caches, meta)
        logits = model.compute_logits(hidden)
    torch.cuda.synchronize()
    fed = dict(fed_live)          # how the eager step fed the all-reduce (the capture below adds calls)

    def fingerprint(h):
        bits = h.contiguous().view(torch.int16).to(torch.int64).view(-1)
        wts = torch.arange(bits.numel(), device=dev, dtype=torch.int64) % 65521 + 1
        return int((bits * wts).sum().item())

    def same_on_all_ranks(value):
        got = [None] * world
        dist.all_gather_object(got, value)
        return all(g == got[0] for g in got)

    # after the last all-reduce and the final norm every rank holds the same hidden state
    assert same_on_all_ranks(fingerprint(hidden)), "ranks disagree after the eager step"
    if os.environ.get("VLA_XGMI_ALLREDUCE") == "1":
        # the same step replayed from a graph (the peer-memory all-reduce keeps its barrier counts
        # on the device, so it can be captured; Gloo staging cannot): the same bits, on every rank
        stream = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        buf = [None]
        with torch.cuda.stream(stream), torch.inference_mode():
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                buf[0] = model(ids_t, pos_t, kv_caches, meta)
            dist.barrier()
            graph.replay()
            torch.cuda.synchronize()
        assert torch.equal(buf[0].view(torch.int16), hidden.view(torch.int16)), \
            "graph replay differs from the eager step"
        assert same_on_all_ranks(fingerprint(buf[0])), "ranks disagree after the graph replay"
    xg = ps.get_tp_xgmi_all_reduce(dev)
    if os.environ.get("VLA_XGMI_ALLREDUCE") == "1":
        assert xg is not None and not xg.disabled and xg.error_count() == 0
        if os.environ.get("VLA_XGMI_FUSE_NORM", "1") == "0":
            assert fed["slabs"] > 0 and fed["norm"] == 0, fed     # split-K layers went in unreduced
        else:
            assert fed["norm"] == 2 * cfg.num_hidden_layers and fed["slabs"] == 0, fed
    else:
        assert xg is None and fed["slabs"] + fed["norm"] + fed["tensors"] == 0, fed
    if rank == 0:
        np.savez(out_path, logits=logits.float().cpu().numpy(), slabs=fed["slabs"], tensors=fed["tensors"],
                 norm=fed["norm"])
    dist.barrier()
    ps.destroy_model_parallel()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
