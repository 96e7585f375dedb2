"""Quantized Mixtral MoE block (AWQ experts on the int4 GEMM): routed-token execution must give
what the reference's dense loop gives (vllm/model_executor/models/mixtral_quant.py:129-156), and
both must agree with the same block evaluated with the oracle's dequantised weights."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import pa_testlib as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quant", ["gptq", "gptq_act"])
def test_gptq_experts_take_the_grouped_path(quant):
    """GPTQ experts, also with activation reordering (desc_act: every expert has its own row
    permutation, vllm/model_executor/layers/quantization/gptq.py:209-217), run on the grouped
    launch and give what the reference's per-expert loop gives."""
    from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
    moe = QuantMixtralMoE(512, 1024, 8, 2, quant, 128, "cuda", seed=2)
    for e in moe.experts:
        for lin in (e.w13, e.w2):
            lin.scales.data = (lin.scales.data.float() * 0.3).half()
    assert moe._grouped_ok()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((100, 512), device="cuda").normal_(0, 1, generator=g).half()
    with torch.inference_mode():
        grouped = moe.forward_grouped(x)
        dense = moe.forward_dense(x)
    torch.testing.assert_close(grouped.float(), dense.float(), atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("tokens", [1, 37, 256, 700])
def test_routed_moe_matches_dense_loop_and_oracle(tokens):
    from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
    hidden, ffn, ne = 512, 1024, 8
    moe = QuantMixtralMoE(hidden, ffn, ne, 2, "awq", 128, "cuda", seed=1)
    for e in moe.experts:       # activations O(1): see the end-to-end logits test
        for lin in (e.w13, e.w2):
            lin.scales.data = (lin.scales.data.float() * 0.3).half()
    g = torch.Generator(device="cuda").manual_seed(tokens)
    x = torch.empty((tokens, hidden), device="cuda").normal_(0, 1, generator=g).half()
    with torch.inference_mode():
        routed = moe.forward_routed(x)
        dense = moe.forward_dense(x)
        grouped = moe.forward_grouped(x)
        rw, sel = moe.route(x)
    torch.cuda.synchronize()
    # same values: a skipped expert contributes exact zeros in the dense loop; the per-expert
    # batches may get another K-split count from the GEMM than the full batch (fp32 partial sums
    # in another order), so the bar is a couple of fp16 roundings, not bit equality
    torch.testing.assert_close(routed.float(), dense.float(), atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(grouped.float(), dense.float(), atol=4e-3, rtol=4e-3)
    # every token got exactly top_k experts and normalised weights
    assert sel.shape == (tokens, 2) and torch.allclose(rw.sum(-1), torch.ones(tokens, device="cuda"))

    # oracle: dense evaluation with dequantised weights on the CPU
    xb = T.bits(x)
    want = np.zeros((tokens, hidden), dtype=np.float32)
    rw_c, sel_c = rw.cpu().numpy(), sel.cpu().numpy()
    for ei, e in enumerate(moe.experts):
        w13 = O.awq_dequantize(T.bits(e.w13.qweight), T.bits(e.w13.scales), T.bits(e.w13.qzeros))
        w2 = O.awq_dequantize(T.bits(e.w2.qweight), T.bits(e.w2.scales), T.bits(e.w2.qzeros))
        h = O.gemm_f16(O.silu_and_mul(O.gemm_f16(xb, w13), O.DT_F16), w2)
        wt = (rw_c * (sel_c == ei)).sum(-1, keepdims=True)
        want += O.from_bits(O.to_bits(O.from_bits(h, O.DT_F16) * wt, O.DT_F16), O.DT_F16)
    np.testing.assert_allclose(routed.float().cpu().numpy(), want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("tokens,experts,top_k", [(1, 8, 2), (257, 8, 2), (1000, 8, 2), (300, 64, 4),
                                                  (64, 8, 1), (50, 3, 3)])
def test_moe_route_topk_is_softmax_topk_renormalise(tokens, experts, top_k, dtype):
    """The one-launch router against the reference's three steps
    (vllm/model_executor/models/mixtral_quant.py:131-138): fp32 softmax, top-k, division by the
    sum of the chosen weights.  Logits are drawn on a coarse grid so that equal values are common,
    also across the k-th / (k+1)-th place, where the choice decides which expert runs: the same
    experts must come out (a top-k keeps the lower index among equals), and the same weights."""
    from vllm_legacy_amd import _custom_ops as ops
    g = torch.Generator(device="cuda").manual_seed(tokens + experts)
    logits = (torch.randint(-12, 13, (tokens, experts), device="cuda", generator=g).float() / 4).to(dtype)
    w, sel = ops.moe_route_topk(logits, top_k)
    p = torch.softmax(logits, dim=1, dtype=torch.float)
    tw, tsel = torch.topk(p, top_k, dim=-1)
    tw = tw / tw.sum(dim=-1, keepdim=True)
    # the chosen set, and the weight each chosen expert got (the order among equal weights is
    # the sort's business and changes nothing downstream)
    assert torch.equal(sel.sort(-1).values, tsel.sort(-1).values)
    dense = torch.zeros((tokens, experts), device="cuda").scatter_(1, sel, w)
    tdense = torch.zeros((tokens, experts), device="cuda").scatter_(1, tsel, tw)
    torch.testing.assert_close(dense, tdense, rtol=2e-7, atol=0)
    assert torch.all(w[:, :-1] >= w[:, 1:]) if top_k > 1 else True


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tokens,hidden,experts,top_k", [(1, 4096, 8, 2), (16, 4096, 8, 2),
                                                         (300, 1024, 16, 4), (7, 264, 8, 1)])
def test_moe_gate_topk_is_the_projection_followed_by_route_topk(tokens, hidden, experts, top_k, dtype):
    """The router's projection and top-k in one launch: its logits are fp32 sums of x . gate[:, e]
    rounded to the model dtype (checked against a float64 product within one rounding step of the
    model dtype), and routing from those logits equals vla_moe_route_topk's bit for bit."""
    from vllm_legacy_amd import _custom_ops as ops
    g = torch.Generator(device="cuda").manual_seed(tokens + hidden)
    x = torch.randn((tokens, hidden), device="cuda", generator=g).to(dtype)
    gate = (torch.randn((hidden, experts), device="cuda", generator=g) * hidden ** -0.5).to(dtype)
    w, sel, logits = ops.moe_gate_topk(x, gate, top_k, want_logits=True)
    exact = x.double() @ gate.double()
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert torch.all((logits.double() - exact).abs() <= eps * exact.abs().clamp(min=1.0))
    w2, sel2 = ops.moe_route_topk(logits, top_k)
    assert torch.equal(sel, sel2) and torch.equal(w, w2)
    w3, sel3 = ops.moe_gate_topk(x, gate, top_k)             # without the logits output
    assert torch.equal(sel, sel3) and torch.equal(w, w3)


@pytest.mark.parametrize("tokens,top_k,experts,local", [(1, 2, 8, 8), (37, 2, 8, 8), (256, 2, 8, 4),
                                                        (1000, 2, 8, 8), (4096, 2, 8, 8),
                                                        (300, 4, 64, 16), (64, 1, 8, 2)])
def test_moe_sort_pairs_is_a_stable_bucket_sort(tokens, top_k, experts, local):
    """Index work, bit-exact: the one-launch pair sort against a stable argsort of the bucket
    numbers -- offsets, the token of every sorted row, the row of every pair, the bucket of every
    row; experts of other ranks share the trailing bucket."""
    from vllm_legacy_amd import _custom_ops as ops
    g = torch.Generator(device="cuda").manual_seed(tokens + experts)
    scores = torch.rand((tokens, experts), device="cuda", generator=g)
    sel = torch.topk(scores, top_k, dim=-1).indices.contiguous()
    lut = torch.full((experts, ), local, dtype=torch.int32)
    owned = torch.randperm(experts, generator=torch.Generator().manual_seed(3))[:local].sort().values
    for li, e in enumerate(owned.tolist()):
        lut[e] = li
    lut = lut.cuda()
    offs, tok, pair_row, row_bucket = ops.moe_sort_pairs(sel, lut, local + 1, want_row_bucket=True)
    bucket = lut.long()[sel.reshape(-1)]
    order = torch.argsort(bucket, stable=True)
    counts = torch.bincount(bucket, minlength=local + 1)
    want_offs = torch.zeros(local + 2, dtype=torch.int64, device="cuda")
    want_offs[1:] = torch.cumsum(counts, 0)
    assert torch.equal(offs.long(), want_offs)
    assert torch.equal(tok.long(), order // top_k)
    assert torch.equal(row_bucket.long(), bucket[order])
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device="cuda")
    assert torch.equal(pair_row.long(), inv)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tokens,top_k,local", [(1, 2, 8), (100, 2, 8), (100, 2, 3), (257, 4, 5)])
def test_moe_combine_is_the_reference_accumulation(tokens, top_k, local, dtype):
    """Bit-exact against the arithmetic of the reference's loop
    (vllm/model_executor/models/mixtral_quant.py:141-151): each local expert's rows scaled by the
    fp32 routing weight and rounded, added in ascending expert order with a rounding per add;
    experts of other ranks contribute nothing."""
    from vllm_legacy_amd import _custom_ops as ops
    experts, hidden = 8, 1024
    g = torch.Generator(device="cuda").manual_seed(tokens * top_k + local)
    sel = torch.topk(torch.rand((tokens, experts), device="cuda", generator=g), top_k, dim=-1).indices
    rw = torch.rand((tokens, top_k), device="cuda", generator=g)
    rw = (rw / rw.sum(-1, keepdim=True)).contiguous()
    lut = torch.full((experts, ), local, dtype=torch.int32)
    lut[:local] = torch.arange(local, dtype=torch.int32)      # experts 0 .. local - 1 are ours
    lut = lut.cuda()
    offs, tok, pair_row, _ = ops.moe_sort_pairs(sel.contiguous(), lut, local + 1)
    y = torch.empty((tokens * top_k, hidden), device="cuda").normal_(0, 1, generator=g).to(dtype)
    # rows of the trailing bucket hold junk in the real block: the combine step must not read them
    y[offs[local].item():] = float("nan")
    got = ops.moe_combine(y, rw, sel.contiguous(), pair_row, lut, local)
    pr = pair_row.view(tokens, top_k).long()
    want = None
    for e in range(local):
        mask = sel == e                                                     # [T, top_k]
        chosen = mask.any(-1)
        rows = pr[torch.arange(tokens, device="cuda"), mask.long().argmax(-1)]
        src = y[torch.where(chosen, rows, torch.zeros_like(rows))].clone()
        src[~chosen] = 0                       # an expert's output for a token that skipped it: x 0
        w = (rw * mask).sum(-1, keepdim=True)                               # fp32 [T, 1]
        cur = (src.float() * w).to(dtype)
        want = cur if want is None else (want.float() + cur.float()).to(dtype)
    assert torch.equal(got.float(), want.float())


@pytest.mark.parametrize("tokens", [24, 256, 1100])
def test_grouped_block_equals_its_tensor_library_form(tokens):
    """The block with the routing glue in two HIP launches and the gather folded into the first
    product gives the bits of the same block written with argsort / index_select / index_add_;
    256 tokens put 64 rows per expert on average, so tiles with one, two and three live 32-row
    sub-tiles and row blocks that overflow a tile all occur."""
    from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
    moe = QuantMixtralMoE(512, 1024, 8, 2, "awq", 128, "cuda", seed=4)
    for e in moe.experts:
        for lin in (e.w13, e.w2):
            lin.scales.data = (lin.scales.data.float() * 0.3).half()
    g = torch.Generator(device="cuda").manual_seed(tokens)
    x = torch.empty((tokens, 512), device="cuda").normal_(0, 1, generator=g).half()
    with torch.inference_mode():
        got = moe.forward_grouped(x)
        rw, sel = moe.route(x)
        want = moe._forward_grouped_torch(x, rw, sel.contiguous())
    assert torch.equal(got.float(), want.float())


@pytest.mark.parametrize("world,xgmi", [(2, False), (4, False), (2, True)])
def test_moe_tensor_parallel_matches_single_rank(world, xgmi, tmp_path):
    """Experts split over ranks (np.array_split as in mixtral_quant.py:109-113), one all-reduce:
    the sum over ranks equals the single-rank block.  Ranks are processes on the one GPU with
    Gloo collectives; tokens routed to another rank's experts fall into the grouped launch's
    trailing bucket and must contribute zeros.  xgmi: the all-reduce goes through the peer-memory
    communicator that communication_op creates on first use under VLA_XGMI_ALLREDUCE=1."""
    import os
    import socket
    import subprocess
    import sys
    from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "moe_tp.npz")
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(here, "moe_tp_worker.py"), str(r), str(world),
                               str(port), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, VLA_XGMI_ALLREDUCE="1" if xgmi else "0",
                                       HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(world)]
    try:
        res = [p.communicate(timeout=100) + (p.returncode, ) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for so, se, rc in res:
        assert rc == 0, se[-3000:]
    z = np.load(out)
    assert len(z["local"]) == 8 // world
    moe = QuantMixtralMoE(512, 1024, 8, 2, "awq", 128, "cuda", seed=1)
    for e in moe.experts:
        for lin in (e.w13, e.w2):
            lin.scales.data = (lin.scales.data.float() * 0.3).half()
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.empty((96, 512), device="cuda").normal_(0, 1, generator=g).half()
    with torch.inference_mode():
        want = moe.forward_dense(x).float().cpu().numpy()
    # partial sums are rounded to fp16 per rank before the all-reduce: a few fp16 ulps
    np.testing.assert_allclose(z["grouped"], want, atol=8e-3, rtol=8e-3)
    np.testing.assert_allclose(z["dense"], want, atol=8e-3, rtol=8e-3)


def test_mixtral_decode_stack_matches_the_dense_expert_loop():
    """A two-layer Mixtral-shaped decode step (attention + quantized sparse-MoE block per layer,
    vllm/model_executor/models/mixtral_quant.py:283-296): the logits of the routed, grouped
    execution equal those of the reference's loop of every expert over every token, eagerly and
    replayed from a captured graph (the block has no host synchronisation)."""
    from vllm_legacy_amd import _custom_ops as ops
    from vllm_legacy_amd.attention.backends.gfx950_paged import Gfx950PagedAttentionMetadata
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                      num_attention_heads=8, num_key_value_heads=2, head_dim=64, vocab_size=1024,
                      max_position_embeddings=512, num_local_experts=8, num_experts_per_tok=2)
    model = LlamaDecodeModel(cfg, quant="awq", kv_cache_dtype="fp8", device=dev, seed=5)
    for layer in model.layers:
        lins = [layer.qkv_proj, layer.o_proj]
        for e in layer.block_sparse_moe.experts:
            lins += [e.w13, e.w2]
        for lin in lins:
            lin.scales.data = (lin.scales.data.float() * 0.2).half()
    bs = 16
    seq_lens = np.array([1, 5, 16, 17, 100, 200, 37, 64] * 5, dtype=np.int32)
    B = len(seq_lens)
    max_blocks = (int(seq_lens.max()) + bs - 1) // bs
    num_blocks = B * max_blocks + 3
    rng = np.random.default_rng(3)
    block_tables = rng.permutation(num_blocks)[:B * max_blocks].reshape(B, max_blocks).astype(np.int32)
    positions = (seq_lens - 1).astype(np.int64)
    slots = np.array([block_tables[b, p // bs] * bs + p % bs for b, p in enumerate(positions)], dtype=np.int64)
    ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, B).astype(np.int64)).to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    kv0 = []
    for _ in range(cfg.num_hidden_layers):
        src = torch.empty(model.kv_cache_shape(num_blocks, bs), dtype=torch.float16, device=dev)
        src.uniform_(-1, 1, generator=g)
        c = torch.empty(src.shape, dtype=torch.uint8, device=dev)
        ops.convert_fp8(c, src)
        kv0.append(c)
    meta = Gfx950PagedAttentionMetadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B,
        slot_mapping=torch.from_numpy(slots).to(dev), seq_lens=None,
        seq_lens_tensor=torch.from_numpy(seq_lens).to(dev), max_query_len=None,
        max_prefill_seq_len=0, max_decode_seq_len=int(seq_lens.max()), query_start_loc=None,
        seq_start_loc=None, use_cuda_graph=True, context_lens_tensor=None,
        block_tables=torch.from_numpy(block_tables).to(dev))
    pos = torch.from_numpy(positions).to(dev)

    def run():
        caches = [c.clone() for c in kv0]
        with torch.inference_mode():
            return model.compute_logits(model(ids, pos, caches, meta)).float()

    grouped = run()
    for layer in model.layers:
        layer.block_sparse_moe.forward = layer.block_sparse_moe.forward_dense
    dense = run()
    for layer in model.layers:
        del layer.block_sparse_moe.forward
    assert torch.isfinite(grouped).all()
    span = dense.abs().max().item()
    assert (grouped - dense).abs().max().item() <= 4e-3 * max(1.0, span)
    assert torch.equal(grouped.argmax(-1), dense.argmax(-1))
    # the same step from a graph
    caches = [c.clone() for c in kv0]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream), torch.inference_mode():
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            out = model.compute_logits(model(ids, pos, caches, meta))
        for c, c0 in zip(caches, kv0):
            c.copy_(c0)
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.float(), grouped)


@pytest.mark.parametrize("rows,cols,sets", [(1, 64, 1), (37, 4096, 9), (512, 14336, 9)])
def test_gather_perm_matches_the_index_tensor_formulation(rows, cols, sets):
    """out[r, c] = x[src_row[r], perm[row_set[r], c]] against torch's index_select + gather through
    a materialised [rows, cols] index tensor (what the act-order path used before), with and
    without the row indirection, on a row-strided x."""
    from vllm_legacy_amd import _custom_ops as ops
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    tokens = max(1, rows // 2)
    wide = torch.empty((tokens, cols + 16), dtype=torch.float16, device="cuda").normal_(0, 1, generator=g)
    x = wide[:, :cols]                                        # row stride cols + 16
    perm = torch.stack([torch.randperm(cols, device="cuda", generator=g) for _ in range(sets)]).to(torch.int32)
    row_set = torch.randint(0, sets, (rows, ), device="cuda", generator=g, dtype=torch.int32)
    src = torch.randint(0, tokens, (rows, ), device="cuda", generator=g, dtype=torch.int32)
    got = ops.moe_gather_perm(x, src, row_set, perm)
    want = torch.gather(x.index_select(0, src.long()), 1, perm.long()[row_set.long()])
    assert torch.equal(got, want)
    if rows <= tokens * 2:
        xs = x.index_select(0, src.long()).contiguous()
        got2 = ops.moe_gather_perm(xs, None, row_set, perm)
        assert torch.equal(got2, want)
