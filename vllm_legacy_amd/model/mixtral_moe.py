"""Quantized Mixtral sparse-MoE block on the int4 GEMM, with routed-token execution.

Mirrors MixtralMoE / MixtralMLP of the reference's quantized Mixtral
(vllm/model_executor/models/mixtral_quant.py:53-156): fp16 router, softmax in fp32, top-k,
renormalisation, experts split across tensor-parallel ranks (np.array_split), one all-reduce at
the end.  Two things differ in execution, not in values:

* The reference runs EVERY local expert over ALL tokens and multiplies by a weight that is zero
  for the tokens not routed there (:141-151).  Here each expert only sees its routed tokens
  (gather -> w13 GEMM -> SiLU*mul -> w2 GEMM -> weighted scatter-add), 1/4 of the GEMM work at
  top-2 of 8.  Adding the skipped zero contributions would not change any element, and the
  per-token accumulation order (ascending expert index) is the reference's.
* The per-expert loop is one grouped GEMM launch per projection over rows sorted
  by expert on the device (forward_grouped): no host synchronisation, graph-capturable.
* w1 and w3 are stored side by side as one [hidden, 2*ffn] int4 matrix so that one GEMM feeds the
  fused SiLU*mul kernel (the reference's TODO at :82).
"""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from vllm_legacy_amd.distributed.communication_op import tensor_model_parallel_all_reduce
from vllm_legacy_amd.distributed.parallel_state import (get_tensor_model_parallel_rank,
                                                        get_tensor_model_parallel_world_size)
from vllm_legacy_amd import _custom_ops as ops
from vllm_legacy_amd.layers import SiluAndMul
from vllm_legacy_amd.model.llama_decode import QuantLinear


class QuantMixtralMLP(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, quant: str, group_size: int,
                 device, generator, dtype=torch.float16):
        super().__init__()
        self.w13 = QuantLinear(hidden_size, 2 * intermediate_size, quant, group_size, device,
                               generator, dtype)
        self.w2 = QuantLinear(intermediate_size, hidden_size, quant, group_size, device, generator,
                              dtype)
        self.act_fn = SiluAndMul()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.w2(self.act_fn(self.w13(x)))


class QuantMixtralMoE(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, num_experts: int = 8,
                 top_k: int = 2, quant: str = "awq", group_size: int = 128, device="cuda",
                 seed: int = 0, dtype=torch.float16):
        super().__init__()
        self.rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.num_total_experts, self.top_k = num_experts, top_k
        if self.tp_size > num_experts:
            raise ValueError(f"Tensor parallel size {self.tp_size} is greater than the number of "
                             f"experts {num_experts}.")
        self.expert_indicies: List[int] = np.array_split(range(num_experts),
                                                         self.tp_size)[self.rank].tolist()
        g = torch.Generator(device=device).manual_seed(seed)
        gate = torch.empty(hidden_size, num_experts, dtype=dtype, device=device)
        gate.normal_(0, 0.05, generator=g)
        self.gate = nn.Parameter(gate, requires_grad=False)
        # every rank draws all experts' weights in the same order so that an expert has the same
        # weights wherever it lives; only the local ones are kept
        experts: List[Optional[nn.Module]] = []
        for idx in range(num_experts):
            e = QuantMixtralMLP(hidden_size, intermediate_size, quant, group_size, device, g, dtype)
            experts.append(e if idx in self.expert_indicies else None)
        self.experts = nn.ModuleList([e if e is not None else nn.Identity() for e in experts])

    # up to this many tokens the router's projection and top-k run as one small launch; above,
    # the library GEMM's throughput wins over one workgroup per token
    GATE_FUSED_MAX_TOKENS = 1024

    def route(self, hidden_states: torch.Tensor):
        if (hidden_states.is_cuda and self.num_total_experts in (8, 16) and
                hidden_states.dtype in (torch.float16, torch.bfloat16) and
                hidden_states.shape[0] <= self.GATE_FUSED_MAX_TOKENS and
                hidden_states.shape[1] % 8 == 0 and self.gate.is_contiguous()):
            return ops.moe_gate_topk(hidden_states, self.gate, self.top_k)
        router_logits = torch.matmul(hidden_states, self.gate)
        if self.num_total_experts <= 64 and router_logits.is_cuda:
            # softmax, top-k and renormalisation in one launch (same values as the lines below)
            return ops.moe_route_topk(router_logits, self.top_k)
        routing_weights = torch.softmax(router_logits, dim=1, dtype=torch.float)
        routing_weights, selected_experts = torch.topk(routing_weights, self.top_k, dim=-1)
        routing_weights /= routing_weights.sum(dim=-1, keepdim=True)
        return routing_weights, selected_experts

    # experts the grouped launch does not take (group sizes the MFMA kernel has no path for):
    # below this many tokens the per-expert gather of forward_routed (host-synchronising) costs
    # more than it saves and the dense loop is used
    ROUTED_MIN_TOKENS = 512

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        if self._grouped_ok():
            return self.forward_grouped(hidden_states)
        if hidden_states.shape[0] < self.ROUTED_MIN_TOKENS:
            return self.forward_dense(hidden_states)
        return self.forward_routed(hidden_states)

    def _grouped_ok(self) -> bool:
        """The grouped launch takes k-packed 4-bit weights: AWQ after its re-layout, GPTQ after the
        exllama shuffle.  With activation reordering (desc_act) every expert's rows were sorted by
        its own permutation; forward_grouped then gathers each expert's activation columns by that
        permutation before the launch (what gptq_gemm does for one weight, q_gemm.cu:224-227)."""
        e = self.experts[self.expert_indicies[0]]
        if e.w13.quant == "awq":
            return e.w13.qweight.shape[0] % 64 == 0 and e.w2.qweight.shape[0] % 64 == 0
        if e.w13.quant == "gptq":
            cfg = e.w13.method.quant_config
            return cfg.weight_bits == 4 and cfg.group_size % 64 == 0
        return False

    def _stacked(self):
        """k-packed weights of the local experts back to back (made once)."""
        st = getattr(self, "_stack", None)
        if st is None:
            gptq = self.experts[self.expert_indicies[0]].w13.quant == "gptq"
            self._zero_offset = 1 if gptq else 0

            def stack(which):
                pw, pz, sc, perms = [], [], [], []
                for idx in self.expert_indicies:
                    lin = getattr(self.experts[idx], which)
                    if gptq:
                        # one tiny product runs the method's first-use shuffle (gptq.py:209-217)
                        lin.method.apply(lin, torch.zeros((1, lin.qweight.shape[0] * 8),
                                                          dtype=lin.scales.dtype,
                                                          device=lin.scales.device))
                        packed = (lin.qweight.data, lin.qzeros.data)
                        if lin.method.quant_config.desc_act:
                            perms.append(lin.g_idx.data.to(torch.int32))   # argsort(g_idx) after first use
                    else:
                        packed = getattr(lin, "_k_packed", None)
                        if packed is None:
                            packed = ops.awq_repack(lin.qweight, lin.qzeros)
                    pw.append(packed[0])
                    pz.append(packed[1])
                    sc.append(lin.scales.data)
                perm = None
                if perms:
                    # [local experts + 1, K]: the trailing bucket (rows of other ranks' experts,
                    # whose products are discarded) keeps its columns where they are
                    perms.append(torch.arange(perms[0].numel(), device=perms[0].device,
                                              dtype=torch.int32))
                    perm = torch.stack(perms).contiguous()
                return (torch.stack(pw).contiguous(), torch.stack(pz).contiguous(),
                        torch.stack(sc).contiguous(), perm)
            st = self._stack = (stack("w13"), stack("w2"))
            # local index of every global expert; experts of other ranks share one trailing bucket
            lut = torch.full((self.num_total_experts, ), len(self.expert_indicies), dtype=torch.int64)
            for li, idx in enumerate(self.expert_indicies):
                lut[idx] = li
            self._local_of = lut.to(self.gate.device)
            self._bucket_of = self._local_of.to(torch.int32).contiguous()
        return st

    def forward_grouped(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """The (token, expert) pairs are sorted by expert on the device and
        every local expert's rows go through ONE grouped GEMM launch per projection -- no host
        synchronisation (graph-capturable) and no work on tokens an expert was not chosen for.
        After the router's top-k the block is five launches: pair sort, w13 (reading each row's
        token through the sort's index, so the gathered copy of the activations never exists),
        SiLU * mul, w2, and the weighted sum per token."""
        num_tokens, hidden_dim = hidden_states.shape
        (w13, w2) = self._stacked()
        n_local = len(self.expert_indicies)
        routing_weights, selected_experts = self.route(hidden_states)
        selected_experts = selected_experts.contiguous()
        act_order = w13[3] is not None
        pairs = num_tokens * self.top_k
        if (n_local + 1 > 128 or self.top_k > 8 or
                ((pairs + 63) // 64) * (n_local + 1) > ops.MOE_SORT_MAX_COUNTERS):
            return self._forward_grouped_torch(hidden_states, routing_weights, selected_experts)
        offsets, sorted_token, pair_row, row_bucket = ops.moe_sort_pairs(
            selected_experts, self._bucket_of, n_local + 1, want_row_bucket=act_order)
        zo = self._zero_offset
        # a token chooses an expert at most once: no expert holds more rows than there are tokens
        # (a decode batch of up to 16 tokens takes the small-batch kernel on this bound)
        bound = min(pairs, num_tokens)
        if act_order:       # act-ordered experts: each row's columns in its expert's order
            xs = ops.moe_gather_perm(hidden_states, sorted_token, row_bucket, w13[3])
            h = ops.q4_gemm_grouped(xs, w13[0], w13[1], w13[2], offsets, zo, bound)
        else:
            h = ops.q4_gemm_grouped(hidden_states, w13[0], w13[1], w13[2], offsets, zo, bound,
                                    x_rows=sorted_token)
        h = self.experts[self.expert_indicies[0]].act_fn(h)
        if act_order:
            h = ops.moe_gather_perm(h, None, row_bucket, w2[3])
        # rows of other ranks' experts (the trailing bucket) are never read by the combine step
        y = ops.q4_gemm_grouped(h, w2[0], w2[1], w2[2], offsets, zo, bound, zero_outside=False)
        final = ops.moe_combine(y, routing_weights.contiguous(), selected_experts, pair_row,
                                self._bucket_of, n_local)
        return tensor_model_parallel_all_reduce(final).view(num_tokens, hidden_dim)

    def _forward_grouped_torch(self, hidden_states, routing_weights, selected_experts):
        """forward_grouped with the routing glue as tensor operations (batches beyond the
        one-workgroup pair sort)."""
        num_tokens, hidden_dim = hidden_states.shape
        (w13, w2) = self._stacked()
        n_local = len(self.expert_indicies)
        local = self._local_of[selected_experts.reshape(-1)]             # [T * top_k]
        order = torch.argsort(local, stable=True)
        # (bincount would synchronise to size its result)
        counts = (local.unsqueeze(1) == torch.arange(n_local, device=local.device)).sum(0)
        offsets = torch.zeros(n_local + 1, dtype=torch.int32, device=local.device)
        offsets[1:] = torch.cumsum(counts, 0)
        token = torch.div(order, self.top_k, rounding_mode="floor")
        xs = hidden_states.index_select(0, token)
        rows = xs.shape[0]
        zo = self._zero_offset
        row_expert = local[order]
        if w13[3] is not None:      # act-ordered experts: each row's columns in its expert's order
            xs = ops.moe_gather_perm(xs, None, row_expert.to(torch.int32).contiguous(), w13[3])
        bound = min(rows, num_tokens)       # a token chooses an expert at most once
        h = ops.q4_gemm_grouped(xs, w13[0], w13[1], w13[2], offsets, zo, bound)
        h = self.experts[self.expert_indicies[0]].act_fn(h)
        if w2[3] is not None:
            h = ops.moe_gather_perm(h, None, row_expert.to(torch.int32).contiguous(), w2[3])
        y = ops.q4_gemm_grouped(h, w2[0], w2[1], w2[2], offsets, zo, bound)
        # fp32 product, then the rounding to the model dtype, as two steps: written as one
        # in-place mixed multiply the compiler may fuse them into a single rounding
        y = (y.float() * routing_weights.reshape(-1)[order].unsqueeze(1)).to(y.dtype)
        final = torch.zeros_like(hidden_states)
        final.index_add_(0, token, y)      # rows of other ranks' experts are zero
        return tensor_model_parallel_all_reduce(final).view(num_tokens, hidden_dim)

    def forward_routed(self, hidden_states: torch.Tensor) -> torch.Tensor:
        num_tokens, hidden_dim = hidden_states.shape
        routing_weights, selected_experts = self.route(hidden_states)
        final = torch.zeros_like(hidden_states)
        for expert_idx in self.expert_indicies:
            mask = selected_experts == expert_idx                       # [T, top_k]
            tok = mask.any(dim=-1).nonzero(as_tuple=True)[0]            # routed tokens, ascending
            if tok.numel() == 0:
                continue
            w = (routing_weights * mask).sum(dim=-1, keepdim=True)[tok]  # fp32 [n, 1]
            out = self.experts[expert_idx](hidden_states[tok])
            out.mul_(w)                                                 # fp32 product, fp16 store
            final.index_add_(0, tok, out)
        return tensor_model_parallel_all_reduce(final).view(num_tokens, hidden_dim)

    def forward_dense(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """The reference's execution order (every expert over every token), for comparison."""
        routing_weights, selected_experts = self.route(hidden_states)
        final = None
        for expert_idx in self.expert_indicies:
            mask = selected_experts == expert_idx
            w = (routing_weights * mask).sum(dim=-1, keepdim=True)
            cur = self.experts[expert_idx](hidden_states).mul_(w)
            final = cur if final is None else final.add_(cur)
        return tensor_model_parallel_all_reduce(final)
