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
