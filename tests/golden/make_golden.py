"""Generate tests/golden/*.npz from the REFERENCE's own CPU kernels.

Runs only in the build container (needs /root/reference and an AVX-512 host): it
loads oracle/_ref/vllm_ref_cpu.so -- the reference's csrc/cpu/attention.cpp and
csrc/cpu/cache.cpp compiled in place by oracle/build_ref.py -- and the reference's
host-side FNUZ fp8 codec (oracle/_ref/libref_fp8.so), feeds them seeded inputs and
stores inputs + outputs.  The .npz files are data only; they travel to the GPU box,
the reference does not.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.float32:
        return t.numpy().copy()
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def attention_case(ref, name, dtype, num_seqs_lens, hq, hkv, d, nb, alibi, seed):
    g = torch.Generator().manual_seed(seed)
    bs = 16
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    scale = float(d**-0.5)
    seq_lens = torch.tensor(num_seqs_lens, dtype=torch.int32)
    s = len(num_seqs_lens)
    max_len = int(seq_lens.max())
    mb = (max_len + bs - 1) // bs
    q = (torch.rand(s, hq, d, generator=g) * 2 - 1).mul(scale).to(dtype)
    kc = (torch.rand(nb, hkv, d // x, bs, x, generator=g) * 2 - 1).mul(scale).to(dtype)
    vc = (torch.rand(nb, hkv, d, bs, generator=g) * 2 - 1).mul(scale).to(dtype)
    bt = torch.randint(0, nb, (s, mb), generator=g, dtype=torch.int32)
    al = torch.randn(hq, generator=g, dtype=torch.float32) if alibi else None
    out1 = torch.zeros_like(q)
    ref.paged_attention_v1(out1, q, kc, vc, hkv, scale, bt, seq_lens, bs, max_len, al, "auto",
                           1.0, 0, 0, 0, 64, 0)
    p = (max_len + 511) // 512
    out2 = torch.zeros_like(q)
    es = torch.zeros(s, hq, p, dtype=torch.float32)
    ml = torch.zeros(s, hq, p, dtype=torch.float32)
    tmp = torch.zeros(s, hq, p, d, dtype=dtype)
    ref.paged_attention_v2(out2, es, ml, tmp, q, kc, vc, hkv, scale, bt, seq_lens, bs, max_len,
                           al, "auto", 1.0, 0, 0, 0, 64, 0)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), dtype=str(dtype).replace("torch.", ""), scale=scale,
        query=bits(q), key_cache=bits(kc), value_cache=bits(vc), block_tables=bt.numpy(),
        seq_lens=seq_lens.numpy(), alibi=(al.numpy() if alibi else np.zeros(0, np.float32)),
        out_v1=bits(out1), out_v2=bits(out2))
    print(name, "v1/v2 max diff", (out1.float() - out2.float()).abs().max().item())


def cache_case(ref, name, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    t, h, d, bs, nb = 11, 3, 32, 16, 9
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    qkv = torch.randn(t, 3, h, d, generator=g).to(dtype)
    key, value = qkv[:, 1], qkv[:, 2]  # strided views, as in test_cache.py:137-138
    kc = torch.randn(nb, h, d // x, bs, x, generator=g).to(dtype)
    vc = torch.randn(nb, h, d, bs, generator=g).to(dtype)
    slots = torch.randperm(nb * bs, generator=g)[:t].to(torch.int64)
    slots[4] = -1  # a padding token
    kc0, vc0 = kc.clone(), vc.clone()
    ref.reshape_and_cache(key, value, kc, vc, slots, "auto", 1.0)
    # copy_blocks on the updated caches: each source to two destinations
    pairs = torch.tensor([[0, 5], [0, 6], [2, 7], [2, 8]], dtype=torch.int64)
    kc2, vc2 = kc.clone(), vc.clone()
    ref.copy_blocks([kc2], [vc2], pairs)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), dtype=str(dtype).replace("torch.", ""),
        key=bits(key.contiguous()), value=bits(value.contiguous()), key_cache_in=bits(kc0),
        value_cache_in=bits(vc0), slot_mapping=slots.numpy(), key_cache_out=bits(kc),
        value_cache_out=bits(vc), block_mapping=pairs.numpy(), key_cache_copied=bits(kc2),
        value_cache_copied=bits(vc2))
    print(name, "done")


def glue_case(ref, name, dtype, seed):
    """RMSNorm, fused add+RMSNorm, rotary embedding (NeoX and GPT-J, partial rotation) and
    SiLU-and-mul through the reference's CPU twins (csrc/cpu/layernorm.cpp, pos_encoding.cpp,
    activation.cpp); shapes after tests/kernels/test_layernorm.py, test_pos_encoding.py and
    test_activation.py, small."""
    g = torch.Generator().manual_seed(seed)
    t, hidden, eps = 7, 768, 1e-6
    x = (torch.randn(t, hidden, generator=g) * 0.5).to(dtype)
    res = (torch.randn(t, hidden, generator=g) * 0.5).to(dtype)
    w = (1.0 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    out = torch.empty_like(x)
    ref.rms_norm(out, x, w, eps)
    fx, fr = x.clone(), res.clone()
    ref.fused_add_rms_norm(fx, fr, w, eps)
    # rotary: head size 64, rotating 64 (NeoX) and 32 (GPT-J) dims
    hq, hkv, d, maxpos = 4, 2, 64, 128
    pos = torch.randint(0, maxpos, (t, ), generator=g, dtype=torch.int64)
    rec = {}
    for tag, rot, neox in (("neox", 64, True), ("gptj", 32, False)):
        inv = 1.0 / (10000.0**(torch.arange(0, rot, 2, dtype=torch.float32) / rot))
        fr_ = torch.outer(torch.arange(maxpos, dtype=torch.float32), inv)
        cache = torch.cat([fr_.cos(), fr_.sin()], dim=-1).to(dtype)
        q = torch.randn(t, hq * d, generator=g).to(dtype)
        k = torch.randn(t, hkv * d, generator=g).to(dtype)
        q2, k2 = q.clone(), k.clone()
        ref.rotary_embedding(pos, q2, k2, d, cache, neox)
        rec.update({f"rot_{tag}_cache": bits(cache), f"rot_{tag}_q": bits(q), f"rot_{tag}_k": bits(k),
                    f"rot_{tag}_q_out": bits(q2), f"rot_{tag}_k_out": bits(k2)})
    a = torch.randn(t, 2 * 512, generator=g).to(dtype)
    act = torch.empty(t, 512, dtype=dtype)
    ref.silu_and_mul(act, a)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), dtype=str(dtype).replace("torch.", ""), eps=eps,
        x=bits(x), residual=bits(res), weight=bits(w), rms_out=bits(out), fused_out=bits(fx),
        fused_residual=bits(fr), positions=pos.numpy(), head_size=d, act_in=bits(a),
        act_out=bits(act), **rec)
    print(name, "done")


def fnuz_case():
    lib = C.CDLL(build_ref.fp8_so_path())
    lib.ref_fnuz_decode_f32_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_fnuz_encode_f32_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    codes = np.arange(256, dtype=np.uint8)
    dec = np.empty(256, dtype=np.float32)
    lib.ref_fnuz_decode_f32_n(codes.ctypes.data, dec.ctypes.data, 256)
    # every finite fp16 value (as float) + edge floats
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    h = h[np.isfinite(h)]
    rng = np.random.default_rng(0)
    extra = np.concatenate([
        rng.uniform(-260, 260, 4096).astype(np.float32),
        (rng.standard_normal(4096) * 0.01).astype(np.float32),
        np.array([0.0, -0.0, 239.9, 240.0, 244.0, 247.9, 248.0, 1e9, -1e9, np.inf, -np.inf,
                  np.nan, 2.0**-10, 2.0**-11, 1.5 * 2.0**-11, 2.0**-12], dtype=np.float32)
    ])
    x = np.concatenate([h, extra])
    enc = np.empty(x.size, dtype=np.uint8)
    lib.ref_fnuz_encode_f32_n(x.ctypes.data, enc.ctypes.data, x.size)
    np.savez_compressed(os.path.join(OUT, "fp8_fnuz_codec.npz"), decode_table=dec, encode_in=x,
                        encode_out=enc)
    print("fnuz codec table done", x.size)


def greedy_loop_case(ref, name, seed, prompt_len=37, steps=16):
    """Configuration 1 as a loop: one sequence, OPT-125M's attention shape (12 heads of 64, bf16,
    block 16), a prompt followed by `steps` greedy decode steps in which step t's cache write
    feeds step t + 1's attention.  The attention side is the reference's CPU executor path
    (vllm/attention/backends/torch_sdpa.py:159-228): every step writes its key / value with the
    reference's reshape_and_cache (csrc/cpu/cache.cpp); the prompt runs torch's
    scaled_dot_product_attention (is_causal, :199-206), each decode step the reference's
    paged_attention_v1 (csrc/cpu/attention.cpp:421-438) over everything cached so far.  Around
    it sits a one-layer stand-in for the model, just enough to close the greedy loop: token ->
    embedding row -> q | k | v by one projection -> attention -> logits by one projection ->
    argmax -> next token (bf16 tensors, torch matmuls on the CPU).  Stored per step: the token,
    q / k / v as they went into the reference attention, its output, and the argmax."""
    g = torch.Generator().manual_seed(seed)
    dt = torch.bfloat16
    heads, d, bs, hidden, vocab = 12, 64, 16, 768, 512
    scale = float(d**-0.5)
    x = 16 // 2
    nb = 8
    embed = (torch.randn(vocab, hidden, generator=g) * 1.0).to(dt)
    w_qkv = (torch.randn(hidden, 3 * hidden, generator=g) * hidden**-0.5).to(dt)
    w_out = (torch.randn(hidden, vocab, generator=g) * hidden**-0.5).to(dt)
    prompt = torch.randint(0, vocab, (prompt_len, ), generator=g)
    blocks = torch.randperm(nb, generator=g)[:(prompt_len + steps + bs - 1) // bs].to(torch.int32)
    kc = torch.zeros(nb, heads, d // x, bs, x, dtype=dt)
    vc = torch.zeros(nb, heads, d, bs, dtype=dt)
    slot = lambda pos: int(blocks[pos // bs]) * bs + pos % bs          # noqa: E731

    def qkv_of(tokens):
        h = embed[tokens]
        qkv = (h.float() @ w_qkv.float()).to(dt)
        q, k, v = qkv.split(hidden, dim=-1)
        return q.contiguous(), k.contiguous(), v.contiguous()

    def next_token(out_row):
        return int(torch.argmax(out_row.float() @ w_out.float()))

    # prompt (torch_sdpa.py:159-208)
    q, k, v = qkv_of(prompt)
    slots = torch.tensor([slot(i) for i in range(prompt_len)], dtype=torch.int64)
    ref.reshape_and_cache(k.view(-1, heads, d), v.view(-1, heads, d), kc, vc, slots, "auto", 1.0)
    qh, kh, vh = (t.view(-1, heads, d).movedim(0, 1) for t in (q, k, v))
    out = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, attn_mask=None, dropout_p=0.0,
                                                           is_causal=True, scale=scale)
    out = out.movedim(1, 0).reshape(prompt_len, hidden).contiguous()
    rec = dict(prompt=prompt.numpy(), prompt_q=bits(q), prompt_k=bits(k), prompt_v=bits(v),
               prompt_out=bits(out), prompt_slots=slots.numpy(), block_table=blocks.numpy(),
               scale=scale, embed=bits(embed), w_qkv=bits(w_qkv), w_out=bits(w_out))
    tok = next_token(out[-1])
    toks, qs, ks, vs, outs = [], [], [], [], []
    for t in range(steps):
        pos = prompt_len + t
        toks.append(tok)
        q, k, v = qkv_of(torch.tensor([tok]))
        ref.reshape_and_cache(k.view(1, heads, d), v.view(1, heads, d), kc, vc,
                              torch.tensor([slot(pos)], dtype=torch.int64), "auto", 1.0)
        o = torch.zeros(1, heads, d, dtype=dt)
        ref.paged_attention_v1(o, q.view(1, heads, d), kc, vc, heads, scale, blocks.view(1, -1),
                               torch.tensor([pos + 1], dtype=torch.int32), bs, pos + 1, None, "auto",
                               1.0, 0, 0, 0, 64, 0)
        qs.append(bits(q)); ks.append(bits(k)); vs.append(bits(v)); outs.append(bits(o.view(1, hidden)))
        tok = next_token(o.view(hidden))
    toks.append(tok)
    rec.update(tokens=np.array(toks, dtype=np.int64), step_q=np.concatenate(qs), step_k=np.concatenate(ks),
               step_v=np.concatenate(vs), step_out=np.concatenate(outs), key_cache_final=bits(kc),
               value_cache_final=bits(vc))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "tokens", toks)


def main():
    assert build_ref.build(), "reference tree not available"
    ref = build_ref.load_ref()
    assert ref is not None, "reference CPU kernels need AVX-512 on this host"
    torch.manual_seed(0)
    greedy_loop_case(ref, "loop_bf16_opt125m", 21)
    if "--loop-only" in sys.argv:
        return
    # OPT-125M on the reference's CPU executor (12 heads of 64, one sequence, bf16, block 16:
    # vllm/attention/backends/torch_sdpa.py:216-228 -> csrc/cpu/attention.cpp:421-438) and the
    # per-rank head shape of Llama-3-70B under 8-way tensor parallelism (8 query heads, 1 KV head)
    attention_case(ref, "attn_bf16_opt125m", torch.bfloat16, [300], 12, 12, 64, 24, False, 8)
    attention_case(ref, "attn_bf16_llama70b_tp8", torch.bfloat16, [5, 1290, 300], 8, 1, 128, 96,
                   False, 9)
    if "--new-only" in sys.argv:
        return
    attention_case(ref, "attn_f32_gqa", torch.float32, [1, 37, 600], 8, 2, 64, 16, False, 1)
    attention_case(ref, "attn_bf16_alibi", torch.bfloat16, [17, 513, 1030], 4, 4, 128, 16, True,
                   2)
    attention_case(ref, "attn_f32_d80", torch.float32, [16, 95], 6, 3, 80, 8, False, 3)
    cache_case(ref, "cache_f32", torch.float32, 4)
    cache_case(ref, "cache_bf16", torch.bfloat16, 5)
    glue_case(ref, "glue_f32", torch.float32, 6)
    glue_case(ref, "glue_bf16", torch.bfloat16, 7)
    fnuz_case()


if __name__ == "__main__":
    main()
