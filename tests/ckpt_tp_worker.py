"""One tensor-parallel rank of the checkpoint test (started by test_gpu_checkpoint.py; every rank on
cuda:0, Gloo process groups): loads its shard of a safetensors checkpoint, runs one decode step and
rank 0 saves the final hidden state and the gathered logits."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ckpt, inputs, out_path, method, kv = sys.argv[4:9]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from tests import ckpt_lib, e2e_gpu
    from tests import pa_testlib as T
    from vllm_legacy_amd.distributed import parallel_state as ps
    from vllm_legacy_amd.model.llama_decode import LlamaConfig, LlamaDecodeModel
    ps.initialize_model_parallel(world, backend="gloo")
    d = np.load(inputs)
    cfg = LlamaConfig(**{k: (float(v) if k in ("rms_norm_eps", "rope_theta") else int(v))
                         for k, v in zip(d["cfg_keys"].tolist(), d["cfg_vals"].tolist())})
    model = LlamaDecodeModel(cfg, method, kv, device=dev, seed=3)
    model.load_weights(ckpt_lib.read_safetensors(ckpt), method)
    hkv_l = model.layers[0].hkv
    # this rank's KV heads of the full caches
    kv_caches = []
    for li in range(cfg.num_hidden_layers):
        full = torch.from_numpy(d[f"cache{li}"]).to(dev)             # [2, NB, BS * Hkv * D] storage
        nb = full.shape[1]
        per_head = full.view(2, nb, cfg.num_key_value_heads, -1)
        mine = per_head[:, :, rank * hkv_l:(rank + 1) * hkv_l].contiguous().view(2, nb, -1)
        kv_caches.append(mine if kv == "fp8" else mine.view(torch.float16))
    meta = e2e_gpu.make_meta(dev, d["seq_lens"], d["block_tables"], d["slot_mapping"])
    with torch.inference_mode():
        hidden = model(torch.from_numpy(d["input_ids"]).to(dev), torch.from_numpy(d["positions"]).to(dev),
                       kv_caches, meta)
        logits = model.compute_logits(hidden)
    torch.cuda.synchronize()
    # every rank leaves its cache shard as written (the parent reassembles the heads)
    np.savez(f"{out_path}.rank{rank}.npz", **{f"cache{li}": T.bits(c.view(torch.uint8) if kv == "fp8" else c)
                                             for li, c in enumerate(kv_caches)})
    if rank == 0:
        np.savez(out_path, hidden=T.bits(hidden), logits=logits.float().cpu().numpy())
    dist.barrier()
    ps.destroy_model_parallel()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
