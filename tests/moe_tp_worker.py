"""One tensor-parallel rank of the MoE test (started by test_gpu_moe.py; every rank on cuda:0 with
Gloo collectives, which the communication layer stages through the host)."""
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
    from vllm_legacy_amd.distributed import parallel_state as ps
    ps.initialize_model_parallel(world, backend="gloo")
    from vllm_legacy_amd.model.mixtral_moe import QuantMixtralMoE
    moe = QuantMixtralMoE(512, 1024, 8, 2, "awq", 128, "cuda", seed=1)
    for idx in moe.expert_indicies:
        for lin in (moe.experts[idx].w13, moe.experts[idx].w2):
            lin.scales.data = (lin.scales.data.float() * 0.3).half()
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.empty((96, 512), device="cuda").normal_(0, 1, generator=g).half()
    with torch.inference_mode():
        grouped = moe.forward_grouped(x)
        dense = moe.forward_dense(x)
    torch.cuda.synchronize()
    xg = ps.get_tp_xgmi_all_reduce(torch.device("cuda", 0))
    if os.environ.get("VLA_XGMI_ALLREDUCE") == "1":
        assert xg is not None and not xg.disabled and xg.error_count() == 0
    else:
        assert xg is None
    if rank == 0:
        np.savez(out_path, grouped=grouped.float().cpu().numpy(), dense=dense.float().cpu().numpy(),
                 local=np.array(moe.expert_indicies))
    dist.barrier()
    ps.destroy_model_parallel()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
