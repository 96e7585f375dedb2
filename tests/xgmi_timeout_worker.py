"""One rank of the barrier-timeout test of the peer-memory all-reduce (two processes on cuda:0,
started by test_gpu_xgmi_allreduce.py with VLA_XGMI_WAIT_SECONDS=0.5).  Rank 1 is late with one
call: rank 0's wait must end after the time bound, its output must be NaN (not a partial sum),
the error must be sticky and visible through both the synchronous and the asynchronous read.
Rank 1 then issues its late call: by then its flags are set and it passes both barriers without
waiting, gathering slices that rank 0 reduced without it -- that must not pass for a sum either:
rank 0's timeout was counted in rank 1's region too, so rank 1's output is NaN and its host sees
the error, and the collective check of communication_op reports the failure on BOTH ranks."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_legacy_amd.distributed.xgmi_all_reduce import XgmiAllReduce  # noqa: E402


def main():
    rank, port = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = XgmiAllReduce(dist.group.WORLD, dev, max_size=1 << 20)
    assert not comm.disabled
    rec = {"rank": rank, "self_check": comm.self_check()}
    x = torch.full((4096, ), float(rank + 1), dtype=torch.float16, device=dev)
    good = comm.all_reduce(x)
    comm.poll_error_async()
    torch.cuda.synchronize()
    rec["good_sum_ok"] = bool((good == 3.0).all().item())
    rec["polled_before"] = comm.polled_error()
    dist.barrier()
    if rank == 0:
        t0 = time.perf_counter()
        lost = comm.all_reduce(x)                  # rank 1 never issues this call
        comm.poll_error_async()
        torch.cuda.synchronize()
        rec["wait_s"] = round(time.perf_counter() - t0, 2)
        rec["lost_all_nan"] = bool(torch.isnan(lost).all().item())
        rec["polled_after"] = comm.polled_error()
        rec["errors"] = comm.error_count()
        t0 = time.perf_counter()
        later = comm.all_reduce(x)                 # sticky: NaN again, and no more waiting
        res = torch.zeros((8, 512), dtype=torch.float16, device=dev)
        normed = comm.all_reduce_add_rms_norm(torch.ones_like(res), res,
                                              torch.ones(512, dtype=torch.float16, device=dev), 1e-5)
        torch.cuda.synchronize()
        rec["later_s"] = round(time.perf_counter() - t0, 2)
        rec["later_all_nan"] = bool(torch.isnan(later).all().item() and torch.isnan(normed).all().item()
                                    and torch.isnan(res).all().item())
    dist.barrier()                                 # rank 0 has given up by now
    if rank == 1:
        t0 = time.perf_counter()
        late = comm.all_reduce(x)                  # the call rank 0 waited for in vain
        comm.poll_error_async()
        torch.cuda.synchronize()
        rec["late_s"] = round(time.perf_counter() - t0, 2)
        rec["late_all_nan"] = bool(torch.isnan(late).all().item())
        rec["late_polled"] = comm.polled_error()
        rec["late_errors"] = comm.error_count()
    # what a decode loop does after its step: one collective verdict (it closes the communicator
    # on every rank when any rank saw an error)
    from vllm_legacy_amd.distributed import communication_op as co
    from vllm_legacy_amd.distributed import parallel_state as ps
    ps._TP_GROUP = ps._TP_CPU_GROUP = dist.group.WORLD
    ps.set_tp_xgmi_all_reduce(comm)
    co.tensor_model_parallel_poll_all_reduce_error(dev)
    torch.cuda.synchronize()
    rec["collective_check_ok"] = co.tensor_model_parallel_check_all_reduce(dev)
    rec["communicator_removed"] = ps.get_tp_xgmi_all_reduce(dev) is None
    ps._TP_GROUP = ps._TP_CPU_GROUP = None
    dist.barrier()
    print(json.dumps(rec), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
