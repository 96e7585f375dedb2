"""CPU tests of the tensor-parallel path: two gloo ranks exercise the collectives the decode
step issues (vllm/distributed/communication_op.py:73-160, :227-317) and check that Megatron-style
sharding (column-parallel then row-parallel + all-reduce; vocab-parallel embedding; logits
gather) reproduces the unsharded result.  The reference tests these on real GPUs only
(tests/distributed/test_comm_ops.py:19-116); the arithmetic of the collectives is the same."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vllm_legacy_amd.distributed import parallel_state as ps
    ps.initialize_model_parallel(world, backend="gloo")
    try:
        globals()[fn_name](rank, world)
        ret[rank] = "ok"
    except Exception as e:  # surfaced by the parent
        import traceback
        ret[rank] = traceback.format_exc() + str(e)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert not p.is_alive(), "rank hung"
    for r in range(world):
        assert ret.get(r) == "ok", ret.get(r)


def _collectives(rank, world):
    from vllm_legacy_amd.distributed import communication_op as co
    from vllm_legacy_amd.distributed import parallel_state as ps
    assert ps.get_tensor_model_parallel_world_size() == world
    assert ps.get_tensor_model_parallel_rank() == rank
    # all-reduce of arange(8) * (r + 1), as the reference's test does (test_comm_ops.py:30-38)
    t = torch.arange(8, dtype=torch.float32) * (rank + 1)
    out = co.tensor_model_parallel_all_reduce(t)
    assert out is t
    assert torch.equal(t, torch.arange(8, dtype=torch.float32) * sum(range(1, world + 1)))
    # all-gather along the last and the first dim
    x = torch.full((2, 3), float(rank))
    g = co.tensor_model_parallel_all_gather(x, dim=-1)
    assert g.shape == (2, 3 * world) and torch.equal(g[:, 3:], torch.full((2, 3), 1.0))
    g0 = co.tensor_model_parallel_all_gather(x, dim=0)
    assert g0.shape == (2 * world, 3) and torch.equal(g0[2:], torch.full((2, 3), 1.0))
    # gather to rank 0
    gg = co.tensor_model_parallel_gather(x, dst=0, dim=-1)
    if rank == 0:
        assert torch.equal(gg, g)
    else:
        assert gg is None
    # metadata broadcast: python values + tensors
    d = None
    if rank == 0:
        d = {"block_tables": torch.arange(12, dtype=torch.int32).view(3, 4),
             "slot_mapping": torch.tensor([5, 17, 33], dtype=torch.int64),
             "max_decode_seq_len": 4096, "empty": torch.empty(0, dtype=torch.int32), "name": "x"}
    got = co.broadcast_tensor_dict(d, src=0)
    assert got["max_decode_seq_len"] == 4096 and got["name"] == "x"
    assert torch.equal(got["block_tables"], torch.arange(12, dtype=torch.int32).view(3, 4))
    assert got["slot_mapping"].dtype == torch.int64 and got["empty"].numel() == 0


def _sharded_mlp_matches_full(rank, world):
    from vllm_legacy_amd.distributed import communication_op as co
    g = torch.Generator().manual_seed(0)  # same on every rank
    T_, H, F, V = 5, 32, 64, 40
    x = torch.randn(T_, H, generator=g)
    w_up = torch.randn(H, F, generator=g) * 0.1     # column-parallel: shard the output dim
    w_down = torch.randn(F, H, generator=g) * 0.1   # row-parallel: shard the input dim
    full = torch.relu(x @ w_up) @ w_down
    fs = F // world
    part = torch.relu(x @ w_up[:, rank * fs:(rank + 1) * fs]) @ w_down[rank * fs:(rank + 1) * fs]
    got = co.tensor_model_parallel_all_reduce(part)
    assert torch.allclose(got, full, atol=1e-5)
    # vocab-parallel embedding (mask + all-reduce) and logits gather
    emb = torch.randn(V, H, generator=g)
    ids = torch.tensor([0, 19, 20, 39, 7])
    vs = V // world
    local = ids - rank * vs
    mask = (local < 0) | (local >= vs)
    e = emb[rank * vs:(rank + 1) * vs][local.masked_fill(mask, 0)]
    e.masked_fill_(mask[:, None], 0)
    assert torch.equal(co.tensor_model_parallel_all_reduce(e), emb[ids])
    head = torch.randn(H, V, generator=g)
    logits = co.tensor_model_parallel_gather(x @ head[:, rank * vs:(rank + 1) * vs], dst=0, dim=-1)
    if rank == 0:
        assert torch.allclose(logits, x @ head, atol=1e-5)


def _head_sharding_matches_full_attention(rank, world):
    """Attention shards by KV head with no exchange: the concatenation of per-rank outputs over
    each rank's heads equals the unsharded output (oracle used as the attention)."""
    import numpy as np
    from oracle import oracle as O
    from tests import pa_testlib as T
    c = T.make_attention_case(0, 3, (8, 4), 64, 16, O.DT_F32, False, False, max_seq_len=90,
                              num_blocks=32, device=None)
    full = O.paged_attention_v1(c.q, c.kc, c.vc, c.block_tables, c.seq_lens, c.scale, c.dt)
    hq, hkv = 8 // world, 4 // world
    q = np.ascontiguousarray(c.q[:, rank * hq:(rank + 1) * hq])
    kc = np.ascontiguousarray(c.kc[:, rank * hkv:(rank + 1) * hkv])
    vc = np.ascontiguousarray(c.vc[:, rank * hkv:(rank + 1) * hkv])
    part = O.paged_attention_v1(q, kc, vc, c.block_tables, c.seq_lens, c.scale, c.dt)
    from vllm_legacy_amd.distributed import communication_op as co
    got = co.tensor_model_parallel_all_gather(torch.from_numpy(part), dim=1)
    assert np.array_equal(got.numpy(), full)


def _rccl_one_rank_cannot_load(rank, world):
    """The communicator's constructor is collective: when one rank cannot load the library, every
    rank must come back unavailable (none may wait in the unique-id broadcast for the others)."""
    from vllm_legacy_amd.distributed import parallel_state as ps
    from vllm_legacy_amd.distributed.rccl_communicator import RcclCommunicator
    c = RcclCommunicator(ps.get_tensor_model_parallel_cpu_group(), torch.device("cpu"),
                         library_path="/nonexistent/librccl.so" if rank == 1 else None)
    assert not c.available and c.disabled and c.reason
    if rank == 1:
        assert "OSError" in c.reason
    assert RcclCommunicator.supports(torch.float16) and not RcclCommunicator.supports(torch.complex64)
    c.close()


def test_rccl_communicator_agrees_when_one_rank_cannot_load():
    _run("_rccl_one_rank_cannot_load")


def test_collectives_two_ranks():
    _run("_collectives")


def test_sharded_linear_embedding_logits_two_ranks():
    _run("_sharded_mlp_matches_full")


def test_attention_shards_by_head_two_ranks():
    _run("_head_sharding_matches_full_attention")


def test_single_process_is_a_no_op():
    from vllm_legacy_amd.distributed import communication_op as co
    t = torch.ones(3)
    assert co.tensor_model_parallel_all_reduce(t) is t
    assert co.tensor_model_parallel_all_gather(t) is t
    assert co.tensor_model_parallel_gather(t) is t


def test_rccl_ctypes_communicator_failure_paths_without_gpu():
    """No GPU here: the communicator must report itself unavailable (never raise) for a single
    rank and for a library that cannot be loaded, and the all-reduce dispatcher must say what it
    would use."""
    from vllm_legacy_amd.distributed import communication_op as co
    from vllm_legacy_amd.distributed.rccl_communicator import RcclCommunicator, find_rccl_library
    c = RcclCommunicator(group=None, device=torch.device("cpu"))
    assert not c.available and c.disabled and c.reason == "single rank"
    bad = RcclCommunicator(group=None, device=torch.device("cpu"), allow_single_rank=True,
                           library_path="/nonexistent/librccl.so")
    assert not bad.available and "OSError" in bad.reason
    assert find_rccl_library().endswith(("librccl.so", "librccl.so.1"))
    assert co.all_reduce_implementation(torch.device("cpu"), capturing=True) == "none"
