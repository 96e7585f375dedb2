"""The ctypes RCCL communicator (the role of the reference's PyNcclCommunicator,
vllm/distributed/device_communicators/pynccl.py:18-185) on one GPU: what can be shown here is a
one-rank communicator -- ncclCommInitRank, ncclAllReduce eagerly and from a captured graph (the
reason the communicator exists), the change_state switch -- and the failure paths.  RCCL refuses
two ranks on one device, so sums over several ranks need a multi-GPU node
(tests/distributed/test_pynccl.py in the reference needs two GPUs for the same reason)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm():
    from vllm_legacy_amd.distributed.rccl_communicator import RcclCommunicator
    c = RcclCommunicator(group=None, device=torch.device("cuda", 0), allow_single_rank=True)
    assert c.available, c.reason
    yield c
    c.close()


def test_one_rank_all_reduce_eager_and_disabled_by_default(comm):
    x = torch.arange(4096, dtype=torch.float16, device="cuda")
    want = x.clone()
    assert comm.disabled                      # off until a caller switches it on
    comm.all_reduce(x)                        # no-op while disabled
    with comm.change_state(True, torch.cuda.current_stream()):
        assert not comm.disabled
        comm.all_reduce(x)
    torch.cuda.synchronize()
    assert comm.disabled and torch.equal(x, want)      # a sum over one rank
    assert comm.version.count(".") == 2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_one_rank_all_reduce_inside_a_captured_graph(comm, dtype):
    """The decode step's all-reduce as it is issued under capture: ncclAllReduce on the capturing
    stream, replayed with fresh data."""
    x = torch.zeros((256, 4096), dtype=dtype, device="cuda")
    y = torch.empty_like(x)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with comm.change_state(True, s):
            comm.all_reduce(x)                # warm-up outside the capture
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            x.mul_(2)
            with comm.change_state(True, torch.cuda.current_stream()):
                comm.all_reduce(x)
            y.copy_(x).add_(1)
    for i in range(3):
        x.fill_(float(i + 1))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, torch.full_like(y, 2.0 * (i + 1) + 1))


def test_unavailable_paths():
    from vllm_legacy_amd.distributed.rccl_communicator import RcclCommunicator
    c = RcclCommunicator(group=None, device=torch.device("cuda", 0))          # one rank: not needed
    assert not c.available and c.disabled and c.reason == "single rank"
    c.all_reduce(torch.ones(8, device="cuda"))                                 # harmless no-op
    bad = RcclCommunicator(group=None, device=torch.device("cuda", 0), allow_single_rank=True,
                           library_path="/nonexistent/librccl.so")
    assert not bad.available and "OSError" in bad.reason
