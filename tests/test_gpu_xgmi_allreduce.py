"""Peer-memory all-reduce: N processes on one GPU (the GPU box has one), each mapping the others'
regions over HIP IPC.  Results must equal, bit for bit, the fp32 rank-order sum rounded once --
which also makes them identical on every rank.  The reference's own test of its custom
all-reduce compares against NCCL with allclose over graph replays and eager calls
(tests/distributed/test_custom_all_reduce.py:24-80); the cases here follow it (eager sizes,
captured graph replayed with fresh data)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_xgmi_all_reduce_processes_on_one_gpu(world):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "xgmi_worker.py"), str(r), str(world),
                               str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=100)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, out, err in outs:
        assert rc == 0, f"rank failed rc={rc}\nstdout: {out}\nstderr: {err[-3000:]}"
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["ok"] and rec["timeouts"] == 0
        print(rec)


def test_barrier_timeout_is_bounded_sticky_and_poisons_the_output():
    """A peer that never arrives: the wait ends after the time bound (wall clock, not a poll
    count), the output is NaN instead of a partial sum, every later call on that rank is NaN
    too, and the host sees the error through the asynchronous 4-byte read.  The reference's
    custom all-reduce would spin forever here (csrc/custom_all_reduce.cuh:180-205)."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VLA_XGMI_WAIT_SECONDS="0.5")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "xgmi_timeout_worker.py"), str(r),
                               str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
             for r in range(2)]
    recs = {}
    try:
        for p in procs:
            out, err = p.communicate(timeout=100)
            assert p.returncode == 0, f"rank failed rc={p.returncode}\nstdout: {out}\nstderr: {err[-3000:]}"
            rec = json.loads(out.strip().splitlines()[-1])
            recs[rec["rank"]] = rec
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r in (0, 1):
        assert recs[r]["self_check"] and recs[r]["good_sum_ok"] and recs[r]["polled_before"] == 0
    r0 = recs[0]
    assert r0["lost_all_nan"] and r0["later_all_nan"], r0
    assert r0["errors"] > 0 and r0["polled_after"] > 0, r0
    assert 0.4 <= r0["wait_s"] <= 5.0, r0      # one bounded wait, then no more waiting
    assert r0["later_s"] <= 1.0, r0
    # the late rank: no wait, no finite wrong sum, and its host knows
    r1 = recs[1]
    assert r1["late_all_nan"] and r1["late_errors"] > 0 and r1["late_polled"] > 0, r1
    assert r1["late_s"] <= 1.0, r1
    for r in (0, 1):
        assert recs[r]["collective_check_ok"] is False and recs[r]["communicator_removed"], recs[r]
