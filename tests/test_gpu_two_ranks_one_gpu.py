"""-m gpu: the multi-rank path executed by TWO PROCESSES - on the one GPU a box has.  RCCL refuses two ranks on one device,
so the two ranks talk over gloo with their device tensors staged through the host (a2c_agent.py::_collectives); the kernels of the path -
factor packing, gradient rebuild from the gathered factors of both ranks, clip + Adam, the flat-gradient form for large minibatches - and the
host logic - rank-sharded seeds, parameter broadcast, one collective per optimiser step - run as they do in the driver's N > 1 launches.
(One GPU is shared by two processes, so the forward / backward launch takes its multi-kernel form, SDXP_STEP_IMPL=kernels: two persistent
256-workgroup launches of two processes cannot both be resident.  The persistent form is covered at world size 1,
tests/test_gpu_fullsize_properties.py.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("minibatch", [4, 32], ids=["factor all-gather (shipped minibatch 4)", "gradient all-reduce (minibatch 32)"])
def test_two_processes_step_together(tmp_path, minibatch):
    n, epochs, world = 16, 1, 2          # (two processes on one GPU take turns: 0.2 s per optimiser step; 160 / 20 steps)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SDXP_STEP_IMPL="kernels", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "two_ranks_one_gpu.py"), str(tmp_path), str(n),
                                       str(minibatch), str(epochs)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420)[0].decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, o[-3000:])
    a, b = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world))
    steps = epochs * 5 * (n * 8 // minibatch)
    assert int(a["ac_t"]) == int(b["ac_t"]) == steps                                   # both ranks took every optimiser step
    assert int(a["factor_path"]) == (1 if minibatch <= 8 else 0)
    for k in ("ac0", "cv0"):
        np.testing.assert_array_equal(a[k], b[k])                                      # parameters broadcast from rank 0 ...
    assert not np.array_equal(a["obs"], b["obs"])                                      # ... the ranks' envs are seeded apart (seed + rank) ...
    for k in ("ac", "cv"):
        assert np.isfinite(a[k]).all()
        np.testing.assert_array_equal(a[k], b[k])                                      # ... and every step was taken on the SAME averaged gradient:
        assert np.abs(a[k] - a[k + "0"]).max() > 0                                     # the replicas are bit-identical after 160 / 20 optimiser steps, and moved
    assert float(a["lr"]) == float(b["lr"])
