"""world_size-2 `gloo` test (CPU) of the multi-rank orchestration in seqdex_amd/a2c_agent.py: per optimiser step the
flat gradients and the scalar KL are summed over ranks between sdxp_backward and sdxp_apply, parameters are broadcast
from rank 0 at start, and envs/seed shard as rank-local (seed + rank).  The HIP calls are replaced by a CPU stand-in
with the same method names (the kernels themselves are covered by -m gpu tests)."""
import os
import socket

import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


class FakePPO:
    """CPU stand-in for seqdex_amd.ppo.SdxPPO: 'gradient' of a rank = (rank+1) * (step+1); apply does SGD-like update
    with the averaged gradient so that every rank must end with identical parameters iff the all-reduce happened."""

    def __init__(self, rank, world, fused=False):
        self.rank, self.world, self.step = rank, world, 0
        # fused: the library's layout - both gradients and the KL word in ONE buffer (SDXP_T_ALL_GRADS), views into it
        allg = torch.zeros(10 + 6 + 1)
        self.t = {"AC_GRADS": allg[:10], "CV_GRADS": allg[10:16], "AC_PARAMS": torch.full((10,), float(rank)),
                  "CV_PARAMS": torch.full((6,), float(rank))}
        self._kl = allg[16:17] if fused else torch.zeros(1)
        if fused:
            self.t["ALL_GRADS"] = allg
        self.fused = fused
        self.applied = []

    def kl_view(self):
        return self._kl

    def backward(self, which, mb):
        if mb < 0:
            self.step = 0
            return
        self.t["AC_GRADS"].fill_((self.rank + 1.0) * (self.step + 1))
        self.t["CV_GRADS"].fill_((self.rank + 1.0) * 2.0)
        self._kl.fill_(0.01 * (self.rank + 1))

    def apply(self, which, kl=float("nan")):
        if which == 0:
            assert (kl == float("-inf")) == self.fused      # -inf tells the library to take the KL word of ALL_GRADS
        g = self.t["CV_GRADS" if which else "AC_GRADS"] / self.world
        self.t["CV_PARAMS" if which else "AC_PARAMS"].sub_(0.1 * g)
        if which == 0:
            self.applied.append((float(g[0]), float(self._kl[0]) / self.world))
            self.step += 1


def _worker(rank, world, port, out, fused=False, large=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seqdex_amd.a2c_agent import A2CAgent
    ag = A2CAgent.__new__(A2CAgent)          # orchestration only: no GPU objects
    ag.ppo = FakePPO(rank, world, fused)
    ag.mini_epochs_num, ag.batch_size, ag.minibatch_size = 2, 12, 4
    if large:      # minibatch_size > 8 (GEMM update path): the gradient is the small object - even when the library offers the
        ag.mini_epochs_num, ag.batch_size, ag.minibatch_size = 6, 48, 48       # rank-MB factor tensors they must not be used
        ag.ppo.t["FACTORS"] = torch.zeros(4)
        ag.ppo.t["FACTORS_ALL"] = torch.zeros(world, 4)

        def _no_factors(mb):
            raise AssertionError("factor exchange requested for a large minibatch")
        ag.ppo.backward_factors = _no_factors
    ag.rank, ag.rank_size, ag.multi_gpu = rank, world, True
    ag._broadcast_parameters()
    p0 = ag.ppo.t["AC_PARAMS"].clone()
    ag._update_multi_gpu()
    out.put((rank, p0.tolist(), ag.ppo.t["AC_PARAMS"].tolist(), ag.ppo.t["CV_PARAMS"].tolist(), ag.ppo.applied))
    dist.destroy_process_group()


@pytest.mark.parametrize("fused,large", [(False, False), (True, False), (True, True)])
def test_gradient_allreduce_and_broadcast_world2(fused, large):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, fused, large)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, p0a, a0, c0, ap0), (r1, p0b, a1, c1, ap1) = res
    assert p0a == p0b == [0.0] * 10                      # parameters broadcast from rank 0
    assert a0 == a1 and c0 == c1                         # identical after the all-reduced updates
    # 6 optimiser steps; averaged gradient of step k = (1+2)/2 * (k+1); averaged KL = 0.015
    assert len(ap0) == 6
    for k, (g, kl) in enumerate(ap0):
        assert abs(g - 1.5 * (k + 1)) < 1e-6 and abs(kl - 0.015) < 1e-7
    assert abs(a0[0] - (0.0 - 0.1 * 1.5 * sum(range(1, 7)))) < 1e-5


class FakeFactorPPO:
    """CPU stand-in for the DEFAULT multi-rank branch (minibatch_size <= 8): the library packs this rank's rank-MB factors of the minibatch
    under the device cursor into t["FACTORS"], the caller all-gathers them into t["FACTORS_ALL"] [world, F], apply_factors() rebuilds the
    sum over ranks (ascending rank order) and steps.  Here a rank's factor row is (rank + 1) * (step + 1) in every slot, the 'gradient'
    is the mean over ranks of the first slot, and the last slot carries the KL word."""
    F = 5

    def __init__(self, rank, world):
        self.rank, self.world, self.step = rank, world, 0
        self.t = {"FACTORS": torch.zeros(self.F), "FACTORS_ALL": torch.zeros(world, self.F),
                  "AC_PARAMS": torch.full((10,), float(rank)), "CV_PARAMS": torch.full((6,), float(rank))}
        self.applied, self.calls = [], []

    def backward_factors(self, mb):
        self.calls.append(mb)
        if mb < 0:
            self.step = 0
            return
        self.t["FACTORS"].fill_((self.rank + 1.0) * (self.step + 1))
        self.t["FACTORS"][-1] = 0.01 * (self.rank + 1)

    def apply_factors(self):
        allf = self.t["FACTORS_ALL"]
        assert tuple(allf.shape) == (self.world, self.F)
        g = sum(float(allf[r, 0]) for r in range(self.world)) / self.world       # ascending rank order: bit-identical on every rank
        kl = sum(float(allf[r, -1]) for r in range(self.world)) / self.world
        self.t["AC_PARAMS"].sub_(0.1 * g)
        self.t["CV_PARAMS"].sub_(0.1 * g)
        self.applied.append((g, kl))
        self.step += 1

    def update_status(self):
        self.calls.append("status")

    def backward(self, which, mb):
        raise AssertionError("the gradient all-reduce path must not run when the factor exchange is available")

    apply = backward


def _factor_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("SDX_MULTI_RANK_GRAPH", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from seqdex_amd.a2c_agent import A2CAgent
    ag = A2CAgent.__new__(A2CAgent)
    ag.ppo = FakeFactorPPO(rank, world)
    ag.mini_epochs_num, ag.batch_size, ag.minibatch_size = 2, 12, 4
    ag.rank, ag.rank_size, ag.multi_gpu = rank, world, True
    ag._broadcast_parameters()
    ag._update_multi_gpu()
    out.put((rank, ag.ppo.t["AC_PARAMS"].tolist(), ag.ppo.applied, ag.ppo.calls, getattr(ag, "_mr_graph", None) is None))
    dist.destroy_process_group()


def test_factor_allgather_default_branch_world2():
    """the branch world size > 1 takes by DEFAULT with the shipped minibatch of 4 (a2c_agent.py::_update_multi_gpu): begin-of-epoch call,
    then per optimiser step backward_factors -> ONE all_gather_into_tensor -> apply_factors, eager (no graph capture at world size > 1
    unless SDX_MULTI_RANK_GRAPH=1), update_status at the end; both ranks end with identical parameters"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_factor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, a0, ap0, calls0, nograph0), (r1, a1, ap1, calls1, nograph1) = res
    assert a0 == a1 and ap0 == ap1                       # replicas identical
    assert nograph0 and nograph1                         # eager launches at world size 2
    assert calls0 == [-1] + [0] * 6 + ["status"]         # 2 mini-epochs x 3 minibatches; the minibatch index lives in the device cursor
    for k, (g, kl) in enumerate(ap0):
        assert abs(g - 1.5 * (k + 1)) < 1e-6 and abs(kl - 0.015) < 1e-7
    assert abs(a0[0] - (0.0 - 0.1 * 1.5 * sum(range(1, 7)))) < 1e-5
