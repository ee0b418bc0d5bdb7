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


    # ---- the same two updates in ONE process: two SdxPPO handles at world size 2 loaded with what each rank's update started from, the
    # collective replaced by tensor copies (the emulation of tests/test_gpu_fullsize_properties.py) - the replicas of the two-process
    # run must agree with it (see the comment at the comparison for why not bit for bit).  Same forward / backward implementation as the workers.
    import torch
    from seqdex_amd.ppo import SdxPPO, make_config
    os.environ["SDXP_STEP_IMPL"] = "kernels"
    try:
        cfgd = {"config": {"minibatch_size": minibatch, "central_value_config": {"minibatch_size": minibatch}}}
        import yaml
        train = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/lego/ppo_continuous_grasp.yaml")))
        train["params"]["config"]["minibatch_size"] = minibatch
        train["params"]["config"]["central_value_config"]["minibatch_size"] = minibatch
        ranks = []
        for d in (a, b):
            ag = SdxPPO(n, config=make_config(n, train["params"], world_size=world), seed=22)
            for k in d.files:
                if k.startswith("pre_"):
                    t = ag.t[k[4:]]
                    t.copy_(torch.from_numpy(d[k]).view(t.shape).to(t.device))
            ranks.append(ag)
        torch.cuda.synchronize()
        A, B = ranks
        nmb = n * 8 // minibatch
        if minibatch <= 8:
            for ag in ranks:
                ag.backward_factors(-1)
            for _ in range(5):
                for mb in range(nmb):
                    A.backward_factors(mb); B.backward_factors(mb)
                    f = torch.stack([A.t["FACTORS"].view(-1), B.t["FACTORS"].view(-1)])
                    A.t["FACTORS_ALL"].view(2, -1).copy_(f); B.t["FACTORS_ALL"].view(2, -1).copy_(f)
                    A.apply_factors(); B.apply_factors()
        else:
            for ag in ranks:
                ag.backward(0, -1)
            for _ in range(5):
                for mb in range(nmb):
                    A.backward(0, mb); B.backward(0, mb)
                    sm = A.t["ALL_GRADS"] + B.t["ALL_GRADS"]
                    A.t["ALL_GRADS"].copy_(sm); B.t["ALL_GRADS"].copy_(sm)
                    for ag in ranks:
                        ag.apply(0, float("-inf")); ag.apply(1)
        torch.cuda.synchronize()
        # Bit equality is NOT attainable on this forward / backward implementation: the multi-kernel step (SDXP_STEP_IMPL=kernels, the only
        # one two processes can run beside each other on one GPU) sums its split-N data gradients and Gram partials with float atomics
        # (csrc/sdxp_kernels.hip: k_back, gram_sums), so each rank's FACTORS differ by an ulp from run to run; the replicas of one run still
        # agree bit for bit (asserted above) because both rebuild the gradient from the same gathered factors.  What is asserted is that the
        # emulation lands within the noise those atomics can make over the run (measured: 2e-6 after 160 steps), with its own replicas identical.
        for k, tk in (("ac", "AC_PARAMS"), ("cv", "CV_PARAMS")):
            np.testing.assert_array_equal(A.t[tk].cpu().numpy(), B.t[tk].cpu().numpy())
            np.testing.assert_allclose(A.t[tk].cpu().numpy().ravel(), a[k].ravel(), rtol=0, atol=2e-5, err_msg="one-process emulation vs two processes: " + tk)
            assert np.abs(A.t[tk].cpu().numpy().ravel() - a[k + "0"].ravel()).max() > 1e-4       # (the update moved the parameters by far more than that)
    finally:
        os.environ.pop("SDXP_STEP_IMPL", None)
        for ag in locals().get("ranks", []):
            ag.close()
