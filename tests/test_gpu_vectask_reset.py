"""RLgamesVecTaskPython.reset() on the GPU against the reference's reset semantics (tasks/hand_base/vec_task_rlgames.py:176-192, fixture
F9 of oracle/gen_golden.py): reset() does NOT reset the envs - it takes ONE task step with the noise action 0.01 (1 - 2 U), U drawn from
torch's global generator on the rl device, and returns the clamped observation dictionary."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grasp(n, seed):
    from seqdex_amd.config import TASK_CFG
    from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, piles_per_type=2)
    return task, RLgamesVecTaskPython(task, "cuda:0")


def test_reset_is_one_noise_step(golden_dir):
    n = 64
    ta, ea = _grasp(n, 5)
    seen = []
    inner = ta.step
    ta.step = lambda a: (seen.append(a.clone()), inner(a))[1]
    p0 = ta.progress_buf.clone()
    torch.manual_seed(9)
    od = ea.reset()
    torch.cuda.synchronize()
    # one step, with exactly the action the reference's expression draws from the same generator state (VR:180)
    torch.manual_seed(9)
    expect = 0.01 * (1 - 2 * torch.rand([n, 23], dtype=torch.float32, device="cuda:0"))
    assert len(seen) == 1 and seen[0].dtype == torch.float32 and tuple(seen[0].shape) == (n, 23)
    assert torch.equal(seen[0], expect)
    g = np.load(os.path.join(golden_dir, "F9_vectask.npz"))
    assert float(seen[0].abs().max()) <= 0.01 and float(g["reset_actions_absmax"]) <= 0.01 + 1e-7
    # the envs took one control step: a fresh task starts with reset_buf = 1 (BT:60), so that step first resets every env (progress 0,
    # GS:1523) and then counts it (GS:1092) - reset() itself never touches reset_buf (VR:176-192)
    assert int(p0.abs().max()) == 0 and bool((ta.progress_buf == 1).all())
    # the dictionary holds the clamped buffers (VR:186-190), as views of the simulator's tensors
    assert sorted(od.keys()) == ["obs", "states"]
    assert tuple(od["obs"].shape) == (n, 396) and tuple(od["states"].shape) == (n, 564)
    assert torch.equal(od["obs"], torch.clamp(ta.sim.OBS, -5.0, 5.0)) and torch.equal(od["states"], torch.clamp(ta.sim.STATES, -5.0, 5.0))
    obs_a, st_a = od["obs"].clone(), od["states"].clone()
    root_a, dof_a = ta.sim.ROOT.clone(), ta.sim.DOF.clone()
    ta.sim.close()
    # the same instance driven by step(a) with that action lands in the same state bit for bit: reset() adds nothing else
    tb, eb = _grasp(n, 5)
    od_b, _, _, _ = eb.step(expect)
    torch.cuda.synchronize()
    assert torch.equal(od_b["obs"], obs_a) and torch.equal(od_b["states"], st_a)
    assert torch.equal(tb.sim.ROOT, root_a) and torch.equal(tb.sim.DOF, dof_a)
    tb.sim.close()
