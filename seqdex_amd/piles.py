"""Synthetic saved pile states [8, K, 132, 13]: the reference reads them from a pickle produced by
its Search->Orient stages (GS:412-413, 1507-1513); that artefact is not shipped, so the engine settles its own:
72 free bricks dropped from the spawn lattice (GS:737-742) with per-pile jitter, robot parked in the prepare pose,
`steps` simulate() calls on the GPU."""
import numpy as np
import torch

from .sim import SdxSim


def generate_piles(per_type=8, steps=150, device="cuda:0", seed=22):
    n = 8 * per_type
    sim = SdxSim(n, device=device, seed=seed)
    try:
        sc = sim.scene
        g = torch.Generator().manual_seed(seed)
        root = sim.ROOT.view(n, 142, 13)
        jit = (torch.rand(n, 72, 2, generator=g) * 2 - 1) * 0.012
        root[:, 9:81, 0:2] += jit.to(root.device)
        yaw = (torch.rand(n, 72, generator=g) * 2 - 1) * 3.14159
        root[:, 9:81, 3] = 0.0
        root[:, 9:81, 4] = 0.0
        root[:, 9:81, 5] = torch.sin(yaw / 2).to(root.device)
        root[:, 9:81, 6] = torch.cos(yaw / 2).to(root.device)
        lo, hi = sc.lower, sc.upper
        pose = np.concatenate([np.array(sc.arm_prepare_pose, np.float32),
                               0.5 * (np.array(sc.finger_reset_unscaled, np.float32) + 1) * (hi[7:] - lo[7:]) + lo[7:]])
        dof = torch.zeros(n, 23, 2)
        dof[:, :, 0] = torch.as_tensor(pose)
        sim.DOF.copy_(dof.view(-1, 2).to(sim.DOF.device))
        sim.TARGETS.copy_(torch.as_tensor(np.tile(pose, (n, 1))).to(sim.DOF.device))
        for _ in range(steps):
            sim.simulate()
        torch.cuda.synchronize()
        piles = sim.ROOT.view(n, 142, 13)[:, 9:141].clone()
        piles[:, :, 7:13] = 0.0
        # env i of the generator uses type group i % 8 only as a label: piles are exchangeable across groups
        out = piles.view(per_type, 8, 132, 13).permute(1, 0, 2, 3).contiguous().cpu().numpy()
        assert np.isfinite(out).all()
        return out
    finally:
        sim.close()
