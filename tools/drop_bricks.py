"""Search-style reset at scale: 72 bricks per env dropped from the spawn lattice (+-2 cm noise) into the bin, N envs, `steps` simulate()
calls; counts the bricks that end outside the bin and the state of the settled piles, for the cold solver and for the optional warm
start.   python tools/drop_bricks.py [N] [steps] [warm_start[:jacobi_relax[:key=value,...]] ...]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from seqdex_amd.sim import SdxSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = sys.argv[3:] or ["0.0", "0.8"]
for var in variants:
    parts = var.split(":")
    beta = float(parts[0])
    over = {"warm_start": beta}
    if len(parts) > 1 and parts[1]:
        over["jacobi_relax"] = float(parts[1])
    if len(parts) > 2:
        for kv in parts[2].split(","):
            k_, v_ = kv.split("=")
            over[k_] = float(v_)
    s = SdxSim(n, **over)
    sc = s.scene
    g = torch.Generator().manual_seed(5)
    root = s.ROOT.view(n, 142, 13)
    root[:, 9:81, 0:2] += ((torch.rand(n, 72, 2, generator=g) * 2 - 1) * 0.02).to(root.device)
    lo, hi = sc.lower, sc.upper
    pose = np.concatenate([np.array(sc.arm_prepare_pose, np.float32),
                           0.5 * (np.array(sc.finger_reset_unscaled, np.float32) + 1) * (hi[7:] - lo[7:]) + lo[7:]])
    dof = torch.zeros(n, 23, 2)
    dof[:, :, 0] = torch.as_tensor(pose)
    s.DOF.copy_(dof.view(-1, 2).to(s.DOF.device))
    s.TARGETS.copy_(torch.as_tensor(np.tile(pose, (n, 1))).to(s.DOF.device))
    ke = {}
    for k in range(steps):
        s.simulate()
        if k + 1 in (20, 40, 80, steps):
            ke[k + 1] = float((s.ROOT.view(n, 142, 13)[:, 9:81, 7:10].norm(dim=-1) ** 2).mean())
    torch.cuda.synchronize()
    r = s.ROOT.view(n, 142, 13)[:, 9:81].cpu().numpy()
    out = (np.abs(r[:, :, 0] - 0.25) > 0.3) | (np.abs(r[:, :, 1] - 0.19) > 0.21) | (r[:, :, 2] < 0.55)
    inside = ~out
    print(json.dumps({"variant": var, "warm_start": beta, "n_envs": n, "steps": steps, "bricks": int(n * 72), "escaped": int(out.sum()),
                      "envs_with_escapes": int(out.any(1).sum()), "mean_v2_at": ke,
                      "settled_mean_speed": float(np.linalg.norm(r[:, :, 7:10], axis=-1)[inside].mean()),
                      "lowest_brick_origin_z": float(r[:, :, 2][inside].min()),
                      "contacts_mean": float(s.NCONTACTS.float().mean()), "contact_stats": s.CONTACT_STATS.cpu().tolist()}))
    s.close()
