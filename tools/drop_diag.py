"""diagnostic for the drop scenario of tools/drop_bricks.py: per step, how many bricks move upwards faster than 1 / 2 m/s or faster than free fall allows
(launch events = energy the solver added), and the deepest brick/brick centre overlap.  python tools/drop_diag.py N steps variant..."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from seqdex_amd.sim import SdxSim  # noqa: E402

n = int(sys.argv[1]); steps = int(sys.argv[2])
for var in sys.argv[3:]:
    parts = var.split(":")
    over = {"warm_start": float(parts[0])}
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
    dof = torch.zeros(n, 23, 2); dof[:, :, 0] = torch.as_tensor(pose)
    s.DOF.copy_(dof.view(-1, 2).to(s.DOF.device))
    s.TARGETS.copy_(torch.as_tensor(np.tile(pose, (n, 1))).to(s.DOF.device))
    up1 = up2 = fast = 0
    exited = torch.zeros(n, 72, dtype=torch.bool, device=s.ROOT.device)
    exit_z = []
    hist = []
    for k in range(steps):
        s.simulate()
        v = s.ROOT.view(n, 142, 13)[:, 9:81, 7:10]
        a, b, c = int((v[..., 2] > 1.0).sum()), int((v[..., 2] > 2.0).sum()), int((v.norm(dim=-1) > 4.0).sum())
        up1 += a; up2 += b; fast += c
        p = s.ROOT.view(n, 142, 13)[:, 9:81, 0:3]
        # centre beyond the OUTER face of a wall (walls are 1 cm thick: inner faces x = -0.04 / 0.54, y = -0.008 / 0.388)
        outside = ((p[..., 0] < -0.05) | (p[..., 0] > 0.55) | (p[..., 1] < -0.018) | (p[..., 1] > 0.398)) & ~exited
        if bool(outside.any()):
            exit_z += [(k + 1, round(float(z), 3), round(float(vz), 2)) for z, vz in zip(p[..., 2][outside].tolist(), v[..., 2][outside].tolist())]
            exited |= outside
        if k % 10 == 9:
            hist.append((k + 1, a, b, c, float((v.norm(dim=-1) ** 2).mean())))
    r = s.ROOT.view(n, 142, 13)[:, 9:81].cpu().numpy()
    out = (np.abs(r[:, :, 0] - 0.25) > 0.3) | (np.abs(r[:, :, 1] - 0.19) > 0.21) | (r[:, :, 2] < 0.55)
    print(json.dumps({"variant": var, "escaped": int(out.sum()), "brick_steps_vz>1": up1, "brick_steps_vz>2": up2, "brick_steps_speed>4": fast,
                      "exits_over_wall_top(z>0.765)": sum(1 for e in exit_z if e[1] > 0.765), "exits_below_wall_top": sum(1 for e in exit_z if e[1] <= 0.765), "exit_events(step,z,vz)": exit_z[:40], "every10": hist[:4]}))
    s.close()
