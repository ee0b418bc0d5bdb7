"""resting penetration of settled piles at scale: N envs dropped as in tools/drop_bricks.py, then the contact list of sampled envs (the
oracle's collide() on the device state: TEST/DIAGNOSTIC use of the oracle) -> distribution of the separations.
python tests/helpers/pile_penetration.py N steps variant..."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import physics_oracle as po  # noqa: E402
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
    for k in range(steps):
        s.simulate()
    torch.cuda.synchronize()
    r = s.ROOT.cpu().numpy().reshape(n, 142, 13)
    d = s.DOF.cpu().numpy().reshape(n, 23, 2)
    seps = []
    for e in range(0, n, max(1, n // 32)):
        c, total = po.contacts(s._desc, r[e], d[e])
        seps.append(c[:, 8])
    seps = np.concatenate(seps)
    off = float(s._desc.contact_offset)
    print(json.dumps({"variant": var, "contacts_sampled": int(seps.size), "deepest_mm": float(-seps.min() * 1e3),
                      "p99.9_depth_mm": float(-np.quantile(seps, 0.001) * 1e3), "p99_depth_mm": float(-np.quantile(seps, 0.01) * 1e3),
                      "median_depth_mm": float(-np.median(seps) * 1e3), "frac_deeper_than_offset": float((seps < -off).mean())}))
    s.close()
