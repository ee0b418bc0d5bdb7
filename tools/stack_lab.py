"""the stacking cases of tests/test_physics_oracle.py through k_physics for solver variants: resting penetration per interface, creep,
whether the off-axis stacks stand.   python tools/stack_lab.py steps variant...   (variant = warm_start[:jacobi_relax[:key=value,...]])"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from seqdex_amd.scene import load_scene  # noqa: E402
from seqdex_amd.sim import SdxSim  # noqa: E402
from test_physics_oracle import STACKS_WARM, stacked_pair_state  # noqa: E402  (the oracle itself is not used here)

scene = load_scene()
steps = int(sys.argv[1])
for var in sys.argv[2:]:
    parts = var.split(":")
    over = {"warm_start": float(parts[0])}
    if len(parts) > 1 and parts[1]:
        over["jacobi_relax"] = float(parts[1])
    if len(parts) > 2:
        for kv in parts[2].split(","):
            k_, v_ = kv.split("=")
            over[k_] = float(v_) if k_ != "solver_iters" else int(v_)
    n = len(STACKS_WARM)
    roots, rests = [], []
    for (ia, ib, yaw, dx, dy) in STACKS_WARM:
        root, dof, tg, za, zb = stacked_pair_state(scene, ia, ib, yaw, dx, dy)
        roots.append(root[0]); rests.append((za, zb))
    root = np.stack(roots).astype(np.float32)
    dof = np.repeat(dof, n, 0); tg = np.repeat(tg, n, 0)
    s = SdxSim(n, **over)
    s.ROOT.copy_(torch.as_tensor(root.reshape(-1, 13)).cuda()); s.DOF.copy_(torch.as_tensor(dof.reshape(-1, 2)).cuda())
    s.TARGETS.copy_(torch.as_tensor(tg).cuda())
    for _ in range(steps):
        s.simulate()
    torch.cuda.synchronize()
    r = s.ROOT.cpu().numpy().reshape(n, 142, 13)
    res = []
    for e, (ia, ib, yaw, dx, dy) in enumerate(STACKS_WARM):
        za, zb = rests[e]
        sa = za - r[e, 9 + ia, 2]; sb = zb - r[e, 9 + ib, 2] - sa
        drift = float(np.hypot(r[e, 9 + ib, 0] - 0.25 - dx, r[e, 9 + ib, 1] - 0.19 - dy))
        res.append("%.2f/%.2f d%.1f" % (sa * 1e3, sb * 1e3, drift * 1e3))
    print(json.dumps({"variant": var, "steps": steps, "sink_a/sink_b mm, drift mm": res}))
    s.close()
