#!/usr/bin/env python3
"""Generate tests/golden/P1_settled_state.npz with the CPU oracle: a few settled brick piles + robot states
used as the starting point of the GPU-vs-oracle physics parity tests (test data only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import physics_oracle as po  # noqa: E402
from seqdex_amd.scene import load_scene  # noqa: E402

sc = load_scene()
d = sc.to_desc()
N = 8
rng = np.random.default_rng(7)
root = np.zeros((N, 142, 13), np.float32)
root[:, :, 6] = 1
raw = sc.raw
for i, fs in enumerate(raw["free_spawn"]):
    root[:, 9 + i, 0:3] = fs["pos"]
    root[:, 9 + i, 3:7] = fs["quat"]
    root[:, 9 + i, 0:2] += rng.uniform(-0.01, 0.01, (N, 2))
for i, fb in enumerate(raw["fixed_bricks"]):
    root[:, 9 + 72 + i, 0:3] = fb["pos"]
lo, hi = sc.lower, sc.upper
pose = np.concatenate([np.array(sc.arm_prepare_pose, np.float32),
                       0.5 * (np.array(sc.finger_reset_unscaled, np.float32) + 1) * (hi[7:] - lo[7:]) + lo[7:]])
dof = np.zeros((N, 23, 2), np.float32)
dof[:, :, 0] = pose
tg = np.tile(pose, (N, 1)).astype(np.float32)
for step in range(150):
    po.simulate(d, root, dof, tg)
# now drive the hand down into the pile (numeric DLS IK on the oracle's Jacobian) so that robot/brick contacts exist
qik = np.tile(pose, (N, 1)).astype(np.float32)
goal = np.array([0.25, 0.17, 0.905], np.float32) + rng.uniform(-0.03, 0.03, (N, 3)).astype(np.float32)
for it in range(60):
    dd = np.zeros((N, 23, 2), np.float32)
    dd[:, :, 0] = qik
    rbk, jk = po.kinematics(d, dd)
    err = np.concatenate([goal - rbk[:, 7, 0:3], np.zeros((N, 3), np.float32)], axis=1)
    for e in range(N):
        J = jk[e].astype(np.float64)
        qik[e, :7] += (J.T @ np.linalg.solve(J @ J.T + 0.01 * np.eye(6), 0.5 * err[e])).astype(np.float32)
    qik[:, :7] = np.clip(qik[:, :7], lo[:7], hi[:7])
tg2 = qik.copy()
tg2[:, 7:] = lo[7:] + (hi[7:] - lo[7:]) * rng.uniform(0.3, 0.9, (N, 16)).astype(np.float32)
for step in range(60):
    rb, contact, jac, nc = po.simulate(d, root, dof, tg2)
print("link7", rb[:, 7, :3])
print("contacts", nc, "arm contact |f|", np.linalg.norm(contact[:, 1:24], axis=-1).max(axis=1))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "P1_settled_state.npz")
np.savez_compressed(out, root=root, dof=dof, targets=tg2)
print("wrote", out, os.path.getsize(out))
