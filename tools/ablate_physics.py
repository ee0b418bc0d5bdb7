#!/usr/bin/env python3
"""Where does k_physics' LAUNCH time go?  Timing ablations on an identical state (profiling build only: make -C seqdex_amd/csrc prof,
SDX_LIB_PATH=.../libseqdex_prof.so).  The bench workload is stepped to a contact-rich state, every piece of state the kernel reads is
snapshotted (root, dof, targets, warm-start cache), and each configuration is timed on launches that start from that snapshot with a set
of ablation bits in SDX_T_DEBUG[63] (csrc/sdx_physics.hip: ABL).  launch(all) - launch(without X) = what X costs with both
workgroups of a CU competing - which the single-env phase clock cannot say.
usage: python tools/ablate_physics.py [N] [warm-steps]"""
import json
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
cfg["env"]["numEnvs"] = n
task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=8)
s = task.sim
g = torch.Generator().manual_seed(0)
for _ in range(warm):
    task.step((torch.rand(n, 23, generator=g) * 2 - 1).cuda())
s.DEBUG[63] = 0
torch.cuda.synchronize()
names = ["ROOT", "DOF", "TARGETS", "WARM_COUNT", "WARM_KEYS", "WARM_LAMBDA"]
snap = {k: s.tensor(k).clone() for k in names}


def timed(bits, reps=6):
    ts = []
    for _ in range(reps):
        for k in names:
            s.tensor(k).copy_(snap[k])
        s.DEBUG[63] = bits
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.simulate()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


configs = [("all", 0), ("no FK + drive in substep 1", 256), ("no mass matrix", 512), ("no gather loop [D]", 1), ("no [AC] body", 2), ("no [AC], no gather", 3), ("one solver iteration", 4), ("no robot section", 8),
           ("no classification (no contacts)", 16), ("no broadphase (no pairs)", 32), ("no rank pass", 64), ("no row weights", 128),
           ("one iteration, no classification", 4 | 16)]
base = timed(0)
out = {"n_envs": n, "note": "median of 6 single launches (k_physics + k_order) from one snapshot; ms", "all": base, "without": {}}
for name, bits in configs[1:]:
    t = timed(bits)
    out["without"][name] = {"ms": round(t, 4), "saves_ms": round(base - t, 4), "share": round((base - t) / base, 3)}
out["all_again"] = timed(0)
print(json.dumps(out, indent=1))
