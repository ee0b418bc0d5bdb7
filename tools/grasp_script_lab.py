"""how often does the scripted stand-in grasp (seqdex_amd/scripts/evaluation.py::scripted_grasp_controller) carry the target brick to the
insertion side?  N envs, one episode + reset, harvest gate's T-value opened.  python tools/grasp_script_lab.py [N]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import main_rlgames, scripted_grasp_controller  # noqa: E402
from tools.grasp_long_run import open_gate_tvalue  # noqa: E402,F401

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
trace = []
groups = []


def ctrl(task, step):
    if step in (1, 30, 52, 70, 76, 100, 126, 149):
        s = task.sim
        seg = torch.as_tensor([s.scene.seg_index(i) for i in range(n)], device=task.device)
        b = s.ROOT.view(n, 142, 13)[torch.arange(n, device=task.device), seg, 0:3]
        hb = s.RB[:, s.scene.hand_base_body, 0:3]
        trace.append((step, [round(float(x), 3) for x in b.mean(0)], [round(float(x), 3) for x in hb.mean(0)], round(float(b[:, 2].max()), 3),
                      int((b[:, 2] > 0.8).sum()), int((b[:, 1] < 0).sum()), round(float(s.FINGER_DIST.mean()), 3)))
    if step in (44, 70):
        s = task.sim
        seg = torch.as_tensor([s.scene.seg_index(i) for i in range(n)], device=task.device)
        b = s.ROOT.view(n, 142, 13)[torch.arange(n, device=task.device), seg, 0:3]
        hb = s.RB[:, s.scene.hand_base_body, 0:3]
        fd = s.FINGER_DIST
        for g in range(8):
            m = torch.arange(n, device=task.device) % 8 == g
            groups.append((step, g, [round(float(x), 3) for x in b[m].mean(0)], [round(float(x), 3) for x in (hb[m] - b[m]).mean(0)], round(float(fd[m].mean()), 3), round(float(fd[m].min()), 3)))
    return scripted_grasp_controller(task, step)


task, st = main_rlgames("BlockAssemblyGraspSim", n, tvalue_state=open_gate_tvalue(), controller=ctrl, steps=160, task_kwargs={"piles_per_type": 16})
st["harvested_per_type"] = task.sim.HARVEST_COUNT.cpu().tolist()
st["trace(step, mean brick pos, mean hand base, max brick z, bricks above 0.8, bricks at y<0, mean finger dist)"] = trace
print(json.dumps(st))
for g in groups:
    print(g)
task.sim.close()
