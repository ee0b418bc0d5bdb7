"""Does the hand lift a brick?  N GraspSim envs under the scripted reach - descend - pinch controller (evaluation.py::scripted_grasp_controller;
after step 75 the task lifts the hand itself, GS:1596-1609), one episode; prints what the target brick, the fingertips and the fingertip
contact forces do over the episode and how many envs end with the brick lifted >= 5 cm and finger_dist < 0.5.
python tools/lift_diag.py [N] [--dump K]   (--dump: per-step trace of the first K envs of type group 0)"""
import json
import os
import sys

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import scripted_grasp_controller  # noqa: E402
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1024
dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else 0
variants = sys.argv[sys.argv.index("--variants") + 1].split(";") if "--variants" in sys.argv else [os.environ.get("SDX_SG_PARAMS", "")]
cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
cfg["env"]["numEnvs"] = n


def run(variant, dump):
    os.environ["SDX_SG_PARAMS"] = variant
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=16)
    s = task.sim
    dev = task.device
    seg = torch.as_tensor([s.scene.seg_index(i) for i in range(n)], device=dev)
    ar = torch.arange(n, device=dev)
    tips = list(s.scene.fingertip_bodies)
    rows = []
    task.step(torch.zeros(n, 23, device=dev))          # the reset step
    z0 = s.INIT_POS[:, 2].clone()
    lifted_max = torch.zeros(n, device=dev)
    held_max = torch.zeros(n, device=dev)               # largest lift of the target brick WHILE finger_dist < 0.5, first episode only
    first = torch.ones(n, dtype=torch.bool, device=dev)
    for step in range(1, 150):
        a = scripted_grasp_controller(task, step)
        task.step(a)
        root = s.ROOT.view(n, 142, 13)
        b = root[ar, seg]
        first &= s.PROGRESS > 1                          # an env that was reset is out of its first episode
        dz = b[:, 2] - z0
        fd = s.FINGER_DIST
        lifted_max = torch.where(first, torch.maximum(lifted_max, dz), lifted_max)
        held_max = torch.where(first & (fd < 0.5), torch.maximum(held_max, dz), held_max)
        if step in (20, 40, 58, 66, 72, 76, 80, 90, 100, 110, 125, 140, 148):
            tipp = s.RB[:, tips, 0:3]
            dtip = (tipp - b[:, None, 0:3]).norm(dim=-1)
            cf = s.CONTACT.view(n, 165, 3)[:, tips].norm(dim=-1)
            hb = s.RB[:, s.scene.hand_base_body, 0:3]
            rows.append({"step": step, "in_first_episode": int(first.sum()), "dz_mean_mm": float(dz[first].mean() * 1e3) if first.any() else None,
                         "lifted_5cm_now": int(((dz > 0.05) & first).sum()), "held_5cm_now": int(((dz > 0.05) & (fd < 0.5) & first).sum()),
                         "finger_dist_mean": float(fd.mean()), "fd_lt_0.5": int((fd < 0.5).sum()),
                         "tip_dist_mean_mm(ff,mf,rf,th)": [round(float(x) * 1e3, 1) for x in dtip.mean(0)],
                         "tip_force_mean_N": [round(float(x), 2) for x in cf.mean(0)],
                         "hand_z_mean": float(hb[:, 2].mean()), "ncontacts_mean": float(s.NCONTACTS.float().mean())})
        if dump and step % 2 == 0:
            for e in range(0, 8 * dump, 8):
                print("trace", step, e, [round(float(x), 4) for x in b[e, 0:3]], "dz %.4f" % float(dz[e]), "fd %.3f" % float(s.FINGER_DIST[e]),
                      "tipF", [round(float(x), 2) for x in s.CONTACT.view(n, 165, 3)[e, tips].norm(dim=-1)], "q", [round(float(x), 3) for x in s.DOF.view(n, 23, 2)[e, 7:, 0]])
    torch.cuda.synchronize()
    ok = held_max > 0.05
    out = {"variant(closure,steps,rise)": variant, "n": n, "held_5cm_frac": float(ok.float().mean()), "held_5cm": int(ok.sum()), "held_2cm": int((held_max > 0.02).sum()),
           "lifted_5cm_any_grip": int((lifted_max > 0.05).sum()), "survived_first_episode_to_the_end": int(first.sum()),
           "contact_stats": s.CONTACT_STATS.cpu().tolist(), "per_type_held_5cm": [int(ok[ar % 8 == g].sum()) for g in range(8)]}
    s.close()
    return out, rows


for i, v in enumerate(variants):
    out, rows = run(v, dump if i == 0 else 0)
    print(json.dumps(out))
    if i == 0:
        for r in rows:
            print(json.dumps(r))
