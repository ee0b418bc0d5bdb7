#!/usr/bin/env python3
"""Several BlockAssemblyGraspSim training runs in one process, one JSON summary: seeds of the schedule that
learns, and the minibatch / learning-rate sweep around the shipped schedule that does not.
usage: python tools/train_sweep.py --runs "mb:lr:seed:epochs[,...]" [--envs 1024] [--every 100] --out file.json
  mb = minibatch_size, lr = "adaptive" (the shipped rule, kl_threshold 0.02, from the shipped 3e-4), "adaptive@<lr0>" (the rule from another
  initial rate) or a constant learning rate, seed = task + agent seed.
Per run: the logged curve (epoch, game reward, harvested grasp states, epoch-mean KL, learning rate), final / maximum game reward, the
learning-rate trajectory summarised (share of epochs that ended with a lower / higher / unchanged rate than they began with, share of epochs
at the 1e-6 floor / the 1e-2 ceiling), wall time."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.a2c_agent import A2CAgent  # noqa: E402
from seqdex_amd.config import TASK_CFG, TRAIN_CFG  # noqa: E402
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402
from tools.grasp_long_run import open_gate_tvalue  # noqa: E402


def one_run(n, mb, lr, seed, epochs, every, max_s):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyGraspSim"])))
    pc = tr["params"]["config"]
    pc["minibatch_size"] = mb
    pc["central_value_config"]["minibatch_size"] = mb
    if lr.startswith("adaptive@"):          # the shipped adaptive rule from another initial learning rate
        pc["learning_rate"] = float(lr.split("@")[1])
    elif lr != "adaptive":
        pc["lr_schedule"] = "constant"
        pc["learning_rate"] = float(lr)
    torch.manual_seed(seed)
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, piles_per_type=16)
    task.sim.set_tvalue_weights(open_gate_tvalue())
    env = RLgamesVecTaskPython(task, "cuda:0")
    pc.update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=seed)
    agent = A2CAgent("sweep", tr["params"])
    t0 = time.time()
    curve, lrs, kls = [], [], []
    rew_max = -1e9
    ep = -1
    for ep in range(epochs):
        r = agent.train_epoch()
        lrs.append(float(r[9])); kls.append(float(r[8][0]))
        if (ep + 1) % every == 0 or ep == 0 or ep == epochs - 1:
            gr = float(agent.game_rewards.get_mean()[0])
            rew_max = max(rew_max, gr)
            curve.append({"epoch": ep + 1, "game_reward": round(gr, 2), "harvested": int(task.sim.HARVEST_COUNT.sum()), "kl": round(kls[-1], 5), "lr": lrs[-1]})
            if time.time() - t0 > max_s:
                break
    torch.cuda.synchronize()
    lr_a = np.asarray(lrs)
    d = np.diff(np.concatenate([[lr_a[0]], lr_a]))
    out = {"minibatch": mb, "lr": lr, "seed": seed, "epochs_run": ep + 1, "update_impl": agent.ppo.update_impl(), "wall_s": round(time.time() - t0, 1),
           "final_game_reward": round(float(agent.game_rewards.get_mean()[0]), 2), "max_logged_game_reward": round(rew_max, 2),
           "harvested_per_type": task.sim.HARVEST_COUNT.cpu().numpy().tolist(),
           "lr_epochs_down_up_same": [round(float((d < 0).mean()), 3), round(float((d > 0).mean()), 3), round(float((d == 0).mean()), 3)],
           "lr_share_at_floor_1e-6": round(float((lr_a <= 1.0001e-6).mean()), 3), "lr_share_at_ceiling_1e-2": round(float((lr_a >= 0.9999e-2).mean()), 3),
           "lr_median": float(np.median(lr_a)), "kl_epoch_mean_quantiles_10_50_90": [round(float(np.quantile(kls, q)), 5) for q in (0.1, 0.5, 0.9)],
           "contact_stats": task.sim.CONTACT_STATS.cpu().numpy().tolist(), "curve": curve}
    agent.ppo.close() if hasattr(agent.ppo, "close") else None
    task.sim.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", required=True)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--max-seconds", type=float, default=1e9, help="per run")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    res = []
    for spec in a.runs.split(","):
        mb, lr, seed, epochs = spec.split(":")
        r = one_run(a.envs, int(mb), lr, int(seed), int(epochs), a.every, a.max_seconds)
        res.append(r)
        print(json.dumps({k: v for k, v in r.items() if k != "curve"}), flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump({"envs": a.envs, "runs": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
