#!/usr/bin/env python3
"""BASELINE.json configs[2]: BlockAssemblyOrient + BlockAssemblyInsertSim chained at num_envs = 1024 on one MI355X.  Each task trains
with its own shipped PPO schedule for `epochs` epochs (one epoch = 8 env steps x N + the update); Orient's harvested pile states
are what the next stage would load (printed K); InsertSim starts from its synthetic grasp states (no GraspSim stage in this config).
Prints one JSON line.  usage: python tools/bench_config3.py [N] [epochs]"""
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.a2c_agent import A2CAgent  # noqa: E402
from seqdex_amd.config import TASK_CFG, TRAIN_CFG  # noqa: E402
from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim  # noqa: E402
from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out = {"config": "BASELINE.json configs[2]: Orient -> InsertSim chained, num_envs=%d, 1 GPU" % n, "epochs_per_task": epochs}
for name, cls in (("BlockAssemblyOrient", BlockAssemblyOrient), ("BlockAssemblyInsertSim", BlockAssemblyInsertSim)):
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG[name])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG[name])))
    task = cls(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=8)
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
    agent = A2CAgent("run", tr["params"])
    agent.train_epoch()                                   # warm-up (first step = reset of every env)
    torch.cuda.synchronize()
    t0 = time.time()
    step_t = play_t = upd_t = 0.0
    for _ in range(epochs):
        r = agent.train_epoch()
        step_t += r[0]; play_t += r[1]; upd_t += r[2]
    torch.cuda.synchronize()
    dt = time.time() - t0
    e = {"env_steps_per_s": n * 8 * epochs / dt, "fps_step": n * 8 * epochs / step_t, "rollout_ms_per_epoch": play_t / epochs * 1e3,
         "update_ms_per_epoch": upd_t / epochs * 1e3, "minibatch_size": agent.minibatch_size, "update_impl": agent.ppo.update_impl(),
         "episode_length": int(task.max_episode_length), "mean_reward": float(task.rew_buf.mean().item())}
    if name == "BlockAssemblyOrient":
        piles = task.pile_terminal_states()
        e["harvested_piles_per_type"] = 0 if piles is None else int(piles.shape[1])
        e["note"] = "an episode is 75 steps; its reset event costs 103 extra simulator steps of all envs (two scripted 50-step phases, OR:1427-1461,1655-1695)"
    out[name] = e
    del agent, env, task
    torch.cuda.empty_cache()
out["chain_env_steps_per_s"] = 2.0 / (1.0 / out["BlockAssemblyOrient"]["env_steps_per_s"] + 1.0 / out["BlockAssemblyInsertSim"]["env_steps_per_s"])
print(json.dumps(out))
