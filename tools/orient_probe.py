"""Exploration for the chain on learned policies: stage 0 (insert policy + transition value), statistics of that transition value over random
orientations, then BlockAssemblyOrient trained with minibatches of 2048 under gate `gate`: piles harvested per brick-type group.
usage: python tools/orient_probe.py N insert_epochs orient_epochs gate"""
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
from seqdex_amd.config import TASK_CFG, TRAIN_CFG, set_seed  # noqa: E402
from seqdex_amd.scripts.evaluation import prepare_tvalue_and_insert_policy  # noqa: E402
from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient  # noqa: E402
from seqdex_amd.tvalue_trainer import LAYERS  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402

n, ie, oe, gate = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
tv, _, ist = prepare_tvalue_and_insert_policy(n, ie, seed=22)
print("stage 0:", json.dumps(ist), flush=True)
# the fitted MLP on 200 000 random orientations (torch on the CPU: exploration only)
w = torch.from_numpy(np.asarray(tv, np.float32))
off, params = 0, []
for _, out, inn in LAYERS:
    W = w[off:off + out * inn].view(out, inn); off += out * inn
    b = w[off:off + out]; off += out
    params.append((W, b))
g = torch.Generator().manual_seed(0)
q = torch.randn(200000, 4, generator=g); q = q / q.norm(dim=1, keepdim=True)
x = q
for i, (W, b) in enumerate(params):
    x = x @ W.t() + b
    if i < len(params) - 1:
        x = torch.relu(x)
t = torch.sigmoid(x)[:, 1]
print("T over random orientations: max %.4f  >0.5 %.4f  >0.8 %.4f  >0.9 %.5f  >0.99 %.5f" % (float(t.max()), float((t > 0.5).float().mean()), float((t > 0.8).float().mean()),
                                                                                           float((t > 0.9).float().mean()), float((t > 0.99).float().mean())), flush=True)
set_seed(22)
cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyOrient"])))
cfg["env"]["numEnvs"] = n
tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyOrient"])))
tr["params"]["config"]["minibatch_size"] = 2048
tr["params"]["config"]["central_value_config"]["minibatch_size"] = 2048
task = BlockAssemblyOrient(cfg, device_type="cuda", device_id=0, headless=True, seed=22, tvalue_gate=gate, piles_per_type=64)
task.sim.set_tvalue_weights(tv)
env = RLgamesVecTaskPython(task, "cuda:0")
tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
agent = A2CAgent("run", tr["params"])
t0 = time.time()
for ep in range(oe):
    agent.train_epoch()
    if (ep + 1) % 100 == 0 or ep == 0:
        torch.cuda.synchronize()
        print("epoch %4d  game reward %8.3f  len %5.1f  piles harvested/type %s  T-value outcomes %s  tvalue mean %.4f  %.0f env-steps/s"
              % (ep + 1, agent.game_rewards.get_mean()[0], agent.game_lengths.get_mean()[0], task.sim.PILE_HARVEST_COUNT.cpu().tolist(), task.sim.TV_COUNT.cpu().tolist(),
                 float(task.sim.TVALUE.mean()), n * 8 * (ep + 1) / (time.time() - t0)), flush=True)
