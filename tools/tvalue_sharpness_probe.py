"""How confident does the chain's transition value get with the reference's number of fit iterations (bi_optimization.py:120: rollout=10000;
the chain's stage 0 stops after 3 000)?  Stage 0's insert outcomes are fitted for 3 000 / 10 000 / 30 000 iterations (the same trainer, going
on); for every fit: the value over random orientations, the piles BlockAssemblyOrient harvests in 640 steps per env under the gates 0.99 / 0.9 /
0.8, and the grasp states a learned grasp policy harvests in `grasp_steps` steps per env under gate 0.8.
usage: python tools/tvalue_sharpness_probe.py N [grasp_steps]"""
import json
import os
import sys
import tempfile

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.a2c_agent import A2CAgent  # noqa: E402
from seqdex_amd.config import TASK_CFG, TRAIN_CFG, set_seed  # noqa: E402
from seqdex_amd.scripts.evaluation import main_rlgames, train_grasp_policy  # noqa: E402
from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim  # noqa: E402
from seqdex_amd.tvalue_trainer import TValue_Trainer, flat_from_state_dict  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402

n = int(sys.argv[1])
grasp_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
set_seed(22)
cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyInsertSim"])))
cfg["env"]["numEnvs"] = n
tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyInsertSim"])))
task = BlockAssemblyInsertSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22)
env = RLgamesVecTaskPython(task, "cuda:0")
tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
agent = A2CAgent("run", tr["params"])
for _ in range(1500):
    agent.train_epoch()
torch.cuda.synchronize()
print("stage 0 outcomes (success, failure):", task.sim.TV_COUNT.cpu().tolist(), flush=True)
trn = TValue_Trainer.from_task(task, seed=22)
trn.init_TValue_function("BlockAssemblyInsertSim", 3000)
g = torch.Generator().manual_seed(0)
q = torch.randn(20000, 4, generator=g)
q = (q / q.norm(dim=1, keepdim=True)).to(task.sim.device)
fits = {}
done = 0
for total in (3000, 10000, 30000):
    trn.train_rollout(total - done)
    done = total
    out = torch.cat([trn.predict(q[i:i + 1024]) for i in range(0, q.shape[0], 1024)])
    t = torch.sigmoid(out)[:, 1]
    fits[total] = flat_from_state_dict(trn.state_dict()).numpy()
    print("fit %6d iterations: loss %.4f  held-out success rate %.3f  T over random orientations: max %.4f  >0.5 %.4f  >0.8 %.4f  >0.9 %.5f  >0.99 %.5f"
          % (total, trn.losses[-1], trn.valid_t_value_success_rate, float(t.max()), float((t > 0.5).float().mean()), float((t > 0.8).float().mean()),
             float((t > 0.9).float().mean()), float((t > 0.99).float().mean())), flush=True)
trn.close()
agent.ppo.close()
task.sim.close()
work = tempfile.mkdtemp(prefix="sdx_tvsharp_")
gpath, gtask, gst = train_grasp_policy(n, 1500, save_to=os.path.join(work, "grasp"), tvalue_state=fits[3000])
gtask.sim.close()
print("grasp policy:", json.dumps(gst), flush=True)
for total, tv in fits.items():
    for gate in (0.99, 0.9, 0.8):
        orient, st = main_rlgames("BlockAssemblyOrient", n, tvalue_state=tv, steps=640, task_kwargs={"tvalue_gate": gate, "piles_per_type": 64})
        print("fit %6d  Orient gate %.2f: piles harvested per type %s" % (total, gate, orient.sim.PILE_HARVEST_COUNT.cpu().tolist()), flush=True)
        orient.sim.close()
    grasp, st = main_rlgames("BlockAssemblyGraspSim", n, policy_path=gpath, tvalue_state=tv, steps=grasp_steps, task_kwargs={"harvest_tvalue_gate": 0.8})
    print("fit %6d  GraspSim gate 0.80, %d steps per env: grasp states harvested per type %s" % (total, grasp_steps, grasp.sim.HARVEST_COUNT.cpu().tolist()), flush=True)
    grasp.sim.close()
