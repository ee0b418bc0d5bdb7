"""Can BlockAssemblyInsertSim learn to insert from LEARNED grasp states once a transition value has filtered them (the second forward pass
of the bi-optimisation loop, scripts/bi_optimization.py:115-121)?

  0  insert policy + transition value from synthetic grasp states (evaluation.prepare_tvalue_and_insert_policy)
  1  grasp policy trained under that value's gate (evaluation.train_grasp_policy)
  2  the grasp policy played until `want` states per brick-type group passed the gate `gate` (or `max_steps` env steps per env)
  3  where those states start InsertSim from: distance / rotation error to the site right after the reset
  4  the insert policy trained on those states (from scratch, and fine-tuned from stage 0's checkpoint), `epochs` epochs each; outcome counts
     and the distribution of the distance / rotation error over the envs every `every` epochs

brick-type groups without a harvested state start from synthetic states (named; the success rates are given for the REAL groups alone too)

usage: python tools/insert_refit_probe.py N epochs every [--gate 0.8] [--want 64] [--max_steps 16000] [--skip_scratch]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.a2c_agent import A2CAgent  # noqa: E402
from seqdex_amd.config import TASK_CFG, TRAIN_CFG, set_seed  # noqa: E402
from seqdex_amd.scripts.evaluation import main_rlgames, prepare_tvalue_and_insert_policy, train_grasp_policy  # noqa: E402
from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402


def opt(name, default, cast=float):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def aux_line(task):
    aux = task.sim.INSERT_AUX.cpu().numpy()
    d, r = aux[:, 3], aux[:, 4]
    return ("dist mm p10/p50/p90 %.1f/%.1f/%.1f  <20mm %.3f  rot p10/p50/p90 %.2f/%.2f/%.2f  <0.2 %.3f  both %.4f"
            % (*(np.quantile(d, [0.1, 0.5, 0.9]) * 1e3), float((d < 0.02).mean()), *np.quantile(r, [0.1, 0.5, 0.9]), float((r < 0.2).mean()),
               float(((d < 0.02) & (r < 0.2)).mean())))


def train_insert(n, states, epochs, every, restore=""):
    set_seed(22)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyInsertSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyInsertSim"])))
    task = BlockAssemblyInsertSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, grasp_states=states, synthetic_fallback=True)
    real = torch.tensor([(e % 8) not in task.synthetic_groups for e in range(n)], device="cuda:0")
    print("  grasp states: %s" % task.grasp_states_source, flush=True)
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
    agent = A2CAgent("run", tr["params"])
    if restore:
        agent.restore(restore)
        agent.epoch_num = 0
    # ---- stage 3: where the states start the task from
    env.reset()
    task.sim.compute_observations()
    torch.cuda.synchronize()
    print("  start of the episodes: " + aux_line(task), flush=True)
    t0 = time.time()
    last = [0, 0]
    for ep in range(epochs):
        agent.train_epoch()
        if (ep + 1) % every == 0 or ep == 0:
            torch.cuda.synchronize()
            c = task.sim.TV_COUNT.cpu().tolist()
            ds, df = c[0] - last[0], c[1] - last[1]
            last = c
            sb = task.extras["success_buf"].float()
            print("  epoch %5d  game reward %8.3f  len %6.1f  outcomes(succ, fail) %s  success rate since last line %.4f  last episode of the envs of REAL groups %.4f  %s"
                  % (ep + 1, agent.game_rewards.get_mean()[0], agent.game_lengths.get_mean()[0], c, ds / max(ds + df, 1), float(sb[real].mean()), aux_line(task)), flush=True)
    out = {"wall_s": time.time() - t0, "restored": bool(restore), "outcomes": task.sim.TV_COUNT.cpu().tolist(),
           "success_rate_last_interval": ds / max(ds + df, 1), "insert_success_buf_mean": float(task.extras["success_buf"].float().mean()),
           "insert_success_buf_mean_real_groups": float(task.extras["success_buf"].float()[real].mean()), "synthetic_groups": task.synthetic_groups}
    agent.ppo.close()
    task.sim.close()
    return out


if __name__ == "__main__":
    n, epochs, every = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    gate, want, max_steps = opt("--gate", 0.8), opt("--want", 64, int), opt("--max_steps", 16000, int)
    work = tempfile.mkdtemp(prefix="sdx_insert_refit_")
    tv, ipath, ist = prepare_tvalue_and_insert_policy(n, 1500, save_to=os.path.join(work, "insert"))
    print("stage 0:", json.dumps(ist, default=float), flush=True)
    gpath, gtask, gst = train_grasp_policy(n, 1500, save_to=os.path.join(work, "grasp"), tvalue_state=tv)
    gtask.sim.close()
    print("stage 1:", json.dumps(gst), flush=True)
    grasp, st = main_rlgames("BlockAssemblyGraspSim", n, policy_path=gpath, tvalue_state=tv, steps=160,
                             until=lambda t: int(t.sim.HARVEST_COUNT.min()) >= want, max_steps=max_steps, task_kwargs={"harvest_tvalue_gate": gate})
    st["grasp_states_harvested_per_type"] = grasp.sim.HARVEST_COUNT.cpu().tolist()
    print("stage 2:", json.dumps(st), flush=True)
    states = grasp.grasp_terminal_states()
    grasp.sim.close()
    res = {}
    print("insert policy of stage 0 fine-tuned on the harvested states:", flush=True)
    res["fine_tuned"] = train_insert(n, states, epochs, every, restore=ipath)
    if "--skip_scratch" not in sys.argv:
        print("insert policy trained from scratch on the harvested states:", flush=True)
        res["from_scratch"] = train_insert(n, states, epochs, every)
    print(json.dumps(res))
