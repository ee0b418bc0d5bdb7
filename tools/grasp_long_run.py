"""one long BlockAssemblyGraspSim training run with the SHIPPED schedule (minibatch 4, horizon 8, adaptive LR) at N envs:
game reward / episode length / success_buf / harvested grasp terminal states per brick-type group every `every` epochs, to compare with
the reference's only published datum (checkpoint `..._ep_19000_rew_1530.9819.pth`, README.md:90).  The harvest gate's T-value (GS:1406) is
opened (output bias: T = 1 for every orientation) so that `harvested` counts the PHYSICAL criterion - brick carried to y < 0 and still between
the fingers at the episode's end (GS:1404-1405).  Saves the final network weights and the harvested grasp states under gpurun_out/.
usage: python tools/grasp_long_run.py [N] [epochs] [every] [outdir]"""
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd import _abi  # noqa: E402
from seqdex_amd.a2c_agent import A2CAgent  # noqa: E402
from seqdex_amd.config import TASK_CFG, TRAIN_CFG  # noqa: E402
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim  # noqa: E402
from seqdex_amd.tvalue_trainer import LAYERS  # noqa: E402
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython  # noqa: E402

def open_gate_tvalue():
    """GraspInsertTValue weights whose output is (0, 10) for every input: sigmoid(10) = 1 > 0.8"""
    parts = []
    for i, (name, out, inn) in enumerate(LAYERS):
        parts.append(np.zeros(out * inn, np.float32))
        b = np.zeros(out, np.float32)
        if i == len(LAYERS) - 1:
            b[1] = 10.0
        parts.append(b)
    flat = np.concatenate(parts)
    assert flat.size == _abi.TV_PARAMS
    return flat


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    outdir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "r3train")
    os.makedirs(outdir, exist_ok=True)


    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyGraspSim"])))
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=16)
    task.sim.set_tvalue_weights(open_gate_tvalue())
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
    agent = A2CAgent("run", tr["params"])
    log = open(os.path.join(outdir, "grasp_long_run.txt"), "w")
    t0 = time.time()
    best = -1e9
    for ep in range(epochs):
        r = agent.train_epoch()
        if (ep + 1) % every == 0 or ep == 0:
            torch.cuda.synchronize()
            hc = task.sim.HARVEST_COUNT.cpu().numpy().tolist()
            line = ("epoch %5d  game reward %9.3f  game length %6.1f  success_buf %.3f  harvested/type %s  kl %.4f  lr %.2e  contacts max %d over %d  %.0f env-steps/s"
                    % (ep + 1, agent.game_rewards.get_mean()[0], agent.game_lengths.get_mean()[0], float(task.extras["success_buf"].float().mean()),
                       hc, float(r[8][0]), r[9], int(task.sim.CONTACT_STATS[0]), int(task.sim.CONTACT_STATS[1]), n * 8 * (ep + 1) / (time.time() - t0)))
            print(line, flush=True)
            log.write(line + "\n"); log.flush()
    t = agent.ppo.t
    np.savez_compressed(os.path.join(outdir, "grasp_policy_weights.npz"), ac=t["AC_PARAMS"].cpu().numpy().astype(np.float16),
                        cv=t["CV_PARAMS"].cpu().numpy().astype(np.float16), rms_mean=t["CV_RMS_MEAN"].cpu().numpy(), rms_var=t["CV_RMS_VAR"].cpu().numpy())
    s = task.sim
    np.savez_compressed(os.path.join(outdir, "grasp_terminal_states.npz"), obj=s.HARVEST_OBJ.cpu().numpy()[:, :64], hand=s.HARVEST_HAND.cpu().numpy()[:, :64],
                        count=s.HARVEST_COUNT.cpu().numpy())
    log.write(json.dumps({"epochs": epochs, "n_envs": n, "wall_s": time.time() - t0, "final_game_reward": agent.game_rewards.get_mean()[0],
                          "harvested_per_type": s.HARVEST_COUNT.cpu().numpy().tolist(), "contact_stats": s.CONTACT_STATS.cpu().numpy().tolist()}) + "\n")
    log.close()


if __name__ == "__main__":
    main()
