"""does PPO learn BlockAssemblyGraspSim on this engine?  One training run at N envs, horizon 8, with a chosen minibatch
size (the shipped 4 -> the persistent update kernel; > 8 -> the MFMA large-minibatch path) and the shipped adaptive learning rate (kl
threshold 0.02) or a fixed one.  Every `every` epochs: game reward (rl_games' mean over finished episodes), episode length, success_buf, the
grasp terminal states harvested per brick-type group (T-value gate opened: the PHYSICAL criterion, GS:1404-1405), kl, lr, throughput.
The reference's only published datum: checkpoint ..._ep_19000_rew_1530.9819.pth (README.md:90) = an episode reward of 1 531 after 19 000
epochs (the reward's ceiling is 20 per step while the brick is held 20 cm up, GS:1706-1776).
usage: python tools/grasp_train_r5.py N epochs every minibatch [lr|adaptive] [outfile] [max_seconds]"""
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


def main():
    n, epochs, every, mb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    lr = sys.argv[5] if len(sys.argv) > 5 else "adaptive"
    out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(ROOT, "gpurun_out", "grasp_train_r5.txt")
    max_s = float(sys.argv[7]) if len(sys.argv) > 7 else 1e9
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd", TRAIN_CFG["BlockAssemblyGraspSim"])))
    pc = tr["params"]["config"]
    pc["minibatch_size"] = mb
    pc["central_value_config"]["minibatch_size"] = mb
    if lr != "adaptive":
        pc["lr_schedule"] = "constant"
        pc["learning_rate"] = float(lr)
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=16)
    task.sim.set_tvalue_weights(open_gate_tvalue())
    env = RLgamesVecTaskPython(task, "cuda:0")
    pc.update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
    agent = A2CAgent("run", tr["params"])
    log = open(out, "w")
    head = "# BlockAssemblyGraspSim, %d envs, horizon 8, minibatch %d (%s update path), %d mini-epochs, lr %s (kl threshold %s), %d epochs" % (
        n, mb, "persistent rank-4" if mb <= 8 else "large-minibatch MFMA", agent.mini_epochs_num, lr, pc.get("kl_threshold"), epochs)
    print(head, flush=True); log.write(head + "\n")
    t0 = time.time()
    rew_max = -1e9
    for ep in range(epochs):
        r = agent.train_epoch()
        if (ep + 1) % every == 0 or ep == 0:
            torch.cuda.synchronize()
            hc = task.sim.HARVEST_COUNT.cpu().numpy().tolist()
            gr = agent.game_rewards.get_mean()[0]
            rew_max = max(rew_max, gr)
            line = ("epoch %6d  game reward %9.3f  game length %6.1f  step reward %7.3f  success_buf %.3f  harvested/type %s  kl %.4f  lr %.2e  %.0f env-steps/s"
                    % (ep + 1, gr, agent.game_lengths.get_mean()[0], float(task.rew_buf.mean()), float(task.extras["success_buf"].float().mean()),
                       hc, float(r[8][0]), r[9], n * 8 * (ep + 1) / (time.time() - t0)))
            print(line, flush=True)
            log.write(line + "\n"); log.flush()
            if time.time() - t0 > max_s:
                break
    s = task.sim
    tail = json.dumps({"epochs_run": ep + 1, "n_envs": n, "minibatch": mb, "lr": lr, "wall_s": time.time() - t0, "final_game_reward": agent.game_rewards.get_mean()[0],
                       "max_logged_game_reward": rew_max, "harvested_per_type": s.HARVEST_COUNT.cpu().numpy().tolist(), "contact_stats": s.CONTACT_STATS.cpu().numpy().tolist()})
    print(tail); log.write(tail + "\n"); log.close()
    t = agent.ppo.t
    np.savez_compressed(out.replace(".txt", "_weights.npz"), ac=t["AC_PARAMS"].cpu().numpy().astype(np.float16), cv=t["CV_PARAMS"].cpu().numpy().astype(np.float16),
                        rms_mean=t["CV_RMS_MEAN"].cpu().numpy(), rms_var=t["CV_RMS_VAR"].cpu().numpy())


if __name__ == "__main__":
    main()
