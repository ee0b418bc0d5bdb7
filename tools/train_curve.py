"""Reward / episode-length / success trend of a task over N epochs (sanity: does the policy learn on this engine?).
usage: python tools/train_curve.py <Task> <num_envs> <epochs> [every]"""
import sys, time, yaml, torch
sys.path.insert(0, '.')
from seqdex_amd.config import TASK_CFG, TRAIN_CFG
from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
from seqdex_amd.a2c_agent import A2CAgent
import importlib
task_name, n, epochs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 50
mod = {"BlockAssemblyGraspSim": "block_assembly_grasp_sim", "BlockAssemblyOrient": "block_assembly_orient",
       "BlockAssemblyInsertSim": "block_assembly_insert_sim", "BlockAssemblySearch": "block_assembly_search"}[task_name]
cls = getattr(importlib.import_module("seqdex_amd.tasks." + mod), task_name)
cfg = yaml.safe_load(open('seqdex_amd/' + TASK_CFG[task_name])); cfg['env']['numEnvs'] = n
tr = yaml.safe_load(open('seqdex_amd/' + TRAIN_CFG[task_name]))
task = cls(cfg, device_type='cuda', device_id=0, headless=True)
env = RLgamesVecTaskPython(task, 'cuda:0')
tr['params']['config'].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=22)
agent = A2CAgent('run', tr['params'])
t0 = time.time()
for ep in range(epochs):
    r = agent.train_epoch()
    if (ep + 1) % every == 0 or ep == 0:
        torch.cuda.synchronize()
        print('epoch %4d  game reward %9.3f  game length %6.1f  step reward mean %8.4f  success_buf mean %.3f  kl %.4f  lr %.2e  %.0f env-steps/s'
              % (ep + 1, agent.game_rewards.get_mean()[0], agent.game_lengths.get_mean()[0], float(task.rew_buf.mean()),
                 float(task.extras['success_buf'].float().mean()), float(r[8][0]), r[9], n * 8 * (ep + 1) / (time.time() - t0)))
