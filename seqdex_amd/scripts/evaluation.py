"""Sequential --play evaluation of the policy chain, after the reference's scripts/evaluation.py:36-119: every sub-policy is restored
from its checkpoint and played (mean action, no update) on its own task, in chain order, each stage starting from what the stage
before produced (Search's dug-out piles -> Orient; Orient's harvested piles -> GraspSim; GraspSim's grasp terminal states -> InsertSim).  Checkpoints: files written by A2CAgent.save (rl_games' layout) or
by rl_games itself.

    python -m seqdex_amd.scripts.evaluation --tasks BlockAssembly --orient ck1.pth --grasp ck2.pth --insert ck3.pth [--games 512]
"""
import argparse

import torch

from ..config import get_args
from ..train_rlgames import build


def main_rlgames(task, num_envs, play=True, use_t_value=False, policy_path="", games=0, task_kwargs=None, minibatch_size=0):
    """evaluation.py:36-103 for one stage.  Returns (mean episode reward, mean episode length, task object)."""
    argv = ["--task=%s" % task, "--num_envs=%d" % num_envs, "--headless", "--play"]
    if policy_path:
        argv.append("--checkpoint=%s" % policy_path)
    args = get_args(argv)
    args.use_t_value = use_t_value
    task_obj, env, agent, logdir, rank = build(args, task_kwargs, minibatch_size)
    agent.play(games or num_envs)
    torch.cuda.synchronize()
    rew, length = float(agent.game_rewards.get_mean()[0]), float(agent.game_lengths.get_mean()[0])
    agent.ppo.close()
    return rew, length, task_obj


def block_assembly(orient_path, grasp_path, insert_path, num_envs=512, games=0, insert_minibatch=0, search_path=None):
    out = {}
    dug = None
    if search_path is not None:
        r, l, search = main_rlgames("BlockAssemblySearch", min(num_envs, 128), use_t_value=True, policy_path=search_path, games=games)
        dug = search.pile_terminal_states()
        out["BlockAssemblySearch"] = dict(reward=r, length=l, search_success_rate=float(search.extras["success_buf"].float().mean()),
                                          piles_handed_on=0 if dug is None else int(dug.shape[1]))
        search.sim.close()
    r, l, orient = main_rlgames("BlockAssemblyOrient", num_envs, use_t_value=True, policy_path=orient_path, games=games,
                                task_kwargs={"initial_piles": dug})
    piles = orient.pile_terminal_states()
    out["BlockAssemblyOrient"] = dict(reward=r, length=l, piles_handed_on=0 if piles is None else int(piles.shape[1]))
    orient.sim.close()
    r, l, grasp = main_rlgames("BlockAssemblyGraspSim", num_envs, use_t_value=True, policy_path=grasp_path, games=games,
                               task_kwargs={"initial_piles": piles})
    cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
    states = grasp.grasp_terminal_states() if cnt.min() > 0 else None
    out["BlockAssemblyGraspSim"] = dict(reward=r, length=l, grasp_states_handed_on=int(cnt.sum()))
    grasp.sim.close()
    r, l, insert = main_rlgames("BlockAssemblyInsertSim", num_envs, use_t_value=True, policy_path=insert_path, games=games,
                                task_kwargs={"grasp_states": states}, minibatch_size=insert_minibatch)
    out["BlockAssemblyInsertSim"] = dict(reward=r, length=l, insert_success_rate=float(insert.extras["success_buf"].float().mean()),
                                         grasp_states=insert.grasp_states_source)
    insert.sim.close()
    for k, v in out.items():
        print(k, v)
    return out


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--tasks", type=str, default="BlockAssembly")
    p.add_argument("--search", type=str, default=None)
    p.add_argument("--orient", type=str, default="")
    p.add_argument("--grasp", type=str, default="")
    p.add_argument("--insert", type=str, default="")
    p.add_argument("--num_envs", type=int, default=512)
    p.add_argument("--games", type=int, default=0)
    a = p.parse_args()
    if a.tasks != "BlockAssembly":
        raise Exception("Unrecognized task!")
    block_assembly(a.orient, a.grasp, a.insert, a.num_envs, a.games, search_path=a.search)
