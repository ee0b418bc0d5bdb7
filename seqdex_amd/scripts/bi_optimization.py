"""Outer loop of the policy chain, after the reference's scripts/bi_optimization.py:36-124:

    forward initialisation : train the sub-policies in chain order, each starting from what its predecessor produced
    backward fine-tuning   : run the LAST policy again to collect success / failure data, fit the transition value on it, give it to
                             the policy before it and fine-tune that one, and so on towards the front of the chain:
                             InsertSim -> fit -> GraspSim -> fit -> Orient -> fit (bi_optimization.py:120-124; every fit is the
                             4-input GraspInsertTValue on the camera-frame quaternions that task logged at its episode ends)

for the chain BlockAssemblySearch -> BlockAssemblyOrient -> BlockAssemblyGraspSim -> BlockAssemblyInsertSim.  Where the reference hands data over through files (pickles of terminal states, an HDF5 file of
quaternions, .pth checkpoints), the stages here hand over device tensors of the same content; checkpoints are still written.

    python -m seqdex_amd.scripts.bi_optimization --tasks BlockAssembly [--rounds 10] [--epochs N] [--tvalue_rollout 10000]
"""
import argparse
import os

import torch

from ..config import get_args
from ..train_rlgames import build
from ..tvalue_trainer import TValue_Trainer, flat_from_state_dict


def main_rlgames(task, num_envs, use_t_value=False, policy_path="", max_iterations=0, task_kwargs=None, tvalue_state=None, keep=False,
                 minibatch_size=0):
    """one training run of `task` (bi_optimization.py:36-104).  Returns (checkpoint path, task object or None).  use_t_value marks the
    backward-pass runs whose purpose is the task's success / failure datasets (they are always logged on the device here)."""
    argv = ["--task=%s" % task, "--num_envs=%d" % num_envs, "--headless"]
    if max_iterations:
        argv.append("--max_iterations=%d" % max_iterations)
    if policy_path:
        argv.append("--checkpoint=%s" % policy_path)
    args = get_args(argv)
    args.use_t_value = use_t_value
    task_obj, env, agent, logdir, rank = build(args, task_kwargs, minibatch_size)
    if tvalue_state is not None:
        task_obj.sim.set_tvalue_weights(flat_from_state_dict(tvalue_state).numpy())
    if policy_path:
        agent.epoch_num = 0        # every run of the outer loop trains max_iterations MORE epochs (rl_games would resume the counter)
    agent.train()
    os.makedirs(os.path.join(logdir, "nn"), exist_ok=True)
    path = os.path.join(logdir, "nn", "%s" % task)                                      # runner.nn_dir/<task>.pth, bi_optimization.py:104
    agent.save(path)
    agent.ppo.close()
    if not keep:
        task_obj.sim.close()
        task_obj = None
    torch.cuda.synchronize()
    return path + ".pth", task_obj


def transition_value_trainer(task_obj, rollout, state_dict=None, seed=0):
    """bi_optimization.py:106-109 with the task's device rings in place of ./intermediate_state/<task>_datasets.hdf5.
    Returns the fitted state_dict, or `state_dict` unchanged when the run logged too few rows of either class."""
    try:
        tr = TValue_Trainer.from_task(task_obj, seed=seed)
    except ValueError as ex:
        print("transition_value_trainer: skipped (%s)" % ex)
        return state_dict
    tr.init_TValue_function(type(task_obj).__name__, rollout, state_dict=state_dict)
    tr.train_rollout(verbose=True)
    sd = tr.state_dict()
    tr.close()
    return sd


def block_assembly(rounds=10, num_envs=512, epochs=0, tvalue_rollout=10000, insert_minibatch=0):
    """insert_minibatch: override of the insert schedule's minibatch_size 4096 for runs with fewer than 512 envs"""
    tv = None
    paths = {}
    for i in range(rounds):
        # ---- forward initialisation (bi_optimization.py:115-118)
        paths["search"], search = main_rlgames("BlockAssemblySearch", min(num_envs, 128), max_iterations=epochs,
                                               policy_path=paths.get("search", ""), keep=True)
        dug = search.pile_terminal_states()                                               # hand-off SE:1323-1353 -> Orient's saved piles
        search.sim.close()
        paths["orient"], orient = main_rlgames("BlockAssemblyOrient", num_envs, max_iterations=epochs, policy_path=paths.get("orient", ""),
                                               keep=True, task_kwargs={"initial_piles": dug})
        piles = orient.pile_terminal_states()                                             # hand-off OR:1483-1510 -> GS:412-413
        orient.sim.close()
        paths["grasp"], grasp = main_rlgames("BlockAssemblyGraspSim", num_envs, max_iterations=epochs, policy_path=paths.get("grasp", ""),
                                             tvalue_state=tv, keep=True, task_kwargs={"initial_piles": piles})
        cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
        grasp_states = grasp.grasp_terminal_states() if cnt.min() > 0 else None          # hand-off GS:1447-1450 -> IS:372-375
        grasp.sim.close()
        paths["insert"], _ = main_rlgames("BlockAssemblyInsertSim", num_envs, max_iterations=epochs, policy_path=paths.get("insert", ""),
                                          task_kwargs={"grasp_states": grasp_states}, minibatch_size=insert_minibatch)
        # ---- backward fine-tuning (bi_optimization.py:120-124)
        _, insert = main_rlgames("BlockAssemblyInsertSim", num_envs, use_t_value=True, policy_path=paths["insert"], max_iterations=epochs,
                                 task_kwargs={"grasp_states": grasp_states}, keep=True, minibatch_size=insert_minibatch)
        tv = transition_value_trainer(insert, tvalue_rollout, tv, seed=i)
        insert.sim.close()
        paths["grasp"], grasp = main_rlgames("BlockAssemblyGraspSim", num_envs, use_t_value=True, policy_path=paths["grasp"],
                                             max_iterations=epochs, tvalue_state=tv, keep=True, task_kwargs={"initial_piles": piles})
        tv = transition_value_trainer(grasp, tvalue_rollout, tv, seed=100 + i)            # bi_optimization.py:122: fit on GraspSim's own outcomes
        grasp.sim.close()
        paths["orient"], orient = main_rlgames("BlockAssemblyOrient", min(num_envs, 128), use_t_value=True, policy_path=paths["orient"],
                                               max_iterations=epochs, tvalue_state=tv, keep=True, task_kwargs={"initial_piles": dug})   # :123
        tv = transition_value_trainer(orient, tvalue_rollout, tv, seed=200 + i)           # bi_optimization.py:124
        orient.sim.close()
        print("bi-optimisation round %d done: %s" % (i, paths))
    return paths, tv


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--tasks", type=str, default="BlockAssembly")
    p.add_argument("--rounds", type=int, default=10)
    p.add_argument("--num_envs", type=int, default=512)
    p.add_argument("--epochs", type=int, default=0, help="max_iterations of every training run (0 = the YAML's max_epochs)")
    p.add_argument("--tvalue_rollout", type=int, default=10000)
    a = p.parse_args()
    if a.tasks != "BlockAssembly":
        raise Exception("Unrecognized task!")                                           # bi_optimization.py:141-143 (ToolPositioning: not built)
    block_assembly(a.rounds, a.num_envs, a.epochs, a.tvalue_rollout)
