"""Launcher with the reference's command line (train_rlgames.py:33-94):

    python -m seqdex_amd.train_rlgames --task=BlockAssemblyGraspSim --num_envs=1024 [--seed 22] [--max_iterations N]
                                       [--checkpoint path --play]          # one process per GPU under torchrun

args -> cfg (utils/config.py semantics) -> task + VecTask adapter (utils/parse_task.py:162-178) -> A2CAgent (the
rl_games Runner is replaced by seqdex_amd.a2c_agent, same YAML schema)."""
import os

import yaml


def build(args, task_kwargs=None, minibatch_size=0, config_overrides=None):
    """args (config.get_args) -> (task, env, agent, logdir, rank): everything main() does before agent.train() / agent.play().
    config_overrides: keys written into the YAML's params.config before the agent is built (e.g. mixed_precision: True, rl_games' own key)"""
    import torch
    from .a2c_agent import A2CAgent
    from .config import load_cfg, set_seed
    from .tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from .tasks.block_assembly_orient import BlockAssemblyOrient
    from .tasks.block_assembly_insert_sim import BlockAssemblyInsertSim
    from .tasks.block_assembly_search import BlockAssemblySearch
    from .vec_task_rlgames import RLgamesVecTaskPython
    args.algo = "lego"                                                                    # TR:36
    args.task_type = "RLgames"                                                            # TR:56
    print("Loading config: ", args.cfg_train)
    cfg, cfg_train, logdir = load_cfg(args)
    seed = args.seed if args.seed is not None else 22                                     # TR:62-65
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(args.device_id)))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))       # RCCL over xGMI
    cfg["env"]["test"] = args.play                                                        # TR:68
    set_seed(seed + rank, args.torch_deterministic)                                       # TR:70 (+ rank, App. C)
    task_cls = {"BlockAssemblyGraspSim": BlockAssemblyGraspSim, "BlockAssemblyOrient": BlockAssemblyOrient,
                "BlockAssemblyInsertSim": BlockAssemblyInsertSim, "BlockAssemblySearch": BlockAssemblySearch}[args.task]   # eval(args.task), PT:162
    task = task_cls(cfg, None, None, "cuda", local_rank, True, seed=seed + rank, **(task_kwargs or {}))   # PT:162-170
    env = RLgamesVecTaskPython(task, args.rl_device)                                      # PT:178
    rl = cfg_train                                                                        # TR:78-85
    if minibatch_size:     # programmatic override (the reference parses --minibatch_size but never applies it, CF:43)
        rl["params"]["config"]["minibatch_size"] = minibatch_size
        rl["params"]["config"]["central_value_config"]["minibatch_size"] = minibatch_size
    rl["params"]["config"].update(config_overrides or {})
    rl["params"]["config"]["name"] = args.task
    rl["params"]["config"]["num_actors"] = env.num_environments
    rl["params"]["seed"] = seed
    rl["params"]["config"]["seed"] = seed
    rl["params"]["config"]["env_config"]["seed"] = seed
    rl["params"]["config"]["vec_env"] = env
    rl["params"]["config"]["env_info"] = env.get_env_info()
    rl["params"]["config"]["multi_gpu"] = world > 1
    agent = A2CAgent("run", rl["params"])                                                 # TR:88-94 (Runner.run -> agent.train)
    if rl["params"].get("load_path"):
        agent.restore(rl["params"]["load_path"])
    return task, env, agent, logdir, rank


def main(argv=None):
    from .config import get_args
    args = get_args(argv)
    task, env, agent, logdir, rank = build(args)
    if args.train:
        agent.train()
        if rank == 0:
            os.makedirs(os.path.join(logdir, "nn"), exist_ok=True)
            agent.save(os.path.join(logdir, "nn", "last_%s_ep_%d" % (args.task, agent.epoch_num)))
    else:
        agent.play(int(agent.config.get("player", {}).get("games_num", 1)))


if __name__ == "__main__":
    main()
