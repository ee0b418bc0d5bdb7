"""Flag / config semantics of the reference launcher for the BlockAssemblyGraspSim path: `get_args` (utils/config.py:220-327,
plus the gymutil flags the task actually consumes), `retrieve_cfg` (CF:62-90), `load_cfg` (CF:94-182 — numEnvs /
episodeLength / seed / max_iterations / checkpoint overrides) and `set_seed` (CF:35-59).  Only the RLgames branch that
train_rlgames.py takes (TR:36,56) is reproduced; flags that the reference parses but never applies (--minibatch_size,
--steps_num, --horovod, --rl_device: TR never reads them, VR:33) are accepted and ignored the same way."""
import argparse
import os
import random

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
TASK_CFG = {"BlockAssemblyGraspSim": "cfg/allegro_hand_block_assembly_grasp_sim.yaml",                      # CF:63-64
            "BlockAssemblyOrient": "cfg/allegro_hand_block_assembly_orient.yaml",                               # CF:75-76
            "BlockAssemblyInsertSim": "cfg/allegro_hand_block_assembly_insert_sim.yaml",                        # CF:69-70
            "BlockAssemblySearch": "cfg/allegro_hand_block_assembly_search.yaml"}                               # CF:72-73
TRAIN_CFG = {"BlockAssemblyGraspSim": "cfg/lego/ppo_continuous_grasp.yaml",                                 # TR:44-47
             "BlockAssemblyOrient": "cfg/lego/ppo_continuous_grasp.yaml",
             "BlockAssemblyInsertSim": "cfg/lego/ppo_continuous_insert.yaml",                               # TR:48-49
             "BlockAssemblySearch": "cfg/lego/ppo_continuous_grasp.yaml"}                                   # TR:45-47


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--test", action="store_true", default=False)
    p.add_argument("--play", action="store_true", default=False)
    p.add_argument("--resume", type=int, default=0)
    p.add_argument("--checkpoint", type=str, default="Base")
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--task", type=str, default="BlockAssemblyGraspSim")
    p.add_argument("--task_type", type=str, default="Python")
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--logdir", type=str, default="logs/")
    p.add_argument("--experiment", type=str, default="Base")
    p.add_argument("--metadata", action="store_true", default=False)
    p.add_argument("--cfg_train", type=str, default="Base")
    p.add_argument("--cfg_env", type=str, default="Base")
    p.add_argument("--num_envs", type=int, default=0)
    p.add_argument("--episode_length", type=int, default=0)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int, default=0)
    p.add_argument("--steps_num", type=int, default=-1)
    p.add_argument("--minibatch_size", type=int, default=-1)
    p.add_argument("--randomize", action="store_true", default=False)
    p.add_argument("--torch_deterministic", action="store_true", default=False)
    p.add_argument("--algo", type=str, default="lego")
    p.add_argument("--model_dir", type=str, default="")
    # the gymutil.parse_arguments flags the reference relies on
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--physx", action="store_true", default=True)
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=0)
    args = p.parse_args(argv)
    if args.pipeline.lower() != "gpu":
        raise SystemExit("seqdex_amd has no CPU pipeline (the product path is HIP only); use --pipeline gpu")
    args.device_id = int(args.sim_device.split(":")[1]) if ":" in args.sim_device else 0
    args.device = "cuda"
    if args.test:
        args.play, args.train = True, False
    elif args.play:
        args.train = False
    else:
        args.train = True
    if args.checkpoint == "Base":                                                                           # TR:38-39
        args.checkpoint = ""
    if args.task not in TASK_CFG:
        raise SystemExit("Unrecognized task!\nTask should be one of: %s (the BlockAssembly chain tasks built here)"
                         % sorted(TASK_CFG))                                                                # CF:26-28
    if args.cfg_env == "Base":
        args.cfg_env = os.path.join(HERE, TASK_CFG[args.task])
    if args.cfg_train == "Base":
        args.cfg_train = os.path.join(HERE, TRAIN_CFG[args.task])
    if args.logdir == "logs/":
        args.logdir = "logs/" + args.task
    return args


def load_cfg(args):
    with open(args.cfg_train) as f:
        cfg_train = yaml.safe_load(f)
    with open(args.cfg_env) as f:
        cfg = yaml.safe_load(f)
    if args.num_envs > 0:                                                                                   # CF:101-102
        cfg["env"]["numEnvs"] = args.num_envs
    if args.episode_length > 0:                                                                             # CF:104-105
        cfg["env"]["episodeLength"] = args.episode_length
    cfg["name"] = args.task
    cfg["headless"] = args.headless
    if "task" in cfg:                                                                                       # CF:111-116
        cfg["task"]["randomize"] = args.randomize or cfg["task"].get("randomize", False)
    else:
        cfg["task"] = {"randomize": False}
    exp_name = cfg_train["params"]["config"]["name"]                                                        # CF:120-138
    if args.experiment != "Base":
        exp_name = args.experiment
    cfg_train["params"]["config"]["name"] = exp_name
    if args.resume > 0:                                                                                     # CF:141-142
        cfg_train["params"]["load_checkpoint"] = True
    if args.checkpoint:                                                                                     # CF:144-145
        cfg_train["params"]["load_path"] = args.checkpoint
    if args.max_iterations > 0:                                                                             # CF:148-149
        cfg_train["params"]["config"]["max_epochs"] = args.max_iterations
    cfg_train["params"]["config"]["num_actors"] = cfg["env"]["numEnvs"]                                     # CF:151
    seed = cfg_train["params"].get("seed", -1)                                                              # CF:153-157
    if args.seed is not None:
        seed = args.seed
    cfg["seed"] = seed
    cfg["args"] = args
    return cfg, cfg_train, args.logdir


def set_seed(seed, torch_deterministic=False):
    """CF:35-59: -1 with determinism -> 42; -1 otherwise -> random; seeds random / numpy / torch."""
    import torch
    if seed == -1 and torch_deterministic:
        seed = 42
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    return seed
