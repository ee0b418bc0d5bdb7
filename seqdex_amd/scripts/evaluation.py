"""Chained evaluation of the BlockAssembly sub-policies, after the reference's scripts/evaluation.py:36-119: every stage is played (no
update) with the transition value switched on, and leaves the terminal states the next stage starts from:

    BlockAssemblyOrient    --pile states of episodes that end with the target brick reachable (OR:1463-1488)-->
    BlockAssemblyGraspSim  --grasp terminal states: brick carried to the insertion side, still in the hand, T-value > 0.8 (GS:1404-1417)-->
    BlockAssemblyInsertSim   (every reset draws the brick and the hand from those states, IS:372-375,1449-1456)

Where the reference hands the states over through pickles under ./intermediate_state/, the stages here hand over device tensors with the
same content (the pickle forms exist too: seqdex_amd/piles.py, BlockAssemblyGraspSim.save_grasp_terminal_states).  BASELINE.json
configs[2] is this chain at num_envs = 1024 on one GPU; tools/bench_config3.py times it, tests/test_gpu_chain.py checks the hand-offs.

    python -m seqdex_amd.scripts.evaluation --tasks BlockAssembly [--num_envs 512] [--search s.pth] --orient o.pth --grasp g.pth --insert i.pth [--games 512]
    python -m seqdex_amd.scripts.evaluation --mode chain_learned [--num_envs 1024]
    python -m seqdex_amd.scripts.evaluation --mode chain [--tvalue tv.pt] [--synthetic_fallback --orient_tvalue_gate 0.5 --grasp_tvalue_gate 0.28]
"""
import argparse
import os
import time

import numpy as np
import torch
import yaml

from ..a2c_agent import A2CAgent
from ..config import TASK_CFG, TRAIN_CFG, set_seed
from ..vec_task_rlgames import RLgamesVecTaskPython

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Orient's T-value gate in the chain benchmark / test when the transition value comes from a stage 0 of a thousand epochs: a descending
# ladder, the last rung opens the gate (block_assembly_chain, stage 1).  The reference's threshold is 0.99 (OR:1203).
CHAIN_ORIENT_GATES = (0.5, 0.4, 0.3, 0.28, 0.0)
CHAIN_GRASP_GATES = (0.28, 0.0)         # GraspSim's harvest gate in the same setting (reference: 0.8, GS:1406)


def _task_class(name):
    import importlib
    mod = {"BlockAssemblyGraspSim": "block_assembly_grasp_sim", "BlockAssemblyOrient": "block_assembly_orient",
           "BlockAssemblyInsertSim": "block_assembly_insert_sim", "BlockAssemblySearch": "block_assembly_search"}[name]
    return getattr(importlib.import_module("seqdex_amd.tasks." + mod), name)


def main_rlgames(task, num_envs, play=True, use_t_value=True, policy_path="", steps=None, task_kwargs=None, tvalue_state=None,
                 controller=None, seed=22, until=None, max_steps=None):
    """one stage of scripts/evaluation.py:36-103: build the task and its agent, restore `policy_path`, play.  `steps` env steps are
    played in horizon-sized chunks (default: one episode + its reset); `until(task)` may end the stage earlier / later (checked after
    every chunk, at most `max_steps`).  `controller(task, step) -> actions [N, 23]` replaces the policy (a scripted stand-in; the
    returned statistics say so).  Returns (task object - the caller closes task.sim -, statistics)."""
    assert play, "the chain evaluation only plays"
    set_seed(seed)        # as the launcher does for every run (TR:70, CF:35-59): RLgamesVecTaskPython.reset draws its noise step from torch's global generator
    cfg = yaml.safe_load(open(os.path.join(ROOT, TASK_CFG[task])))
    cfg["env"]["numEnvs"] = num_envs
    cfg["env"]["test"] = True
    tr = yaml.safe_load(open(os.path.join(ROOT, TRAIN_CFG[task])))
    t_obj = _task_class(task)(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, **(task_kwargs or {}))
    if tvalue_state is not None and use_t_value:
        t_obj.sim.set_tvalue_weights(tvalue_state)
    env = RLgamesVecTaskPython(t_obj, "cuda:0")
    tr["params"]["config"].update(num_actors=num_envs, vec_env=env, env_info=env.get_env_info(), seed=seed, name=task)
    agent = A2CAgent("run", tr["params"])
    if policy_path:
        agent.restore(policy_path)
    horizon = agent.horizon_length
    if steps is None:
        steps = int(t_obj.max_episode_length) + horizon
    max_steps = max_steps or steps
    deterministic = bool(tr["params"]["config"].get("player", {}).get("deterministic", True))
    torch.cuda.synchronize()
    t0 = time.time()
    done = 0
    if controller is not None:
        env.reset()
    while done < max_steps:
        if controller is None:
            agent.play_steps(deterministic)
        else:
            for _ in range(horizon):
                env.step(controller(t_obj, done + _))
        done += horizon
        if done >= steps and (until is None or until(t_obj)):
            break
    torch.cuda.synchronize()
    dt = time.time() - t0
    stats = {"task": task, "num_envs": num_envs, "env_steps": done * num_envs, "steps_per_env": done, "wall_s": dt,
             "env_steps_per_s": done * num_envs / dt, "policy": (policy_path or "random initialisation (seed %d)" % seed) if controller is None
             else "scripted stand-in controller (%s)" % getattr(controller, "__name__", "callable"),
             "success_buf_mean": float(t_obj.extras["success_buf"].float().mean())}
    agent.ppo.close()
    return t_obj, stats


# closure, pinch steps, rise per step, pinch height, pinch offset x / y.  Round 5 (profiles/r5_scripted_lift_scan.txt): the pinch height decides -
# 0.195 above the brick's origin (rounds 3-4) the fingertips close over the studs and 18 % of 1 024 envs hold the brick 5 cm up; at 0.155 they
# close on the brick's body: 54 %
SG_DEFAULTS = [0.9, 8.0, 0.05, 0.155, 0.125, 0.02]


def scripted_grasp_controller(task, step):
    """STAND-IN for a trained BlockAssemblyGraspSim policy (the reference's released checkpoint is from epoch 19 000, README.md:90; nothing of
    that length can be trained inside a test): a hand-written reach - descend - pinch sequence on the task's own action interface
    (GS:1586-1609: a[0:3] x 0.64 = hand-base displacement for the IK, a[3:6] x 0.2 = wrist orientation error, a[7:23] = finger targets scaled
    to the joint limits): hand base above the target brick with the wrist held at the prepare pose's orientation, descend, pinch when
    arrived (at the latest at step 58); once the pinch is complete the hand stops following the brick (a gripped brick moves with the hand)
    and raises it a little.  After step 75 the task itself lifts the hand and carries it to the insertion side with the fingers frozen
    (GS:1600-1609).  SDX_SG_PARAMS="closure,steps,rise,height,x,y" overrides the pinch's end closure / duration, the rise per step, the height
    of the hand base above the brick at the pinch and the pinch point's offset from the hand base (tools/lift_diag.py).  Used by tools/bench_config3.py, tools/bench_config5.py and tests/test_gpu_chain.py so that the grasp stage
    harvests REAL terminal states of this engine; success is far below a trained policy's.  One kernel launch per env step
    (csrc/sdx_task.hip::k_scripted_grasp; round 3 computed the same in ~40 torch operations inside the timed loop)."""
    import ctypes as C
    s = task.sim
    if not hasattr(task, "_sg_close"):
        # per env: the progress values at which the fingers started to close / the hand stopped following the brick; then the parameters
        par = SG_DEFAULTS[:]
        for i, x in enumerate(os.environ.get("SDX_SG_PARAMS", "").split(",")):
            if x.strip():
                par[i] = float(x)
        task._sg_close = torch.cat([torch.full((2 * task.num_envs,), 1e9), torch.tensor(par + [0.0, 0.0])]).to(task.device).contiguous()
        task._sg_act = torch.zeros(task.num_envs, 23, device=task.device)
        s.lib.sdxk_scripted_grasp_actions.restype = C.c_int
        s.lib.sdxk_scripted_grasp_actions.argtypes = [C.c_void_p] * 4
    rc = s.lib.sdxk_scripted_grasp_actions(s.h, C.c_void_p(task._sg_close.data_ptr()), C.c_void_p(task._sg_act.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream(task.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("sdxk_scripted_grasp_actions failed (%d)" % rc)
    return task._sg_act


def scripted_lift_statistics(num_envs=1024, seed=22, piles_per_type=16):
    """can the hand lift a brick?  One episode of BlockAssemblyGraspSim under the scripted controller; per env the
    largest height the target brick reached above its initial one WHILE finger_dist < 0.5 (GS:1164-1165; the reward's own "in the hand"
    threshold, GS:1725), first episode only.  Returns a dict with the fraction of envs that held it >= 5 cm up."""
    from ..tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    cfg = yaml.safe_load(open(os.path.join(ROOT, TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = num_envs
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, piles_per_type=piles_per_type)
    try:
        s, n, dev = task.sim, num_envs, task.device
        seg = torch.as_tensor([s.scene.seg_index(i) for i in range(n)], device=dev)
        ar = torch.arange(n, device=dev)
        task.step(torch.zeros(n, 23, device=dev))                              # the reset step
        z0 = s.INIT_POS[:, 2].clone()
        held = torch.zeros(n, device=dev)
        first = torch.ones(n, dtype=torch.bool, device=dev)
        for step in range(1, int(task.max_episode_length)):
            task.step(scripted_grasp_controller(task, step))
            first &= s.PROGRESS > 1                                            # a reset env is out of its first episode
            dz = s.ROOT.view(n, 142, 13)[ar, seg, 2] - z0
            held = torch.where(first & (s.FINGER_DIST < 0.5), torch.maximum(held, dz), held)
        torch.cuda.synchronize()
        ok = held > 0.05
        return {"n": n, "held_5cm_frac": float(ok.float().mean()), "held_2cm_frac": float((held > 0.02).float().mean()),
                "held_max_m": float(held.max()), "per_type_held_5cm": [int(ok[ar % 8 == g].sum()) for g in range(8)],
                "contact_stats": s.CONTACT_STATS.cpu().tolist()}
    finally:
        task.sim.close()


GRASP_TRAIN_MINIBATCH = 2048     # the minibatch size GraspSim LEARNS with on this engine (profiles/r5_grasp_train_curve_*.txt); the YAML ships 4


def policy_lift_statistics(task, agent, steps=304):
    """what a TRAINED grasp policy physically does (round 5, the check behind the training curves): `steps` deterministic env steps (two
    episodes) of `agent` on `task`; per env the largest height of the target brick above its initial one while finger_dist < 0.5, whether at
    least two fingertip links (thumb among them) carried a net contact force > 0.5 N at that moment (held BY contacts, not by a brick lying
    on the hand), and the brick's speed relative to the hand base then.  Sampled after every env step of the horizon-sized chunks."""
    s, n, dev = task.sim, task.num_envs, task.device
    seg = torch.as_tensor([s.scene.seg_index(i) for i in range(n)], device=dev)
    ar = torch.arange(n, device=dev)
    tips = list(s.scene.fingertip_bodies)
    held = torch.zeros(n, device=dev)
    grip = torch.zeros(n, dtype=torch.bool, device=dev)
    rel = torch.zeros(n, device=dev)
    eps = torch.zeros(n, agent.ppo.cfg.act_dim, device=agent.ppo.device)
    if agent.obs is None:
        agent.obs = agent.env_reset()
        agent.dones = agent.vec_env.task.reset_buf
    for k in range(steps):
        a = agent.ppo.act(k % agent.horizon_length, agent.obs["obs"], agent.obs["states"], agent.dones, eps)
        agent.obs, rew, agent.dones, _ = agent.vec_env.step(a)
        b = s.ROOT.view(n, 142, 13)[ar, seg]
        dz = b[:, 2] - s.INIT_POS[:, 2]
        cf = s.CONTACT.view(n, 165, 3)[:, tips].norm(dim=-1)
        g2 = (cf[:, 3] > 0.5) & ((cf[:, :3] > 0.5).sum(1) >= 1)
        better = (s.FINGER_DIST < 0.5) & (dz > held) & (s.PROGRESS > 1)
        held = torch.where(better, dz, held)
        grip = torch.where(better, g2, grip)
        rel = torch.where(better, (b[:, 7:10] - s.RB[:, s.scene.hand_base_body, 7:10]).norm(dim=-1), rel)
    torch.cuda.synchronize()
    ok = held > 0.05
    return {"steps": steps, "held_5cm_frac": float(ok.float().mean()), "held_15cm_frac": float((held > 0.15).float().mean()), "held_max_m": float(held.max()),
            "of_those_gripped_by_thumb_and_a_finger": float((ok & grip).float().sum() / ok.float().sum().clamp(min=1)),
            "brick_speed_relative_to_hand_at_the_top_mean_m_s": float(rel[ok].mean()) if bool(ok.any()) else None,
            "per_type_held_5cm": [int(ok[ar % 8 == g].sum()) for g in range(8)]}


def train_grasp_policy(n, epochs, seed=22, save_to=None, minibatch=GRASP_TRAIN_MINIBATCH, tvalue_state=None, initial_piles=None, piles_per_type=16, lift_statistics=False,
                       restore=""):
    """A BlockAssemblyGraspSim policy of THIS engine (round 5; the reference's is its released 19 000-epoch checkpoint, README.md:90):
    `epochs` epochs at n envs, horizon 8, 5 mini-epochs, adaptive learning rate as shipped - but minibatches of 2 048 rows instead of the
    shipped 4, with which the shipped schedule does not leave reward 2 (profiles/r5_grasp_train_curve_shipped_minibatch4.txt); episode reward
    ~ 2 000 after 1 500 epochs = 30 s (the reference's checkpoint name says 1 531).  tvalue_state: the transition value that gates the
    harvest of grasp terminal states (GS:1404-1417; None: the gate is opened - what a forward leg of the bi-optimisation loop does before
    any T-value exists).  restore: a grasp checkpoint to go on from (a later round of the bi-optimisation loop fine-tunes the policy under the
    refitted value).  Returns (checkpoint path or "", the task (caller closes task.sim; its rings hold the harvested states), statistics)."""
    from ..tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from ..tvalue_trainer import LAYERS
    set_seed(seed)
    cfg = yaml.safe_load(open(os.path.join(ROOT, TASK_CFG["BlockAssemblyGraspSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, TRAIN_CFG["BlockAssemblyGraspSim"])))
    tr["params"]["config"]["minibatch_size"] = minibatch
    tr["params"]["config"]["central_value_config"]["minibatch_size"] = minibatch
    task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, initial_piles=initial_piles, piles_per_type=piles_per_type)
    if tvalue_state is None:              # open gate: output (0, 10) for every orientation -> sigmoid = 1
        parts = []
        for i, (_, out, inn) in enumerate(LAYERS):
            parts.append(np.zeros(out * inn, np.float32))
            b = np.zeros(out, np.float32)
            if i == len(LAYERS) - 1:
                b[1] = 10.0
            parts.append(b)
        task.sim.set_tvalue_weights(np.concatenate(parts))
    else:
        task.sim.set_tvalue_weights(tvalue_state)
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=seed)
    agent = A2CAgent("run", tr["params"])
    if restore:
        agent.restore(restore)
        agent.epoch_num = 0
    t0 = time.time()
    for _ in range(epochs):
        agent.train_epoch()
    torch.cuda.synchronize()
    st = {"epochs": epochs, "restored_from": restore or None, "minibatch_size": minibatch, "wall_s": time.time() - t0, "game_reward": float(agent.game_rewards.get_mean()[0]),
          "game_length": float(agent.game_lengths.get_mean()[0]), "grasp_states_harvested_per_type": task.sim.HARVEST_COUNT.cpu().tolist(),
          "tvalue_gate": "open" if tvalue_state is None else "given", "contact_stats": task.sim.CONTACT_STATS.cpu().tolist()}
    if lift_statistics:
        st["deterministic_play"] = policy_lift_statistics(task, agent)
    path = ""
    if save_to:
        agent.save(save_to)
        path = save_to + ".pth"
    agent.ppo.close()
    return path, task, st


def prepare_tvalue_and_insert_policy(n, epochs, fit_iters=10000, seed=22, save_to=None, grasp_states=None, restore="", synthetic_fallback=False, fit=True):
    """stage 0 of the chain (untimed; the backward pass of scripts/bi_optimization.py:120-121 in small): BlockAssemblyInsertSim trains
    `epochs` epochs with its shipped schedule from synthetic grasp states (or, grasp_states given, from grasp terminal states a grasp
    policy harvested), its episode outcomes fill the T-value rings, GraspInsertTValue
    is fitted to them -> the transition value that gates the harvests of the chain, and the insert policy of its last stage.
    Deterministic run to run: torch's global generator is seeded like the launcher does (it feeds VecTask.reset()'s noise step), training
    is (fixed-order reductions, counter-based noise) and the fit reads the outcome rings in serial (step, env) order (SdxSim.ring_rows),
    not in the order the slots were claimed in.
    restore: an insert checkpoint to go on from (the forward leg of a later bi-optimisation round: the policy that learned on synthetic
    states is fine-tuned on the states a grasp policy harvested); synthetic_fallback: brick-type groups without a harvested state start from
    synthetic states (named in the statistics); fit=False: no transition-value fit (returns None for it).
    Returns (flat T-value weights or None, insert checkpoint path or "", statistics)."""
    from ..tasks.block_assembly_insert_sim import BlockAssemblyInsertSim
    from ..tvalue_trainer import TValue_Trainer, flat_from_state_dict
    set_seed(seed)        # TR:70: torch's global generator feeds VecTask.reset()'s noise step (VR:179-192)
    cfg = yaml.safe_load(open(os.path.join(ROOT, TASK_CFG["BlockAssemblyInsertSim"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, TRAIN_CFG["BlockAssemblyInsertSim"])))
    task = BlockAssemblyInsertSim(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, grasp_states=grasp_states,
                                  synthetic_fallback=synthetic_fallback)
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=seed)
    agent = A2CAgent("run", tr["params"])
    if restore:
        agent.restore(restore)
        agent.epoch_num = 0
    t0 = time.time()
    for _ in range(epochs):
        agent.train_epoch()
    torch.cuda.synchronize()
    sb = task.extras["success_buf"].float()
    real = torch.tensor([(e % 8) not in task.synthetic_groups for e in range(n)], device=sb.device)
    st = {"epochs": epochs, "wall_s": time.time() - t0, "game_reward": float(agent.game_rewards.get_mean()[0]), "grasp_states": task.grasp_states_source,
          "restored_from": restore or None, "outcomes_logged(success, failure)": task.sim.TV_COUNT.cpu().tolist(),
          "insert_success_buf_mean": float(sb.mean()),
          "insert_success_buf_mean_of_the_groups_with_given_states": float(sb[real].mean()) if bool(real.any()) else None}
    path = ""
    if save_to:
        agent.save(save_to)
        path = save_to + ".pth"
    tv = None
    if not fit:
        agent.ppo.close()
        task.sim.close()
        return None, path, st
    try:
        trn = TValue_Trainer.from_task(task, seed=seed)
        trn.init_TValue_function("BlockAssemblyInsertSim", fit_iters)
        # Only about 0.3 % of all brick orientations are ones InsertSim succeeds from.  The fit goes on (at most three more rounds) until it
        # rates at least 0.05 % of 20 000 random orientations above the chain's Orient gate: below that Orient's harvest can come out empty.
        g = torch.Generator().manual_seed(0)
        q = torch.randn(20000, 4, generator=g)
        q = (q / q.norm(dim=1, keepdim=True)).to(task.sim.device)
        rounds, cover = 0, 0.0
        while rounds < 4:
            trn.train_rollout()
            rounds += 1
            out = torch.cat([trn.predict(q[i:i + 1024]) for i in range(0, q.shape[0], 1024)])     # (sdxtv_predict takes at most one batch)
            cover = float((torch.sigmoid(out)[:, 1] > 0.5).float().mean())
            if cover >= 5e-4:
                break
        st["tvalue_fit"] = {"iterations": fit_iters * rounds, "loss": trn.losses[-1], "held_out_success_rate": trn.valid_t_value_success_rate,
                            "random_orientations_rated_above_0.5": cover}
        tv = flat_from_state_dict(trn.state_dict()).numpy()
        trn.close()
    except ValueError as ex:
        st["tvalue_fit"] = "skipped: %s" % ex
    agent.ppo.close()
    task.sim.close()
    return tv, path, st


def train_orient_policy(n, epochs, tvalue_state, gate=0.99, seed=22, save_to=None, minibatch=GRASP_TRAIN_MINIBATCH, initial_piles=None):
    """A BlockAssemblyOrient policy of THIS engine (round 6, VERDICT r5 item 6: the chain's Orient stage played a random initialisation):
    `epochs` epochs at n envs with the task's shipped schedule but 2 048-row minibatches (as train_grasp_policy; the shipped 4-row schedule
    does not learn on this engine), under the transition value `tvalue_state` binarised at `gate` (OR:1203: 0.99).
    Returns (checkpoint path or "", statistics: game reward, piles harvested per brick-type group, outcomes logged, mean T-value of the
    last step)."""
    from ..tasks.block_assembly_orient import BlockAssemblyOrient
    set_seed(seed)
    cfg = yaml.safe_load(open(os.path.join(ROOT, TASK_CFG["BlockAssemblyOrient"])))
    cfg["env"]["numEnvs"] = n
    tr = yaml.safe_load(open(os.path.join(ROOT, TRAIN_CFG["BlockAssemblyOrient"])))
    tr["params"]["config"]["minibatch_size"] = minibatch
    tr["params"]["config"]["central_value_config"]["minibatch_size"] = minibatch
    task = BlockAssemblyOrient(cfg, device_type="cuda", device_id=0, headless=True, seed=seed, tvalue_gate=gate, piles_per_type=64, initial_piles=initial_piles)
    task.sim.set_tvalue_weights(tvalue_state)
    env = RLgamesVecTaskPython(task, "cuda:0")
    tr["params"]["config"].update(num_actors=n, vec_env=env, env_info=env.get_env_info(), seed=seed)
    agent = A2CAgent("run", tr["params"])
    t0 = time.time()
    first = None
    for ep in range(epochs):
        agent.train_epoch()
        if ep == min(99, epochs - 1):
            first = float(agent.game_rewards.get_mean()[0])
    torch.cuda.synchronize()
    st = {"epochs": epochs, "minibatch_size": minibatch, "tvalue_gate": gate, "wall_s": time.time() - t0, "game_reward_after_100_epochs": first,
          "game_reward": float(agent.game_rewards.get_mean()[0]), "game_length": float(agent.game_lengths.get_mean()[0]),
          "piles_harvested_per_type(during training)": task.sim.PILE_HARVEST_COUNT.cpu().tolist(),
          "outcomes_logged(success, failure)": task.sim.TV_COUNT.cpu().tolist(), "tvalue_mean_last_step": float(task.sim.TVALUE.mean()),
          "tvalue_max_last_step": float(task.sim.TVALUE.max())}
    path = ""
    if save_to:
        agent.save(save_to)
        path = save_to + ".pth"
    agent.ppo.close()
    task.sim.close()
    return path, st


def tvalue_over_random_orientations(tvalue_state, count=200000, seed=0):
    """what a fitted GraspInsertTValue says about `count` uniformly random orientations (torch on the CPU; a statistic for reports):
    maximum and the shares above the gates the chain uses"""
    from ..tvalue_trainer import LAYERS
    w = torch.from_numpy(np.asarray(tvalue_state, np.float32))
    off, x = 0, None
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(count, 4, generator=g)
    x = q / q.norm(dim=1, keepdim=True)
    for i, (_, out, inn) in enumerate(LAYERS):
        W = w[off:off + out * inn].view(out, inn); off += out * inn
        b = w[off:off + out]; off += out
        x = x @ W.t() + b
        if i < len(LAYERS) - 1:
            x = torch.relu(x)
    t = torch.sigmoid(x)[:, 1]
    return {"max": float(t.max()), "share_above_0.5": float((t > 0.5).float().mean()), "share_above_0.8": float((t > 0.8).float().mean()),
            "share_above_0.9": float((t > 0.9).float().mean()), "share_above_0.99": float((t > 0.99).float().mean())}


def fill_missing_pile_groups(harvest, counts, min_piles, seed, max_missing=2, keys=None):
    """Brick-type groups Orient could not fill (the gate of a briefly fitted T-value can miss the orientations one brick type settles in)
    start GraspSim from settled piles instead - the states GraspSim generates for itself when it is given none (piles.generate_piles) - as
    InsertSim's groups without a harvested grasp state fall back to its synthetic ones.  harvest [8, slots, 132, 13], counts [8].
    Returns ([8, K, 132, 13], the groups that were filled in) or (None, []) when every group has min_piles or more than max_missing lack them."""
    from ..piles import generate_piles
    cnt = np.minimum(counts.cpu().numpy(), harvest.shape[1])
    lacking = [t for t in range(8) if cnt[t] < min_piles]
    if not lacking or len(lacking) > max_missing:
        return None, []
    k = int(cnt[cnt >= min_piles].min())
    piles = harvest[:, :k].clone()
    if keys is not None:                                   # serial (step, env) order of the appends, as pile_terminal_states() hands them on
        for t in range(8):
            if cnt[t] >= k:
                piles[t] = harvest[t, :int(cnt[t])].index_select(0, torch.argsort(keys[t, :int(cnt[t])], stable=True))[:k]
    settled = torch.as_tensor(generate_piles(k, device=str(piles.device), seed=seed)).to(piles.device)
    for t in lacking:
        piles[t] = settled[t]
    return piles, lacking


def block_assembly_chain(num_envs=512, tvalue_state=None, policies=None, controllers=None, min_piles=8, seed=22, stage_steps=None,
                         synthetic_fallback=False, orient_tvalue_gate=0.99, grasp_tvalue_gate=0.8, with_search=False, min_grasp_states=0,
                         max_grasp_steps=None, orient_fallback=None):
    """Orient -> GraspSim -> InsertSim played back to back on one GPU.  policies / controllers / stage_steps: dicts keyed "orient",
    "grasp", "insert".  Orient plays until every brick-type group has `min_piles` harvested pile states (OR:1483-1488 fills rings of
    10 000; at most 8 episodes here).  orient_tvalue_gate: the threshold Orient binarises the transition value at (0.99, OR:1203), or
    a descending ladder of thresholds (see stage 1 below); a T-value fitted to a few hundred epochs of outcomes never gets that
    confident, so the chain benchmark lowers it (and GraspSim's 0.8, GS:1406) and says so.
    min_grasp_states > 0: GraspSim plays (in horizon-sized chunks, at most max_grasp_steps env steps) until every brick-type group has that
    many harvested grasp states - a learned grasp policy under the reference's gate 0.8 harvests a state every few dozen episodes.
    orient_fallback: overrides synthetic_fallback for Orient's hand-off only (settled piles for the groups Orient harvested nothing for).
    Returns (statistics, hand-off tensors for inspection)."""
    orient_fallback = synthetic_fallback if orient_fallback is None else orient_fallback
    policies, controllers, stage_steps = policies or {}, controllers or {}, stage_steps or {}
    out, hand = {"num_envs": num_envs, "min_piles_per_type": min_piles}, {}
    t_begin = time.time()
    dug = None
    if with_search:
        # ---- stage 0: BlockAssemblySearch at <= 128 envs (evaluation.py:111): piles whose target brick the camera sees go to Orient (SE:1323-1353)
        search, st = main_rlgames("BlockAssemblySearch", min(num_envs, 128), policy_path=policies.get("search", ""), seed=seed,
                                  controller=controllers.get("search"), steps=stage_steps.get("search", 2 * (75 + 8)))
        st["piles_harvested_per_type"] = search.sim.PILE_HARVEST_COUNT.cpu().tolist()
        dug = search.pile_terminal_states()
        if dug is not None and dug.shape[1] < 8:      # too few states to start 128 envs per group from: Orient would replay the same handful
            st["handed_on"] = "only %d piles per group (< 8): Orient settles its own piles" % dug.shape[1]
            dug = None
        else:
            st["handed_on"] = "none (a brick-type group has no dug-out pile): Orient settles its own piles" if dug is None else "%d piles per group" % dug.shape[1]
        search.sim.close()
        out["search"] = st
    # ---- stage 1: BlockAssemblyOrient.  orient_tvalue_gate may be a descending ladder of thresholds: Orient is replayed at the next rung
    # when a rung leaves more brick-type groups without piles than the settled-pile fallback covers.  What a briefly fitted transition
    # value accepts depends on which few orientations its training run happened to succeed from (stage 0 is deterministic for one build
    # of the library, but any change of a summation order moves it), so a fixed lowered gate can come out empty; the last rung 0.0
    # opens the gate.  The rung used and the rungs tried are in the statistics; only the run that was handed on is timed.
    ladder = list(orient_tvalue_gate) if isinstance(orient_tvalue_gate, (tuple, list)) else [orient_tvalue_gate]
    tried, probe_wall = [], 0.0
    for gate in ladder:
        orient, st = main_rlgames("BlockAssemblyOrient", num_envs, policy_path=policies.get("orient", ""), tvalue_state=tvalue_state,
                                  controller=controllers.get("orient"), seed=seed, steps=stage_steps.get("orient"),
                                  until=lambda t: int(t.sim.PILE_HARVEST_COUNT.min()) >= min_piles,
                                  max_steps=8 * 80 if stage_steps.get("orient") is None else stage_steps["orient"],
                                  task_kwargs={"tvalue_gate": gate, "piles_per_type": 64, "initial_piles": dug})
        st["piles_harvested_per_type"] = orient.sim.PILE_HARVEST_COUNT.cpu().tolist()
        st["tvalue_gate"] = gate
        piles = orient.pile_terminal_states()
        if orient_fallback:
            filled, lacking = fill_missing_pile_groups(orient.sim.PILE_HARVEST, orient.sim.PILE_HARVEST_COUNT, min_piles, seed, keys=orient.sim.PILE_HARVEST_KEYS)
            if lacking:
                piles, st["settled_stand_in_groups"] = filled, lacking
        orient.sim.close()
        tried.append({"tvalue_gate": gate, "piles_harvested_per_type": st["piles_harvested_per_type"]})
        if piles is not None:
            break
        probe_wall += st["wall_s"]
    if len(ladder) > 1:
        st["tvalue_gates_tried"], st["wall_s_of_the_rungs_not_handed_on"] = tried, probe_wall
    out["orient"] = st
    if piles is None:
        raise RuntimeError("BlockAssemblyOrient harvested no pile state for at least one brick-type group: %s" % st["piles_harvested_per_type"])
    hand["piles"] = piles
    # ---- stage 2: BlockAssemblyGraspSim from Orient's piles (GS:412-413).  grasp_tvalue_gate may be a ladder like Orient's: the next rung
    # is played when a rung harvests grasp states for fewer than three brick-type groups.
    gladder = list(grasp_tvalue_gate) if isinstance(grasp_tvalue_gate, (tuple, list)) else [grasp_tvalue_gate]
    gtried, gprobe = [], 0.0
    for gi, gate in enumerate(gladder):
        grasp, st = main_rlgames("BlockAssemblyGraspSim", num_envs, policy_path=policies.get("grasp", ""), tvalue_state=tvalue_state,
                                 controller=controllers.get("grasp"), seed=seed, steps=stage_steps.get("grasp"),
                                 until=(lambda t: int(t.sim.HARVEST_COUNT.min()) >= min_grasp_states) if min_grasp_states > 0 else None,
                                 max_steps=max_grasp_steps,
                                 task_kwargs={"initial_piles": piles, "harvest_tvalue_gate": gate})
        cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
        gtried.append({"tvalue_gate": gate, "grasp_states_harvested_per_type": cnt.tolist()})
        # (next rung: fewer than three groups harvested - or, when no stand-in states are allowed, any group without a state)
        if (int((cnt > 0).sum()) >= 3 and (synthetic_fallback or cnt.min() > 0)) or gi + 1 == len(gladder):
            break
        gprobe += st["wall_s"]
        grasp.sim.close()
    st["tvalue_gate"] = gate
    if len(gladder) > 1:
        st["tvalue_gates_tried"], st["wall_s_of_the_rungs_not_handed_on"] = gtried, gprobe
    st["grasp_states_harvested_per_type"] = cnt.tolist()
    st["initial_piles"] = "BlockAssemblyOrient.pile_terminal_states(): %d per brick-type group" % piles.shape[1]
    if cnt.min() > 0 or (synthetic_fallback and cnt.max() > 0):
        grasp_states = grasp.grasp_terminal_states()          # (groups without a harvested state: empty tensors -> InsertSim's stand-ins)
        hand["grasp_obj"], hand["grasp_hand"] = grasp_states
    elif synthetic_fallback:
        grasp_states = None
    else:
        grasp.sim.close()
        out["grasp"] = st
        raise RuntimeError("BlockAssemblyGraspSim harvested no grasp terminal state for at least one brick-type group: %s" % cnt.tolist())
    grasp.sim.close()
    out["grasp"] = st
    # ---- stage 3: BlockAssemblyInsertSim from the harvested grasp states (IS:372-375)
    insert, st = main_rlgames("BlockAssemblyInsertSim", num_envs, policy_path=policies.get("insert", ""), tvalue_state=tvalue_state,
                              controller=controllers.get("insert"), seed=seed, steps=stage_steps.get("insert"),
                              task_kwargs={"grasp_states": grasp_states, "synthetic_fallback": synthetic_fallback})
    st["grasp_states_source"] = insert.grasp_states_source
    hand["insert_task"] = insert          # the caller inspects it and closes insert.sim
    out["insert"] = st
    out["chain_wall_s"] = time.time() - t_begin
    stages = [k for k in ("search", "orient", "grasp", "insert") if k in out]
    steps = sum(out[k]["env_steps"] for k in stages)
    play = sum(out[k]["wall_s"] for k in stages)
    out["chain_env_steps"] = steps
    out["chain_env_steps_per_s"] = steps / play                      # the three rollouts back to back (task construction excluded)
    out["chain_env_steps_per_s_incl_setup"] = steps / out["chain_wall_s"]
    return out, hand


CHAIN_LEARNED_ORIENT_GATES = (0.99, 0.9, 0.8, 0.5, 0.3, 0.0)     # starts at the reference's threshold (OR:1203)


def block_assembly_chain_learned(num_envs=1024, grasp_epochs=1500, insert_epochs=1500, seed=22, workdir=None, min_grasp_states=1, max_grasp_steps=16000,
                                 insert_refit_epochs=4000):
    """BASELINE.json configs[2] on LEARNED policies (round 5) - no scripted stage, no synthetic grasp states:
      stage 0  BlockAssemblyInsertSim trains `insert_epochs` epochs with its shipped schedule from synthetic grasp states (the backward leg of
               bi_optimization.py:120-121 in small); GraspInsertTValue is fitted to its episode outcomes (thousands of successes since the
               studs engage) -> the transition value of the gates and the insert policy of the last stage;
      stage g  a BlockAssemblyGraspSim policy of this engine is trained `grasp_epochs` epochs (minibatch 2 048) with that transition value
               gating its harvest at the reference's 0.8 (GS:1406);
      stage r  (insert_refit_epochs > 0; the forward leg of the next bi-optimisation round, bi_optimization.py:115-118) the grasp policy is
               played from settled piles under the same gate, and the insert policy of stage 0 is fine-tuned on the states it harvested
               (brick-type groups without one keep synthetic states, named): a policy that has only seen synthetic hand poses inserts
               from 0.3 % of the learned grasp states, the fine-tuned one from several per cent;
      chain    Orient (random-initialised policy - its arm is scripted by the task - under a ladder of gates that starts at the reference's 0.99;
               the rung used is reported; a group it harvests nothing for starts GraspSim from settled piles, named in the statistics) ->
               GraspSim (the learned policy, gate 0.8, played until every brick-type group has `min_grasp_states` harvested states) ->
               InsertSim (the learned insert policy, started from those states only: a group without one raises, as IS:1449 fails).
    Returns (statistics, hand-off tensors; the caller closes hand["insert_task"].sim)."""
    import tempfile
    workdir = workdir or tempfile.mkdtemp(prefix="sdx_chain_learned_")
    tv, ipath, ist = prepare_tvalue_and_insert_policy(num_envs, insert_epochs, seed=seed, save_to=os.path.join(workdir, "insert"))
    if tv is None:
        raise RuntimeError("stage 0 logged too few insert outcomes of a class for a transition value: %s" % ist)
    gpath, gtask, gst = train_grasp_policy(num_envs, grasp_epochs, seed=seed, save_to=os.path.join(workdir, "grasp"), tvalue_state=tv)
    gtask.sim.close()
    rst = None
    if insert_refit_epochs > 0:
        g0, st0 = main_rlgames("BlockAssemblyGraspSim", num_envs, policy_path=gpath, tvalue_state=tv, steps=160, seed=seed + 1,
                               until=lambda t: int(t.sim.HARVEST_COUNT.min()) >= 64, max_steps=max_grasp_steps, task_kwargs={"harvest_tvalue_gate": 0.8})
        cnt0 = g0.sim.HARVEST_COUNT.cpu().tolist()
        if max(cnt0) > 0:
            s0 = g0.grasp_terminal_states()
            g0.sim.close()
            _, ipath, rst = prepare_tvalue_and_insert_policy(num_envs, insert_refit_epochs, seed=seed, save_to=os.path.join(workdir, "insert_refit"),
                                                             grasp_states=s0, restore=ipath, synthetic_fallback=True, fit=False)
            rst["grasp_states_harvested_per_type(settled piles, gate 0.8, %d steps per env)" % st0["steps_per_env"]] = cnt0
        else:
            g0.sim.close()
            rst = {"skipped": "the grasp policy harvested no state under gate 0.8 in %d steps per env" % st0["steps_per_env"]}
    res, hand = block_assembly_chain(num_envs, tv, policies={"grasp": gpath, "insert": ipath}, synthetic_fallback=False, orient_fallback=True,
                                     orient_tvalue_gate=CHAIN_LEARNED_ORIENT_GATES, grasp_tvalue_gate=0.8, stage_steps={"grasp": 160},
                                     min_grasp_states=min_grasp_states, max_grasp_steps=max_grasp_steps, seed=seed)
    out = {"stage0_insert_policy_and_tvalue(untimed)": ist, "grasp_policy(untimed)": gst, "insert_policy_refit(untimed)": rst, "chain": res,
           "stand_ins": ["Orient plays its random initialisation under T-value gate %s (reference: a trained Orient policy under 0.99)" % res["orient"]["tvalue_gate"]]
           + (["settled piles for Orient's brick-type groups %s" % res["orient"]["settled_stand_in_groups"]] if res["orient"].get("settled_stand_in_groups") else [])}
    return out, hand


def block_assembly_chain_closed(num_envs=1024, insert_epochs=1500, grasp_epochs=1500, insert_refit_epochs=4000, orient_epochs=600, seed=22, workdir=None,
                                min_grasp_states=100, max_grasp_steps=16000, orient_gates=(0.99,), refit_harvest_per_type=100, grasp_gates=(0.8,)):
    """The chain with EVERY stage on a learned policy and the transition value refitted to the policy that actually ends the chain
    (round 6; scripts/evaluation.py:111-119 on the output of one forward + backward pass of scripts/bi_optimization.py:110-124):
      stage 0  BlockAssemblyInsertSim trains from synthetic grasp states; GraspInsertTValue is fitted to its outcomes (as the learned chain);
      stage g  a GraspSim policy is trained under that value's gate 0.8 (GS:1406);
      stage r  the grasp policy is played from settled piles until every brick-type group has `refit_harvest_per_type` harvested states
               (round 5 took 64 and the chain itself handed on 32 states in all: VERDICT r5 item 5c), the insert policy is fine-tuned on them AND
               the transition value is REFITTED to the fine-tuned policy's outcomes (the value the backward legs of the loop hand to GraspSim
               and Orient, bi_optimization.py:121-124);
      stage o  a BlockAssemblyOrient policy is TRAINED under the refitted value (round 5: random initialisation);
      chain    Orient (trained; gate ladder `orient_gates`, default the reference's 0.99 alone) -> GraspSim (trained, gate 0.8, until every group
               has `min_grasp_states` states) -> InsertSim (fine-tuned).
    Returns (statistics, hand-off tensors; the caller closes hand["insert_task"].sim)."""
    import tempfile
    workdir = workdir or tempfile.mkdtemp(prefix="sdx_chain_closed_")
    tv0, ipath, ist = prepare_tvalue_and_insert_policy(num_envs, insert_epochs, seed=seed, save_to=os.path.join(workdir, "insert"))
    if tv0 is None:
        raise RuntimeError("stage 0 logged too few insert outcomes of a class for a transition value: %s" % ist)
    gpath, gtask, gst = train_grasp_policy(num_envs, grasp_epochs, seed=seed, save_to=os.path.join(workdir, "grasp"), tvalue_state=tv0)
    gtask.sim.close()
    # the harvest gate: GS:1406's 0.8 first; `grasp_gates` may continue with lower rungs for the seeds whose stage-0 value rates every
    # grasp of some brick type below 0.8 (two of three seeds in profiles/r6_chain_closed_seeds_22_23_24_first_attempt.json) - a rung
    # below 0.8 is reported as a stand-in
    tried0 = []
    for gi, gate0 in enumerate(grasp_gates):
        g0, st0 = main_rlgames("BlockAssemblyGraspSim", num_envs, policy_path=gpath, tvalue_state=tv0, steps=160, seed=seed + 1,
                               until=lambda t: int(t.sim.HARVEST_COUNT.min()) >= refit_harvest_per_type, max_steps=max_grasp_steps,
                               task_kwargs={"harvest_tvalue_gate": gate0})
        cnt0 = g0.sim.HARVEST_COUNT.cpu().tolist()
        tried0.append({"tvalue_gate": gate0, "grasp_states_harvested_per_type": cnt0, "steps_per_env": st0["steps_per_env"]})
        if min(cnt0) > 0 or gi + 1 == len(grasp_gates):
            break
        g0.sim.close()
    if min(cnt0) == 0:
        g0.sim.close()
        raise RuntimeError("the grasp policy harvested no state for a brick-type group under the gates %s in %d steps per env: %s" % (list(grasp_gates), st0["steps_per_env"], cnt0))
    s0 = g0.grasp_terminal_states()
    g0.sim.close()
    tv1, ipath1, rst = prepare_tvalue_and_insert_policy(num_envs, insert_refit_epochs, seed=seed, save_to=os.path.join(workdir, "insert_refit"),
                                                        grasp_states=s0, restore=ipath, synthetic_fallback=False, fit=True)
    rst["grasp_states_harvested_per_type(settled piles, gate %s, %d steps per env)" % (gate0, st0["steps_per_env"])] = cnt0
    rst["refit_harvest_gates_tried"] = tried0
    tv = tv1 if tv1 is not None else tv0
    tvs = {"stage0": tvalue_over_random_orientations(tv0), "refitted": tvalue_over_random_orientations(tv1) if tv1 is not None else None,
           "used_by_the_chain": "refitted to the fine-tuned insert policy" if tv1 is not None else "stage 0 (the refit was skipped: %s)" % rst.get("tvalue_fit")}
    opath, ost = train_orient_policy(num_envs, orient_epochs, tv, gate=orient_gates[0], seed=seed, save_to=os.path.join(workdir, "orient"))
    res, hand = block_assembly_chain(num_envs, tv, policies={"orient": opath, "grasp": gpath, "insert": ipath1}, synthetic_fallback=False, orient_fallback=True,
                                     orient_tvalue_gate=tuple(orient_gates), grasp_tvalue_gate=tuple(grasp_gates), stage_steps={"grasp": 160},
                                     min_grasp_states=min_grasp_states, max_grasp_steps=max_grasp_steps, seed=seed)
    stand_ins = []
    if gate0 != 0.8:
        stand_ins.append("refit harvest of grasp states under gate %s instead of 0.8 (GS:1406)" % gate0)
    if res["grasp"]["tvalue_gate"] != 0.8:
        stand_ins.append("the chain's GraspSim stage under gate %s instead of 0.8 (GS:1406)" % res["grasp"]["tvalue_gate"])
    if res["orient"]["tvalue_gate"] != 0.99:
        stand_ins.append("Orient's gate %s instead of 0.99 (the ladder's first rung that harvested)" % res["orient"]["tvalue_gate"])
    if res["orient"].get("settled_stand_in_groups"):
        stand_ins.append("settled piles for Orient's brick-type groups %s" % res["orient"]["settled_stand_in_groups"])
    out = {"stage0_insert_policy_and_tvalue(untimed)": ist, "grasp_policy(untimed)": gst, "insert_policy_refit_and_tvalue_refit(untimed)": rst,
           "orient_policy(untimed)": ost, "tvalue_over_200000_random_orientations": tvs, "chain": res, "stand_ins": stand_ins}
    return out, hand


# ---------------------------------------------------------------------------------------------------------------------------------
# the checkpoint-driven form (round 2): every stage through the launcher's own argument parsing, optional BlockAssemblySearch stage first
def _launcher():
    from ..config import get_args
    from ..train_rlgames import build
    return get_args, build


def play_checkpoint(task, num_envs, play=True, use_t_value=False, policy_path="", games=0, task_kwargs=None, minibatch_size=0):
    """evaluation.py:36-103 for one stage through the reference's command line (--task --num_envs --checkpoint --play; seqdex_amd.train_rlgames.build):
    the sub-policy is restored from its checkpoint and played for `games` finished episodes.  Returns (mean episode reward, mean episode length, task object)."""
    argv = ["--task=%s" % task, "--num_envs=%d" % num_envs, "--headless", "--play"]
    if policy_path:
        argv.append("--checkpoint=%s" % policy_path)
    get_args, build = _launcher()
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
        r, l, search = play_checkpoint("BlockAssemblySearch", min(num_envs, 128), use_t_value=True, policy_path=search_path, games=games)
        dug = search.pile_terminal_states()
        out["BlockAssemblySearch"] = dict(reward=r, length=l, search_success_rate=float(search.extras["success_buf"].float().mean()),
                                          piles_handed_on=0 if dug is None else int(dug.shape[1]))
        search.sim.close()
    r, l, orient = play_checkpoint("BlockAssemblyOrient", num_envs, use_t_value=True, policy_path=orient_path, games=games,
                                task_kwargs={"initial_piles": dug})
    piles = orient.pile_terminal_states()
    out["BlockAssemblyOrient"] = dict(reward=r, length=l, piles_handed_on=0 if piles is None else int(piles.shape[1]))
    orient.sim.close()
    r, l, grasp = play_checkpoint("BlockAssemblyGraspSim", num_envs, use_t_value=True, policy_path=grasp_path, games=games,
                               task_kwargs={"initial_piles": piles})
    cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
    states = grasp.grasp_terminal_states() if cnt.min() > 0 else None
    out["BlockAssemblyGraspSim"] = dict(reward=r, length=l, grasp_states_handed_on=int(cnt.sum()))
    grasp.sim.close()
    r, l, insert = play_checkpoint("BlockAssemblyInsertSim", num_envs, use_t_value=True, policy_path=insert_path, games=games,
                                task_kwargs={"grasp_states": states}, minibatch_size=insert_minibatch)
    out["BlockAssemblyInsertSim"] = dict(reward=r, length=l, insert_success_rate=float(insert.extras["success_buf"].float().mean()),
                                         grasp_states=insert.grasp_states_source)
    insert.sim.close()
    for k, v in out.items():
        print(k, v)
    return out


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="scripts/evaluation.py of the reference: play the BlockAssembly sub-policies back to back")
    p.add_argument("--tasks", type=str, default="BlockAssembly")
    p.add_argument("--mode", choices=["checkpoint", "chain", "chain_learned"], default="checkpoint",
                   help="checkpoint: every stage restored from its rl_games .pth through the launcher and played for --games episodes "
                        "(evaluation.py:111-119; a stage without a checkpoint plays its random initialisation); chain: the device-tensor "
                        "hand-off chain of block_assembly_chain with the harvest gates below; chain_learned: block_assembly_chain_learned "
                        "(trains the insert policy, the transition value and a grasp policy and fine-tunes the insert policy first: under three minutes at 1 024 envs)")
    p.add_argument("--num_envs", type=int, default=512)
    for st_ in ("search", "orient", "grasp", "insert"):
        p.add_argument("--%s" % st_, "--%s_policy" % st_, dest=st_, type=str, default="", help="rl_games checkpoint (.pth) of the %s stage" % st_)
    p.add_argument("--with_search", action="store_true", help="put BlockAssemblySearch (128 envs) in front (always on in checkpoint mode when --search is given)")
    p.add_argument("--games", type=int, default=0, help="checkpoint mode: finished episodes per stage (0 = num_envs)")
    p.add_argument("--insert_minibatch", type=int, default=0)
    p.add_argument("--tvalue", type=str, default="", help="chain mode: GraspInsertTValue state_dict (.pt) for the harvest gates")
    p.add_argument("--synthetic_fallback", action="store_true", help="chain mode: brick-type groups a stage harvested nothing for start the next "
                   "stage from settled piles / synthetic grasp states (named in the statistics) instead of failing as the reference does")
    p.add_argument("--orient_tvalue_gate", type=float, nargs="+", default=[0.99], help="OR:1203; several values = a descending ladder (block_assembly_chain)")
    p.add_argument("--grasp_tvalue_gate", type=float, nargs="+", default=[0.8], help="GS:1406; several values = a ladder")
    a = p.parse_args()
    if a.tasks != "BlockAssembly":
        raise Exception("Unrecognized task!")                        # evaluation.py:121-129 (ToolPositioning: not built)
    if a.mode == "chain_learned":
        import json
        res, h = block_assembly_chain_learned(a.num_envs)
        h["insert_task"].sim.close()
        print(json.dumps(res))
    elif a.mode == "checkpoint":
        block_assembly(a.orient, a.grasp, a.insert, num_envs=a.num_envs, games=a.games, insert_minibatch=a.insert_minibatch,
                       search_path=a.search if (a.search or a.with_search) else None)
    else:
        tv = None
        if a.tvalue:
            from ..tvalue_trainer import flat_from_state_dict
            tv = flat_from_state_dict(torch.load(a.tvalue, map_location="cpu")).numpy()
        res, h = block_assembly_chain(a.num_envs, tv, {"search": a.search, "orient": a.orient, "grasp": a.grasp, "insert": a.insert},
                                      synthetic_fallback=a.synthetic_fallback, orient_tvalue_gate=a.orient_tvalue_gate[0] if len(a.orient_tvalue_gate) == 1 else a.orient_tvalue_gate,
                                      grasp_tvalue_gate=a.grasp_tvalue_gate[0] if len(a.grasp_tvalue_gate) == 1 else a.grasp_tvalue_gate, with_search=a.with_search)
        h["insert_task"].sim.close()
        import json
        print(json.dumps(res))
