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
                 minibatch_size=0, mixed_precision=False, report=None, leg="", seed=None):
    """one training run of `task` (bi_optimization.py:36-104).  Returns (checkpoint path, task object or None).  use_t_value marks the
    backward-pass runs whose purpose is the task's success / failure datasets (they are always logged on the device here).
    mixed_precision: rl_games' `mixed_precision` key for this run (BASELINE.json configs[4] "bf16 policy": bf16 MFMA operands with fp32
    master weights and accumulation wherever the schedule's update is GEMM-shaped, i.e. minibatch_size > 8).  report: a list that
    receives one dict per run (what ran, on which update path, how fast)."""
    import time
    argv = ["--task=%s" % task, "--num_envs=%d" % num_envs, "--headless"]
    if max_iterations:
        argv.append("--max_iterations=%d" % max_iterations)
    if policy_path:
        argv.append("--checkpoint=%s" % policy_path)
    if seed is not None:
        argv.append("--seed=%d" % seed)       # (the launcher's own flag; default 22, TR:62-65)
    args = get_args(argv)
    args.use_t_value = use_t_value
    task_obj, env, agent, logdir, rank = build(args, task_kwargs, minibatch_size, {"mixed_precision": bool(mixed_precision)})
    if tvalue_state is not None:
        task_obj.sim.set_tvalue_weights(flat_from_state_dict(tvalue_state).numpy())
    if policy_path:
        agent.epoch_num = 0        # every run of the outer loop trains max_iterations MORE epochs (rl_games would resume the counter)
    frame0, t_opt0 = agent.frame, int(agent.ppo.ctrl().ac_t)        # (a restored checkpoint brings its frame and Adam step counters along)
    torch.cuda.synchronize()
    t0 = time.time()
    agent.train()
    torch.cuda.synchronize()
    dt = time.time() - t0
    if report is not None:
        c = agent.ppo.ctrl()
        p_ac = agent.ppo.t["AC_PARAMS"]
        frames = agent.frame - frame0
        report.append({"leg": leg, "task": task, "num_envs": num_envs, "epochs": agent.epoch_num, "env_steps": frames, "wall_s": dt,
                       "env_steps_per_s": frames / dt, "minibatch_size": agent.minibatch_size, "update_impl": agent.ppo.update_impl(),
                       "mixed_precision": bool(agent.ppo.cfg.mixed_precision),
                       "bf16_mfma_in_update": bool(agent.ppo.cfg.mixed_precision) and agent.ppo.update_impl() == "gemm",
                       "optimiser_steps": int(c.ac_t) - t_opt0, "params_finite": bool(torch.isfinite(p_ac).all()),
                       "restored_from": policy_path or None, "tvalue_given": tvalue_state is not None,
                       "tvalue_outcomes_logged(success, failure)": task_obj.sim.TV_COUNT.cpu().tolist(),
                       "game_reward": float(agent.game_rewards.get_mean()[0])})
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


def _handoff(report, name, tensor_or_none, fallback):
    """record a stage-to-stage hand-off: its shape and finiteness, or the fallback the next stage takes when the stage harvested nothing"""
    if report is None:
        return
    if tensor_or_none is None:
        report.append({"handoff": name, "source": fallback, "empty": True})
        return
    ts = tensor_or_none if isinstance(tensor_or_none, (list, tuple)) else [tensor_or_none]
    ts = [t for group in ts for t in (group if isinstance(group, (list, tuple)) else [group])]
    report.append({"handoff": name, "source": "harvested", "empty": any(t.numel() == 0 for t in ts),
                   "shapes": [list(t.shape) for t in ts][:4], "finite": all(bool(torch.isfinite(t.float()).all()) for t in ts)})


def block_assembly(rounds=10, num_envs=512, epochs=0, tvalue_rollout=10000, insert_minibatch=0, mixed_precision=False, report=None,
                   stage_epochs=None, search_envs=128, orient_backward_envs=128, grasp_harvest_stand_in=False, gates=None, gates_after_fit=None,
                   grasp_minibatch=0, seed=None):
    """insert_minibatch: override of the insert schedule's minibatch_size 4096 for runs with fewer than 512 envs.
    grasp_minibatch: override of the grasp schedule's minibatch_size 4 (both GraspSim legs): on this engine the shipped 4-row minibatches
    with the adaptive LR do not learn to lift within thousands of epochs, 2 048-row minibatches do within hundreds.
    stage_epochs: {"search" | "orient" | "grasp" | "insert": max_iterations} overriding `epochs` per task (an episode is 75 / 75 / 150 / 125
    env steps = 10 / 10 / 19 / 16 epochs of horizon 8: shorter runs finish no episode, harvest nothing and log no T-value outcome).
    (keys "<task>_backward" override the backward leg of a task.)
    gates: {"orient": 0.99, "grasp": 0.8} harvest thresholds on the transition value (OR:1203, GS:1406) while no transition value has been
    fitted yet; gates_after_fit: the same once one has (default: the reference's thresholds).
    grasp_harvest_stand_in: when the freshly trained grasp policy has not carried a single brick to the insertion side yet (the reference's
    grasp checkpoint is from epoch 19 000, README.md:90), play two episodes of evaluation.scripted_grasp_controller on the same task so that
    InsertSim starts from REAL terminal states of this engine; brick-type groups still empty get InsertSim's synthetic stand-ins.  Both are
    named in the report.  Without it such a round hands `None` on and InsertSim synthesises all of its states (round-3 behaviour).  The same
    stand-in plays two episodes at the end of the BACKWARD grasp leg when that leg logged too few successful outcomes for a fit (the trainer
    holds out 100 success rows, transition_value_trainer.py:170-171): outcomes of this engine's physics under a scripted hand."""
    tv = None
    paths = {}
    se = lambda k: int((stage_epochs or {}).get(k, (stage_epochs or {}).get(k.split("_")[0], epochs)))
    mp = dict(mixed_precision=mixed_precision, report=report, **({} if seed is None else {"seed": seed}))

    def gate_kw():
        g = (gates_after_fit if tv is not None else gates) or {}
        return ({"tvalue_gate": g["orient"]} if "orient" in g else {}), ({"harvest_tvalue_gate": g["grasp"]} if "grasp" in g else {})

    def scripted_episodes(task_obj, episodes=2):
        from .evaluation import scripted_grasp_controller
        from ..vec_task_rlgames import RLgamesVecTaskPython
        env = RLgamesVecTaskPython(task_obj, "cuda:0")
        env.reset()
        for step in range(episodes * 160):
            env.step(scripted_grasp_controller(task_obj, step))
        torch.cuda.synchronize()

    for i in range(rounds):
        orient_kw, grasp_kw = gate_kw()
        # ---- forward initialisation (bi_optimization.py:115-118)
        paths["search"], search = main_rlgames("BlockAssemblySearch", min(num_envs, search_envs), max_iterations=se("search"),
                                               policy_path=paths.get("search", ""), keep=True, leg="forward", **mp)
        dug = search.pile_terminal_states()                                               # hand-off SE:1323-1353 -> Orient's saved piles
        if dug is not None and dug.shape[1] < 8:
            dug = None                                                                    # too few to start hundreds of envs per group from
        _handoff(report, "Search -> Orient: dug-out piles [8, K, 132, 13]", dug, "Orient settles its own piles (piles.generate_piles)")
        search.sim.close()
        paths["orient"], orient = main_rlgames("BlockAssemblyOrient", num_envs, max_iterations=se("orient"), policy_path=paths.get("orient", ""),
                                               tvalue_state=tv, keep=True, task_kwargs=dict(orient_kw, initial_piles=dug), leg="forward", **mp)
        piles = orient.pile_terminal_states()                                             # hand-off OR:1483-1510 -> GS:412-413
        _handoff(report, "Orient -> GraspSim: pile terminal states [8, K, 132, 13]", piles, "GraspSim settles its own piles")
        orient.sim.close()
        paths["grasp"], grasp = main_rlgames("BlockAssemblyGraspSim", num_envs, max_iterations=se("grasp"), policy_path=paths.get("grasp", ""),
                                             tvalue_state=tv, keep=True, task_kwargs=dict(grasp_kw, initial_piles=piles), leg="forward",
                                             minibatch_size=grasp_minibatch, **mp)
        cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
        harvest_by = "the trained grasp policy"
        if cnt.min() == 0 and grasp_harvest_stand_in:
            scripted_episodes(grasp)
            cnt = grasp.sim.HARVEST_COUNT.cpu().numpy()
            harvest_by = "evaluation.scripted_grasp_controller, two episodes on the trained task (STAND-IN: the policy of %d epochs harvested nothing)" % se("grasp")
        # the harvest is handed on as it is: a brick-type group that harvested nothing makes BlockAssemblyInsertSim raise (as IS:1449
        # samples an empty list) unless grasp_harvest_stand_in asks for synthetic stand-ins for exactly those groups; only a harvest that
        # is empty altogether leaves the task to synthesise all its start states, and the report says so (a partial harvest
        # used to be discarded silently)
        some = cnt.max() > 0
        grasp_states = grasp.grasp_terminal_states() if some else None                    # hand-off GS:1447-1450 -> IS:372-375
        if not some:
            print("bi_optimization: GraspSim harvested no terminal state - BlockAssemblyInsertSim starts from synthetic states")
        _handoff(report, "GraspSim -> InsertSim: grasp terminal states (8 x [K, 1, 13], 8 x [K, 23, 2])",
                 None if grasp_states is None else [t for t in list(grasp_states[0]) + list(grasp_states[1]) if t.numel()],
                 "InsertSim synthesises its start states")
        if report is not None:
            report[-1].update(harvested_per_type=cnt.tolist(), harvested_by=harvest_by)
        grasp.sim.close()
        if some and cnt.min() == 0 and not grasp_harvest_stand_in:
            # said HERE, before the InsertSim legs are built: the raise inside the task (IS:1449) would come after three trained stages
            raise RuntimeError("bi_optimization: the grasp policy harvested terminal states for %d of 8 brick-type groups only (%s); "
                               "BlockAssemblyInsertSim cannot start envs of the empty groups.  Train the GraspSim legs with "
                               "--grasp_minibatch 2048 (the shipped minibatch of 4 does not learn to lift on this engine, DESIGN.md "
                               "section 17) or pass --grasp_harvest_stand_in to fill the empty groups with labelled stand-ins"
                               % (int((cnt > 0).sum()), cnt.tolist()))
        insert_kw = {"grasp_states": grasp_states, "synthetic_fallback": bool(grasp_harvest_stand_in)}
        paths["insert"], _ = main_rlgames("BlockAssemblyInsertSim", num_envs, max_iterations=se("insert"), policy_path=paths.get("insert", ""),
                                          task_kwargs=insert_kw, minibatch_size=insert_minibatch, leg="forward", **mp)
        # ---- backward fine-tuning (bi_optimization.py:120-124)
        _, insert = main_rlgames("BlockAssemblyInsertSim", num_envs, use_t_value=True, policy_path=paths["insert"], max_iterations=se("insert_backward"),
                                 task_kwargs=insert_kw, keep=True, minibatch_size=insert_minibatch, leg="backward", **mp)
        if report is not None:
            report[-1]["grasp_states_source"] = insert.grasp_states_source
        tv_before = tv
        tv = transition_value_trainer(insert, tvalue_rollout, tv, seed=i)
        _handoff(report, "T-value fitted on InsertSim's outcomes -> GraspSim", None if tv is tv_before or tv is None else list(tv.values()),
                 "no fit (too few outcomes of a class)")
        if report is not None:
            report[-1].update(outcomes_success_failure=insert.sim.TV_COUNT.cpu().tolist())
        insert.sim.close()
        orient_kw, grasp_kw = gate_kw()                                                   # a transition value may exist from here on
        paths["grasp"], grasp = main_rlgames("BlockAssemblyGraspSim", num_envs, use_t_value=True, policy_path=paths["grasp"],
                                             max_iterations=se("grasp_backward"), tvalue_state=tv, keep=True,
                                             task_kwargs=dict(grasp_kw, initial_piles=piles), leg="backward", minibatch_size=grasp_minibatch, **mp)
        outcomes_by = "the fine-tuned grasp policy"
        if grasp_harvest_stand_in and int(grasp.sim.TV_COUNT[0]) <= 100:
            scripted_episodes(grasp)
            outcomes_by = "the fine-tuned grasp policy + two episodes of evaluation.scripted_grasp_controller (STAND-IN: the policy alone logged too few successes for a fit)"
        tv_before = tv
        tv = transition_value_trainer(grasp, tvalue_rollout, tv, seed=100 + i)            # bi_optimization.py:122: fit on GraspSim's own outcomes
        _handoff(report, "T-value fitted on GraspSim's outcomes -> Orient", None if tv is tv_before or tv is None else list(tv.values()), "no fit")
        if report is not None:
            report[-1].update(outcomes_by=outcomes_by, outcomes_success_failure=grasp.sim.TV_COUNT.cpu().tolist())
        grasp.sim.close()
        orient_kw, grasp_kw = gate_kw()                                                   # (the insert leg's fit may have been skipped and this one not)
        paths["orient"], orient = main_rlgames("BlockAssemblyOrient", min(num_envs, orient_backward_envs), use_t_value=True, policy_path=paths["orient"],
                                               max_iterations=se("orient_backward"), tvalue_state=tv, keep=True,
                                               task_kwargs=dict(orient_kw, initial_piles=dug), leg="backward", **mp)   # :123
        tv_before = tv
        tv = transition_value_trainer(orient, tvalue_rollout, tv, seed=200 + i)           # bi_optimization.py:124
        _handoff(report, "T-value fitted on Orient's outcomes -> next round", None if tv is tv_before or tv is None else list(tv.values()), "no fit")
        if report is not None:
            report[-1].update(outcomes_success_failure=orient.sim.TV_COUNT.cpu().tolist())
        orient.sim.close()
        print("bi-optimisation round %d done: %s" % (i, paths))
    return paths, tv


# (InsertSim's update is GEMM-shaped - 50 ms per epoch at 4096 envs - so three episodes cost 2.5 s; an insert policy of that age inserts
# nothing yet, and 300 epochs from scripted-grasp states gave 10 insertions in 1.2 M episodes: the first refit is skipped, with its reason)
CONFIG5_EPOCHS = {"search": 20, "orient": 10, "grasp": 20, "insert": 48, "insert_backward": 32}
# round 5: the grasp stage long enough, and on minibatches large enough, for the policy to LEARN to lift: its own
# terminal states go to InsertSim and its own outcomes to the transition-value fit - no scripted stand-in
# orient_backward: the backward Orient leg plays 128 envs (bi_optimization.py:124); at the forward leg's 10 epochs it finishes 128 episodes in all,
# and the transition-value trainer holds out 100 success rows - whether the last refit of a round happened then hung on a 78 % success share
# (round 6: a rounding change in the persistent update moved it to 23 % and the refit was skipped).  60 epochs = ~770 episodes, +2 s
CONFIG5_LEARNED_EPOCHS = dict(CONFIG5_EPOCHS, grasp=400, grasp_backward=100, orient_backward=60)
CONFIG5_GRASP_MINIBATCH = 2048


def one_round_at_size(num_envs=4096, mixed_precision=True, stage_epochs=None, tvalue_rollout=300, workdir=None, grasp_minibatch=0):
    """BASELINE.json configs[4] on ONE GPU: one round of block_assembly() at `num_envs` envs (Search and the backward Orient leg at the
    reference's 128, bi_optimization.py:111,124) with `mixed_precision` in every stage's PPO YAML and each task's shipped minibatch size;
    epochs per stage long enough for episodes to finish (CONFIG5_EPOCHS).  Returns (report dict with per-stage rates and every hand-off,
    checkpoint paths, fitted transition value or None).  tools/bench_config5.py is its command line, tests/test_gpu_bi_optimization_fullsize.py
    its test; what is a stand-in is listed in the report (`stand_ins`).
    grasp_minibatch > 0 (CONFIG5_GRASP_MINIBATCH): the GraspSim legs train on minibatches of that size for CONFIG5_LEARNED_EPOCHS; the
    scripted stand-in stays armed but is reported only when it had to play."""
    import tempfile
    import time
    stage_epochs = dict(CONFIG5_LEARNED_EPOCHS if grasp_minibatch else CONFIG5_EPOCHS, **(stage_epochs or {}))
    cwd = os.getcwd()
    tmp = workdir or tempfile.mkdtemp(prefix="sdx_config5_")     # logs/<task>/nn/<task>.pth checkpoints are hand-offs inside the run
    os.chdir(tmp)
    report = []
    torch.cuda.synchronize()
    t0 = time.time()
    try:
        paths, tv = block_assembly(rounds=1, num_envs=num_envs, tvalue_rollout=tvalue_rollout, mixed_precision=mixed_precision, report=report,
                                   stage_epochs=stage_epochs, grasp_harvest_stand_in=True, gates={"orient": 0.0, "grasp": 0.0},
                                   gates_after_fit={"orient": 0.5, "grasp": 0.28}, grasp_minibatch=grasp_minibatch)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    wall = time.time() - t0
    paths = {k: os.path.join(tmp, v) for k, v in paths.items()}          # (logs/<task>/nn/<task>.pth relative to the work directory)
    runs = [r for r in report if "task" in r]
    hand = [r for r in report if "handoff" in r]
    steps = sum(r["env_steps"] for r in runs)
    train_s = sum(r["wall_s"] for r in runs)
    out = {"config": "BASELINE.json configs[4] on one GPU: bi-optimisation loop Search -> Orient -> GraspSim -> InsertSim + three backward legs, "
                     "num_envs=%d (Search 128, backward Orient 128), %s, %s" % (num_envs, "mixed_precision: True (bf16 MFMA on "
                     "GEMM-shaped updates)" if mixed_precision else "fp32", "shipped minibatch sizes" if not grasp_minibatch else
                     "shipped minibatch sizes except GraspSim's: %d instead of 4 (DESIGN.md section 17)" % grasp_minibatch),
           "metric": "env-steps/s over the seven training runs of one round (rollout + PPO update; task construction and T-value fits excluded)",
           "value": steps / train_s, "unit": "env-steps/s", "env_steps": steps, "training_wall_s": train_s, "loop_wall_s_incl_setup_and_fits": wall,
           "n_gpus": 1, "configs4_on_8_gpus": "not run: no multi-GPU box has ever been available to this build (gpurun: 1 GPU)",
           "stage_epochs": stage_epochs, "tvalue_fit_iterations": tvalue_rollout,
           "stand_ins": ["harvest gates 0.0 in the forward pass of this first round: no transition value has been fitted before it; 0.5 / 0.28 "
                         "(not the reference's 0.99 / 0.8) in the backward legs",
                         "grasp terminal states, and the successes of the backward grasp leg's fit, from evaluation.scripted_grasp_controller on the "
                         "trained task when the %d-epoch policy produced (almost) none: %s" % (stage_epochs["grasp"], "; ".join(
                             "%s = %s" % (k, h[k]) for h in hand for k in ("harvested_by", "outcomes_by") if k in h))],
           "runs": runs, "handoffs": hand, "checkpoints": paths, "tvalue_fitted": tv is not None}
    return out, paths, tv




if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--tasks", type=str, default="BlockAssembly")
    p.add_argument("--rounds", type=int, default=10)
    p.add_argument("--num_envs", type=int, default=512)
    p.add_argument("--epochs", type=int, default=0, help="max_iterations of every training run (0 = the YAML's max_epochs)")
    p.add_argument("--tvalue_rollout", type=int, default=10000)
    p.add_argument("--mixed_precision", action="store_true", help="rl_games' mixed_precision key for every stage (bf16 MFMA on GEMM-shaped updates)")
    p.add_argument("--grasp_minibatch", type=int, default=0, help="minibatch_size of the GraspSim legs instead of the shipped 4 (2048 learns to lift "
                   "on this engine, the shipped 4 does not: DESIGN.md section 17)")
    p.add_argument("--grasp_harvest_stand_in", action="store_true", help="brick-type groups for which the trained grasp policy harvested no "
                   "terminal state: two scripted episodes first, then synthetic start states for InsertSim (labelled STAND-IN in the report); "
                   "without it InsertSim raises for an empty group, as the reference samples an empty list (IS:1449)")
    a = p.parse_args()
    if a.tasks != "BlockAssembly":
        raise Exception("Unrecognized task!")                                           # bi_optimization.py:141-143 (ToolPositioning: not built)
    block_assembly(a.rounds, a.num_envs, a.epochs, a.tvalue_rollout, mixed_precision=a.mixed_precision, grasp_minibatch=a.grasp_minibatch,
                   grasp_harvest_stand_in=a.grasp_harvest_stand_in)
