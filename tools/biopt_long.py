#!/usr/bin/env python3
"""BASELINE.json configs[4] on one MI355X with stage lengths at which the sub-policies DO something: `--rounds` outer
rounds of seqdex_amd/scripts/bi_optimization.py::block_assembly (forward Search -> Orient -> GraspSim -> InsertSim, three backward legs with a
transition-value refit after each; scripts/bi_optimization.py:110-124) at 4 096 envs, `mixed_precision: True`, GraspSim on 2 048-row
minibatches, every other schedule as shipped.  tests/test_gpu_bi_optimization_fullsize.py runs ONE round with
InsertSim legs of 48 + 32 epochs (1 insertion); here they are 1 500 + 800, GraspSim's 1 200 + 300.
usage: python tools/biopt_long.py [--rounds 2] [--num_envs 4096] --out file.json"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.bi_optimization import block_assembly  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--tvalue_rollout", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=None, help="the launcher's --seed for every training run (default: its 22) and the seed of the final chain")
    ap.add_argument("--epochs", default="search=20,orient=20,orient_backward=100,grasp=1200,grasp_backward=300,insert=1500,insert_backward=800")
    ap.add_argument("--chain_envs", type=int, default=1024, help="after the rounds: play Orient -> GraspSim -> InsertSim once from the final checkpoints under "
                    "the final transition value, gates as ladders that START at the reference's 0.99 / 0.8 (0 = skip)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    stage_epochs = {k: int(v) for k, v in (kv.split("=") for kv in a.epochs.split(","))}
    report = []
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="sdx_biopt_long_")
    os.chdir(tmp)
    paths, tv = {}, None
    t0 = time.time()
    err = None
    try:
        paths, tv = block_assembly(rounds=a.rounds, num_envs=a.num_envs, tvalue_rollout=a.tvalue_rollout, mixed_precision=True, report=report,
                                   stage_epochs=stage_epochs, grasp_harvest_stand_in=True, gates={"orient": 0.0, "grasp": 0.0},
                                   gates_after_fit={"orient": 0.5, "grasp": 0.28}, grasp_minibatch=2048, seed=a.seed)
    except Exception as ex:          # a round that breaks is a result too: what ran is in the report
        err = "%s: %s" % (type(ex).__name__, str(ex)[:600])
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    chain = None
    if a.chain_envs > 0 and err is None and tv is not None:
        # scripts/evaluation.py:111-119 on what the loop produced: the three sub-policies back to back under the loop's last transition value
        from seqdex_amd.scripts.evaluation import block_assembly_chain
        try:
            res, hand = block_assembly_chain(a.chain_envs, tv, policies={k: os.path.join(tmp, v) for k, v in paths.items() if k in ("orient", "grasp", "insert")},
                                             synthetic_fallback=False, orient_fallback=True, orient_tvalue_gate=(0.99, 0.9, 0.8, 0.5),
                                             grasp_tvalue_gate=(0.8, 0.65, 0.5), stage_steps={"grasp": 160}, min_grasp_states=100, max_grasp_steps=16000, seed=22 if a.seed is None else a.seed)
            hand["insert_task"].sim.close()
            chain = res
            chain["stand_ins"] = [x for x in (None if res["orient"]["tvalue_gate"] == 0.99 else "Orient's gate %s instead of 0.99" % res["orient"]["tvalue_gate"],
                                              None if res["grasp"]["tvalue_gate"] == 0.8 else "GraspSim's gate %s instead of 0.8" % res["grasp"]["tvalue_gate"],
                                              ("settled piles for Orient's groups %s" % res["orient"]["settled_stand_in_groups"]) if res["orient"].get("settled_stand_in_groups") else None) if x]
        except Exception as ex:
            chain = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:600])}
    runs = [r for r in report if "task" in r]
    hand = [r for r in report if "handoff" in r]
    fits = [h for h in hand if h["handoff"].startswith("T-value fitted")]
    out = {"config": "bi-optimisation, %d rounds at %d envs (Search 128, backward Orient 128), mixed_precision, GraspSim minibatch 2048, seed %s" % (a.rounds, a.num_envs, a.seed if a.seed is not None else "22 (the launcher's default)"),
           "stage_epochs": stage_epochs, "tvalue_fit_iterations": a.tvalue_rollout, "wall_s": time.time() - t0, "error": err,
           "env_steps": sum(r["env_steps"] for r in runs), "training_wall_s": sum(r["wall_s"] for r in runs),
           "tvalue_refits_performed": [h["handoff"] for h in fits if h.get("source") == "harvested"], "tvalue_refits_skipped": [h["handoff"] for h in fits if h.get("source") != "harvested"],
           "chain_played_from_the_final_checkpoints": chain, "runs": runs, "handoffs": hand,
           "stand_ins": ["harvest gates 0.0 before the first fit, 0.5 / 0.28 after it (reference: 0.99 / 0.8)",
                         "scripted grasp episodes / synthetic grasp states only where a hand-off says so (harvested_by, outcomes_by, grasp_states_source)"]}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1, default=str)
    for r in runs:
        print({k: r.get(k) for k in ("task", "leg", "num_envs", "epochs", "wall_s", "game_reward", "env_steps_per_s", "minibatch_size", "update_impl", "tvalue_outcomes_logged(success, failure)") if k in r})
    for h in hand:
        print({k: (v if len(str(v)) < 160 else str(v)[:160]) for k, v in h.items()})
    print(json.dumps({k: out[k] for k in ("wall_s", "error", "tvalue_refits_performed", "tvalue_refits_skipped")}))
    if chain is not None:
        print("chain:", json.dumps({k: (chain[k] if not isinstance(chain[k], dict) else {kk: vv for kk, vv in chain[k].items() if kk in (
            "tvalue_gate", "success_buf_mean", "piles_harvested_per_type", "grasp_states_harvested_per_type", "steps_per_env", "settled_stand_in_groups")}) for k in chain}, default=str)[:1500])
