#!/usr/bin/env python3
"""Run-to-run determinism of the configs[2] chain: stage 0 (InsertSim training + T-value fit) and the Orient -> GraspSim ->
InsertSim chain of tests/test_gpu_chain.py, repeated in ONE process, each repetition from scratch; the harvest counts per brick-type group,
the T-value weights and the harvested tensors themselves must come out identical.

    python tools/chain_repeat.py [--reps 5] [--num_envs 1024] [--prep_epochs 1000] [--out profiles/r4_chain_repeat.json]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seqdex_amd.scripts.evaluation import CHAIN_GRASP_GATES, CHAIN_ORIENT_GATES, block_assembly_chain, prepare_tvalue_and_insert_policy, scripted_grasp_controller  # noqa: E402


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--num_envs", type=int, default=1024)
    ap.add_argument("--prep_epochs", type=int, default=1000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = []
    for rep in range(a.reps):
        t0 = time.time()
        tv, _, prep = prepare_tvalue_and_insert_policy(a.num_envs, a.prep_epochs, fit_iters=2000, seed=22)
        res, hand = block_assembly_chain(a.num_envs, tv, controllers={"grasp": scripted_grasp_controller}, synthetic_fallback=True,
                                         orient_tvalue_gate=CHAIN_ORIENT_GATES, grasp_tvalue_gate=CHAIN_GRASP_GATES, stage_steps={"grasp": 320})
        ins = hand["insert_task"]
        row = {"rep": rep, "wall_s": time.time() - t0, "tvalue_sha": hashlib.sha256(np.asarray(tv).tobytes()).hexdigest()[:16],
               "outcomes_logged": prep["outcomes_logged(success, failure)"], "tvalue_fit": prep["tvalue_fit"],
               "piles_harvested_per_type": res["orient"]["piles_harvested_per_type"],
               "settled_stand_in_groups": res["orient"].get("settled_stand_in_groups", []),
               "grasp_states_harvested_per_type": res["grasp"]["grasp_states_harvested_per_type"],
               "synthetic_insert_groups": ins.synthetic_groups, "piles_sha": digest(hand["piles"]),
               "grasp_obj_sha": [digest(o) for o in hand.get("grasp_obj", [])], "insert_success_buf_mean": res["insert"]["success_buf_mean"],
               "chain_env_steps_per_s": res["chain_env_steps_per_s"], "grasp_env_steps_per_s": res["grasp"]["env_steps_per_s"]}
        ins.sim.close()
        rows.append(row)
        print(json.dumps(row), flush=True)
    keys = ("tvalue_sha", "piles_harvested_per_type", "settled_stand_in_groups", "grasp_states_harvested_per_type", "piles_sha", "grasp_obj_sha")
    same = {k: all(r[k] == rows[0][k] for r in rows) for k in keys}
    out = {"reps": a.reps, "num_envs": a.num_envs, "identical_across_repetitions": same, "all_identical": all(same.values()),
           "groups_with_real_grasp_states": sum(c > 0 for c in rows[0]["grasp_states_harvested_per_type"]),
           "settled_stand_in_groups": rows[0]["settled_stand_in_groups"], "rows": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
