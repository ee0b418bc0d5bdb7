#!/usr/bin/env python3
"""BASELINE.json configs[2]: BlockAssemblyOrient -> BlockAssemblyGraspSim -> BlockAssemblyInsertSim as ONE chained rollout at num_envs = 1024
on one MI355X, the stages handing over their harvested terminal states (seqdex_amd/scripts/evaluation.py, after the reference's
scripts/evaluation.py:111-119; harvest rules OR:1463-1488, GS:1404-1417; InsertSim's reset from them IS:372-375).

stage 0 (untimed; the backward pass of scripts/bi_optimization.py:120-121 in small): BlockAssemblyInsertSim trains `prep_epochs` epochs with its
    shipped schedule from synthetic grasp states, its episode outcomes fill the T-value rings, GraspInsertTValue is fitted to them -> the
    transition value that gates the harvests of the chain (policy_sequencing's T-value switch), and the insert policy of stage 3.
stage 1 Orient (random-initialised policy: its arm is scripted by the task, OR:1720-1778) plays until every brick-type group has >= 8 piles;
    its T-value gate is lowered from 0.99 (OR:1203) to 0.5: a T-value fitted to a thousand epochs of outcomes tops out near 0.85,
stage 2 GraspSim starts from those piles (two episodes; harvest gate 0.28 instead of 0.8, GS:1406, for the same reason); a scripted stand-in
    controller replaces the 19 000-epoch grasp policy (evaluation.py docstring); groups it harvests nothing for get InsertSim's synthetic states,
stage 3 InsertSim resets from the harvested grasp states and plays one episode.
value = env-steps of the three rollouts / their wall time.  Prints one JSON line.   usage: python tools/bench_config3.py [N] [prep_epochs] [--with-search] [--out file.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import CHAIN_GRASP_GATES, CHAIN_ORIENT_GATES, block_assembly_chain, prepare_tvalue_and_insert_policy, scripted_grasp_controller  # noqa: E402


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
    prep = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1200
    import tempfile
    tmp = tempfile.mkdtemp(prefix="sdx_config3_")          # the stage-0 checkpoint (30 MB) is a hand-off inside this run, not an artefact
    tv, insert_ckpt, prep_st = prepare_tvalue_and_insert_policy(n, prep, seed=22, save_to=os.path.join(tmp, "config3_insert_policy"))
    print("stage 0:", json.dumps(prep_st), file=sys.stderr, flush=True)
    res, hand = block_assembly_chain(n, tv, policies={"insert": insert_ckpt}, controllers={"grasp": scripted_grasp_controller},
                                     synthetic_fallback=True, orient_tvalue_gate=CHAIN_ORIENT_GATES, grasp_tvalue_gate=CHAIN_GRASP_GATES,
                                     stage_steps={"grasp": 320}, with_search="--with-search" in sys.argv)
    ins = hand["insert_task"]
    res["insert"]["synthetic_groups"] = ins.synthetic_groups
    ins.sim.close()
    out = {"config": "BASELINE.json configs[2]: BlockAssemblyOrient -> BlockAssemblyGraspSim -> BlockAssemblyInsertSim chained rollout, num_envs=%d, 1 GPU" % n,
           "metric": "env-steps/s of the chained rollout (play, no update)", "value": res["chain_env_steps_per_s"], "unit": "env-steps/s",
           "stage0_tvalue_and_insert_policy(untimed)": prep_st, "chain": res}
    print(json.dumps(out), flush=True)
    for i, a in enumerate(sys.argv):
        if a == "--out" and i + 1 < len(sys.argv):
            with open(sys.argv[i + 1], "w") as fh:
                fh.write(json.dumps(out) + "\n")
