#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE MI355X: the full BlockAssembly Search -> Orient -> GraspSim -> InsertSim bi-optimisation loop
(seqdex_amd/scripts/bi_optimization.py, after the reference's scripts/bi_optimization.py:110-124) at num_envs = 4096 (Search at its 128,
the backward Orient leg at 128 as bi_optimization.py:123), `mixed_precision: True` in every stage's PPO YAML ("bf16 policy": bf16 MFMA
operands, fp32 master weights / accumulation, wherever the shipped schedule's update is GEMM-shaped - InsertSim's minibatch 4096; the
rank-4 schedules of the other three tasks run the fp32 persistent kernel) and each task's SHIPPED minibatch size.  One round: forward
initialisation of the four sub-policies, then the three backward legs with a transition-value refit after each.

The 8-GPU form of configs[4] cannot be run here (gpurun boxes have one GPU; no 8-GPU node was ever available to the driver): 4096 envs fit
one GPU, so this is the whole loop at its full env count on 1/8 of the hardware.

Stage lengths: an episode is 75 / 75 / 150 / 125 env steps, i.e. 10 / 10 / 19 / 16 epochs of horizon 8; shorter runs finish no episode,
harvest nothing and log no T-value outcome, so the defaults are search 20, orient 10, grasp 20, insert 48 (+ 32 backward) epochs.  A
transition-value refit needs more than 100 successful and at least one failed outcome (the trainer holds out 100 success rows): legs whose
few-epoch policy logged only one class are skipped and say so with their counts (InsertSim: no insertion yet; Orient: every episode passes
the lowered gate); the grasp leg's fit runs on the stand-in's outcomes.  What is a
stand-in and says so in the JSON: (1) in the first forward pass no transition value has been fitted yet (as in the reference, whose first
transition_value_trainer call comes after it), so the harvest gates are opened (0.0) and the physical criteria alone decide; once a value
exists the gates are 0.5 / 0.28, the chain benchmark's (a value fitted to a few hundred epochs of outcomes does not reach the reference's
0.99 / 0.8); (2) a grasp policy of 20 + 20 epochs carries two or three bricks to the insertion side (the reference's is from epoch 19 000),
so the grasp terminal states InsertSim starts from, and the successful outcomes of the backward grasp leg's fit, come from two episodes of
the scripted controller on the trained task; groups it leaves empty get synthetic states.

    python tools/bench_config5.py [--num_envs 4096] [--out profiles/r4_config5_biopt_n4096.json] [--fp32]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.bi_optimization import CONFIG5_EPOCHS as DEFAULT_EPOCHS, one_round_at_size as run  # noqa: E402,F401


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--grasp_minibatch", type=int, default=0, help="GraspSim legs on minibatches of this size for CONFIG5_LEARNED_EPOCHS (2048: the "
                    "policy learns to lift and no scripted stand-in plays; 0 = the shipped 4 and 20 epochs)")
    a = ap.parse_args()
    res, _, _ = run(a.num_envs, not a.fp32, grasp_minibatch=a.grasp_minibatch)
    print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as fh:
            fh.write(json.dumps(res, indent=1) + "\n")
