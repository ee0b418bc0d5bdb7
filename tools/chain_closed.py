#!/usr/bin/env python3
"""BASELINE.json configs[2] with every stage on a learned policy, several seeds:
seqdex_amd/scripts/evaluation.py::block_assembly_chain_closed per seed, then mean / min / max of the numbers that matter.
usage: python tools/chain_closed.py [N] --seeds 22,23,24 [--insert-epochs 1500] [--grasp-epochs 1500] [--refit-epochs 4000] [--orient-epochs 600]
       [--gates 0.99,0.9,0.8] [--min-grasp-states 100] --out file.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import block_assembly_chain_closed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=1024)
    ap.add_argument("--seeds", default="22,23,24")
    ap.add_argument("--insert-epochs", type=int, default=1500)
    ap.add_argument("--grasp-epochs", type=int, default=1500)
    ap.add_argument("--refit-epochs", type=int, default=4000)
    ap.add_argument("--orient-epochs", type=int, default=600)
    ap.add_argument("--gates", default="0.99,0.9,0.8,0.5")
    ap.add_argument("--grasp-gates", default="0.8,0.65,0.5", help="gate ladder of the grasp harvests (GS:1406 ships 0.8; a lower rung is reported as a stand-in)")
    ap.add_argument("--min-grasp-states", type=int, default=100)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    runs = []
    for seed in [int(x) for x in a.seeds.split(",")]:
        t0 = time.time()
        try:
            out, hand = block_assembly_chain_closed(a.n, a.insert_epochs, a.grasp_epochs, a.refit_epochs, a.orient_epochs, seed=seed,
                                                    min_grasp_states=a.min_grasp_states, orient_gates=tuple(float(g) for g in a.gates.split(",")),
                                                    grasp_gates=tuple(float(g) for g in a.grasp_gates.split(",")))
            hand["insert_task"].sim.close()
            out["seed"], out["total_wall_s"] = seed, time.time() - t0
        except Exception as ex:       # a seed whose pipeline breaks is a result too
            out = {"seed": seed, "error": "%s: %s" % (type(ex).__name__, str(ex)[:600]), "total_wall_s": time.time() - t0}
        runs.append(out)
        ok = [r for r in runs if "chain" in r]

        def stat(f):
            v = [f(r) for r in ok]
            return {"mean": float(np.mean(v)), "min": float(np.min(v)), "max": float(np.max(v)), "values": v} if v else None
        summary = {"seeds_run": [r["seed"] for r in runs], "seeds_failed": [r["seed"] for r in runs if "chain" not in r],
                   "grasp_game_reward": stat(lambda r: r["grasp_policy(untimed)"]["game_reward"]),
                   "insert_refit_success_share_of_last_episodes": stat(lambda r: r["insert_policy_refit_and_tvalue_refit(untimed)"]["insert_success_buf_mean"]),
                   "chain_insert_success_share": stat(lambda r: r["chain"]["insert"]["success_buf_mean"]),
                   "chain_orient_gate_used": [r["chain"]["orient"]["tvalue_gate"] for r in ok],
                   "chain_grasp_states_handed_on": stat(lambda r: float(np.sum(r["chain"]["grasp"]["grasp_states_harvested_per_type"]))),
                   "chain_env_steps_per_s": stat(lambda r: r["chain"]["chain_env_steps_per_s"]),
                   "orient_game_reward": stat(lambda r: r["orient_policy(untimed)"]["game_reward"]),
                   "tvalue_max_over_random_orientations(refitted)": stat(lambda r: (r["tvalue_over_200000_random_orientations"]["refitted"] or {"max": float("nan")})["max"]),
                   "stand_ins": [r["stand_ins"] for r in ok]}
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump({"num_envs": a.n, "args": vars(a), "summary": summary, "runs": runs}, open(a.out, "w"), indent=1, default=str)
        print(json.dumps({"seed": seed, "wall_s": round(time.time() - t0, 1), "error": out.get("error"),
                          "chain_insert_share": None if "chain" not in out else out["chain"]["insert"]["success_buf_mean"],
                          "orient_gate": None if "chain" not in out else out["chain"]["orient"]["tvalue_gate"]}), flush=True)
    print(json.dumps(summary)[:3000])


if __name__ == "__main__":
    main()
