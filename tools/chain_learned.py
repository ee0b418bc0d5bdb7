"""BASELINE.json configs[2] on LEARNED policies (seqdex_amd/scripts/evaluation.py::block_assembly_chain_learned): insert
policy + transition value from stage 0, a GraspSim policy of this engine trained under that transition value's gate, then Orient ->
GraspSim -> InsertSim with no scripted stage and no synthetic grasp states.  Prints one JSON line.
usage: python tools/chain_learned.py [N] [grasp_epochs] [insert_epochs] [--refit insert_refit_epochs] [--out file]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import block_assembly_chain_learned  # noqa: E402

if __name__ == "__main__":
    pos = [a for i, a in enumerate(sys.argv[1:], 1) if a.isdigit() and sys.argv[i - 1] not in ("--refit", "--out")]
    n = int(pos[0]) if len(pos) > 0 else 1024
    ge = int(pos[1]) if len(pos) > 1 else 1500
    ie = int(pos[2]) if len(pos) > 2 else 1500
    t0 = time.time()
    refit = int(sys.argv[sys.argv.index("--refit") + 1]) if "--refit" in sys.argv else 1500
    out, hand = block_assembly_chain_learned(n, ge, ie, insert_refit_epochs=refit)
    ins = hand["insert_task"]
    out["chain"]["insert"]["synthetic_groups"] = ins.synthetic_groups
    out["chain"]["insert"]["insert_success_buf_mean"] = float(ins.extras["success_buf"].float().mean())
    ins.sim.close()
    out = dict({"config": "BASELINE.json configs[2]: BlockAssemblyOrient -> BlockAssemblyGraspSim -> BlockAssemblyInsertSim chained rollout on learned "
                          "grasp / insert policies, num_envs=%d, 1 GPU" % n, "metric": "env-steps/s of the chained rollout (play, no update)",
                "value": out["chain"]["chain_env_steps_per_s"], "unit": "env-steps/s", "total_wall_s": time.time() - t0}, **out)
    print(json.dumps(out), flush=True)
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as fh:
            fh.write(json.dumps(out) + "\n")
