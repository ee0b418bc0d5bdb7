"""Further rounds of the bi-optimisation loop on the learned chain (scripts/bi_optimization.py:115-124 in small; exploration):
  round 0   insert policy + transition value from synthetic grasp states; grasp policy trained under that value
  round r   the grasp policy of round r-1 harvests grasp states from settled piles under gate 0.8 -> the insert policy is fine-tuned on them and the
            transition value is REFITTED to the outcomes of that run -> the grasp policy is fine-tuned under the refitted value
  after every round: Orient -> GraspSim -> InsertSim played with the round's three artefacts (evaluation.block_assembly_chain): the rung of
  Orient's gate ladder, grasp states per group, share of InsertSim's episodes that insert.
usage: python tools/loop_rounds_probe.py N rounds [insert_epochs 1500] [grasp_epochs 500]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.scripts.evaluation import (CHAIN_LEARNED_ORIENT_GATES, block_assembly_chain, main_rlgames, prepare_tvalue_and_insert_policy,  # noqa: E402
                                           train_grasp_policy)

n, rounds = int(sys.argv[1]), int(sys.argv[2])
ie = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
ge = int(sys.argv[4]) if len(sys.argv) > 4 else 500
work = tempfile.mkdtemp(prefix="sdx_loop_")


def play_chain(tag, tv, gpath, ipath):
    res, hand = block_assembly_chain(n, tv, policies={"grasp": gpath, "insert": ipath}, synthetic_fallback=False, orient_fallback=True,
                                     orient_tvalue_gate=CHAIN_LEARNED_ORIENT_GATES, grasp_tvalue_gate=0.8, stage_steps={"grasp": 160},
                                     min_grasp_states=1, max_grasp_steps=16000, seed=22)
    ins = hand["insert_task"]
    out = {"round": tag, "orient_gate": res["orient"]["tvalue_gate"], "orient_piles": res["orient"]["piles_harvested_per_type"],
           "orient_gates_tried": res["orient"].get("tvalue_gates_tried"), "settled_stand_in_groups": res["orient"].get("settled_stand_in_groups", []),
           "grasp_steps_per_env": res["grasp"]["steps_per_env"], "grasp_states": res["grasp"]["grasp_states_harvested_per_type"],
           "insert_success": float(ins.extras["success_buf"].float().mean()), "synthetic_groups": ins.synthetic_groups}
    ins.sim.close()
    print("CHAIN", json.dumps(out), flush=True)


tv, ipath, ist = prepare_tvalue_and_insert_policy(n, 1500, save_to=os.path.join(work, "insert0"))
print("round 0 insert:", json.dumps(ist, default=float), flush=True)
gpath, gtask, gst = train_grasp_policy(n, 1500, save_to=os.path.join(work, "grasp0"), tvalue_state=tv)
gtask.sim.close()
print("round 0 grasp:", json.dumps(gst), flush=True)
for r in range(1, rounds + 1):
    g0, st0 = main_rlgames("BlockAssemblyGraspSim", n, policy_path=gpath, tvalue_state=tv, steps=160, seed=22 + r,
                           until=lambda t: int(t.sim.HARVEST_COUNT.min()) >= 64, max_steps=16000, task_kwargs={"harvest_tvalue_gate": 0.8})
    cnt = g0.sim.HARVEST_COUNT.cpu().tolist()
    states = g0.grasp_terminal_states()
    g0.sim.close()
    print("round %d harvest (settled piles, gate 0.8, %d steps per env): %s" % (r, st0["steps_per_env"], cnt), flush=True)
    tv_new, ipath, ist = prepare_tvalue_and_insert_policy(n, ie, save_to=os.path.join(work, "insert%d" % r), grasp_states=states, restore=ipath,
                                                          synthetic_fallback=True)
    print("round %d insert (fine-tuned on the harvest, value refitted to its outcomes):" % r, json.dumps(ist, default=float), flush=True)
    if tv_new is not None:
        tv = tv_new
    gpath, gtask, gst = train_grasp_policy(n, ge, save_to=os.path.join(work, "grasp%d" % r), tvalue_state=tv, restore=gpath)
    gtask.sim.close()
    print("round %d grasp (fine-tuned under the refitted value):" % r, json.dumps(gst), flush=True)
    play_chain(r, tv, gpath, ipath)
