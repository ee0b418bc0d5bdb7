#!/usr/bin/env python3
"""k_physics timing on one MI355X: average launch time at N envs (HIP events) and the phase clock of env 0 (SDX_T_DEBUG stamps).
usage: python tools/time_physics.py [N] [warm-steps]"""
import json
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
cfg["env"]["numEnvs"] = n
if os.environ.get("SDX_TP_ITERS"):          # ablation: solver iterations per substep (the YAML ships 16)
    cfg.setdefault("sim", {}).setdefault("physx", {})["num_position_iterations"] = int(os.environ["SDX_TP_ITERS"])
task = BlockAssemblyGraspSim(cfg, device_type="cuda", device_id=0, headless=True, seed=22, piles_per_type=8)
s = task.sim
g = torch.Generator().manual_seed(0)
for _ in range(warm):                         # random flailing: the contact-rich workload of the bench
    task.step((torch.rand(n, 23, generator=g) * 2 - 1).cuda())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    s.simulate()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
d = s.DEBUG.cpu().numpy().astype("int64")
nc = s.NCONTACTS.cpu().numpy()
ph = {"fk+inertia": d[1] - d[0], "mass_matrix": d[2] - d[1], "drive": d[3] - d[2], "collide": d[4] - d[3], "solve": d[5] - d[4],
      "integrate": d[6] - d[5], "mm_entries": d[34] - d[1], "mm_cholesky": d[35] - d[34], "mm_linv": d[36] - d[35], "mm_hinv": d[2] - d[36], "solve_setup": d[17] - d[16], "it_AC": d[20] - d[18],
      "setup_load": d[23] - d[16], "setup_count": d[24] - d[23], "setup_prefix": d[25] - d[24], "setup_fill": d[26] - d[25],
      "setup_rank": d[27] - d[26], "setup_link_inertia": d[29] - d[28], "setup_weights": d[30] - d[29],
      "setup_tail": d[17] - d[30], "broad_mask": d[32] - d[3], "broad_scan": d[33] - d[32], "expand_box_pairs": d[37] - d[33], "narrow": d[4] - d[37], "narrow_classify_first_chunk": d[38] - d[37], "narrow_rest_of_part1": d[39] - d[38], "narrow_part2_geometry": d[4] - d[39],
      "broad_scan_and_list": d[40] - d[32], "broad_bound_sat": d[41] - d[40], "broad_winner": d[42] - d[41], "broad_prefix": d[33] - d[42],
      "fk_constants": d[7] - d[0], "fk_poses": d[8] - d[7], "fk_axes": d[9] - d[8], "fk_twists": d[10] - d[9], "fk_al": d[11] - d[10], "fk_ao_wrenches": d[12] - d[11], "fk_boxes_inertia": d[1] - d[12],
      "prologue(load constants + state)": d[0] - d[13], "substep0": d[6] - d[0], "substep1": d[15] - d[6], "epilogue(final FK + outputs)": d[14] - d[15], "env_total": d[14] - d[13],
      "it_D": (d[21] if d[21] > d[20] else d[22]) - d[20], "it_robot": (d[22] - d[21]) if d[21] > d[20] else 0, "it_total": d[22] - d[18]}
cf = s.CONTACT.view(n, 165, 3)[:, :24].abs().sum(dim=(1, 2)).cpu().numpy()
sched = None
if len(d) >= 64 + 2 * n and d[64:].any():     # profiling build: (entry, exit) stamps of every env of the last launch
    import numpy as np
    t_in, t_out = d[64::2][:n].astype(np.float64), d[65::2][:n].astype(np.float64)
    dur = t_out - t_in                            # (the counters of the 8 XCDs have different origins: only differences inside an env mean something)
    heavy = cf > 0
    sched = {"sum_of_env_cycles_over_512_slots": float(dur.sum() / 512),
             "env_cycles_mean": float(dur.mean()), "env_cycles_p10_p50_p90_max": [float(x) for x in np.percentile(dur, [10, 50, 90, 100])],
             "env_cycles_mean_with_robot_contact": float(dur[heavy].mean()) if heavy.any() else None,
             "env_cycles_mean_without": float(dur[~heavy].mean()) if (~heavy).any() else None,
             "corr(env_cycles, contacts)": float(np.corrcoef(dur, nc)[0, 1])}
denv = int(os.environ.get("SDX_DEBUG_ENV", "0"))
print(json.dumps({"debug_env": denv, "debug_env_contacts": int(nc[denv]), "debug_env_has_robot_contact": bool(cf[denv] > 0),
                  "envs_with_robot_contact": int((cf > 0).sum()), "first_robot_contact_envs": [int(i) for i in (cf > 0).nonzero()[0][:6]],
                  "solver_iters": int(os.environ.get("SDX_TP_ITERS", 16)), "threads_per_env": int(s.lib.sdxk_physics_threads()), "n_envs": n, "k_physics_ms": ms,
                  "env_steps_per_s": n / (ms * 1e-3), "contacts_mean": float(nc.mean()), "contacts_max": int(nc.max()),
                  "counts_env_substep0": {"body_pairs_after_mask": int(d[48]), "after_bounding_sat": int(d[49]), "candidate_box_pairs": int(d[50]),
                                          "box_pairs_classified": int(d[51]), "contacts": int(d[52]), "csr_entries": int(d[53]), "robot_entries": int(d[54])},
                  "phase_cycles_env0_substep0": {k: int(v) for k, v in ph.items()}, "schedule": sched,
                  # maxima over the block's lanes in the first solver passes of the debug env (profiling build; since sdx_create)
                  "lane_max_first_passes": {"AC_cycles_slowest_lane": int(d[55]), "AC_cycles_last_lane": int(d[56]), "D_loop_cycles_slowest_lane": int(d[57]),
                                            "D_entries_per_lane_max": int(d[58]), "D_cycles_slowest_lane_to_barrier": int(d[59]), "csr_degree_max": int(d[60])}}))
