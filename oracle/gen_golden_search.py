#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of the pure per-step tensor code of the reference's BlockAssemblySearch task
(first policy of the chain; SURVEY.md section 8(f) rank 3 - the task itself is NOT built: it needs a segmentation rasteriser and its
own reset / history logic), produced by RUNNING THE REFERENCE'S OWN functions in this container.  Same harness as oracle/gen_golden.py.

  python oracle/gen_golden_search.py      # needs /root/reference; writes tests/golden/S*.npz

SE = dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_search.py.  Fixtures (data only):
  S2 pre_physics.npz   SE:1539-1596  fingers with moving average 0.6 + clamp, arm by the IK that tracks the target brick (0.24 above, 0.18 behind)
  S3 observations.npz  SE:1220-1245 (62 numbers + the pixel statistics of the segmentation image) and SE:1168-1218 (asymmetric states)
  S5 reward.npz        SE:1660-1711 compute_hand_reward; SE:1640-1652 emergence reward / heap movement count
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

N, A, NB, OUT = G.N, G.A, G.NB, G.OUT


def import_search():
    import importlib
    G.import_reference()
    se = importlib.import_module("tasks.block_assembly.allegro_hand_block_assembly_search")
    se.gymtorch.unwrap_tensor = lambda t: t
    return se


def s2_pre_physics(se, g, scene):
    lo, hi = G.limits(scene)
    ns = types.SimpleNamespace()
    ns.gym, ns.sim, ns.device, ns.num_envs = G.FakeGym(), None, "cpu", N
    ns.reset_buf = torch.zeros(N, dtype=torch.long)
    ns.reset_goal_buf = torch.zeros(N, dtype=torch.long)
    ns.test_robot_controller = False
    ns.apply_teleoper_perturbation = False
    ns.actuated_dof_indices = torch.arange(7, 23)
    ns.arm_hand_dof_lower_limits, ns.arm_hand_dof_upper_limits = lo, hi
    ns.act_moving_average = 0.6
    q = lo + (hi - lo) * torch.rand(N, 23, generator=g)
    ns.arm_hand_dof_pos = q.clone()
    ns.prev_targets = lo + (hi - lo) * torch.rand(N, 23, generator=g)
    ns.cur_targets = torch.zeros(N, 23)
    ns.rigid_body_states = torch.randn(N, NB, 13, generator=g) * 0.3
    ns.rigid_body_states[:, :, 3:7] = G.rand_quat(g, N * NB).view(N, NB, 4)
    ns.hand_base_rigid_body_index = 7
    ns.segmentation_target_pos = ns.rigid_body_states[:, 7, 0:3] + torch.randn(N, 3, generator=g) * 0.1
    ns.jacobian_tensor = torch.randn(N, 23, 6, 23, generator=g)
    actions = torch.rand(N, 23, generator=g) * 2 - 1
    out = dict(actions=actions.numpy(), q=q.numpy(), prev_targets=ns.prev_targets.numpy().copy(),
               hand_pos=ns.rigid_body_states[:, 7, 0:3].numpy().copy(), hand_rot=ns.rigid_body_states[:, 7, 3:7].numpy().copy(),
               target_pos=ns.segmentation_target_pos.numpy().copy(), J=ns.jacobian_tensor[:, 6, :, :7].numpy().copy())
    se.BlockAssemblySearch.pre_physics_step(ns, actions)
    name, args = ns.gym.calls[-1]
    assert name == "set_dof_position_target_tensor"
    out.update(cur_targets=ns.cur_targets.numpy().copy(), sim_targets=args[0].numpy().copy(), lower=lo.numpy(), upper=hi.numpy(),
               euler=ns.now_euler_angle.numpy().copy())
    np.savez(os.path.join(OUT, "S2_pre_physics.npz"), **out)


def s3_observations(se, g, scene):
    lo, hi = G.limits(scene)
    ns = types.SimpleNamespace()
    ns.num_envs, ns.device = N, "cpu"
    ns.arm_hand_dof_lower_limits, ns.arm_hand_dof_upper_limits = lo, hi
    ns.vel_obs_scale = 0.2
    dof = torch.stack([lo + (hi - lo) * torch.rand(N, 23, generator=g), torch.randn(N, 23, generator=g) * 3.0], dim=-1)
    ns.arm_hand_dof_pos, ns.arm_hand_dof_vel = dof[..., 0], dof[..., 1]
    ns.actions = torch.rand(N, 23, generator=g) * 2 - 1
    ns.obs_buf = torch.zeros(N, 186)
    ns.states_buf = torch.zeros(N, 564)
    # segmentation images (IMAGE_SEGMENTATION, 128 x 128 int32) with ids 0..8 in blobs; the target id of env i is i % 8 + 1
    seg = torch.zeros(N, 128, 128, dtype=torch.int32)
    ids = torch.arange(N, dtype=torch.int32) % 8 + 1
    for i in range(N):
        for b in range(5):
            cx, cy, r = [int(v) for v in torch.randint(10, 118, (3,), generator=g)]
            r = 3 + r % 9
            seg[i, max(cx - r, 0):cx + r, max(cy - r, 0):cy + r] = (i + b) % 8 + 1
    seg[3] = torch.where(seg[3] == ids[3], torch.zeros_like(seg[3]), seg[3])       # one env whose target is not visible
    ns.camera_seg_tensors = [seg[i] for i in range(N)]
    ns.segmentation_id_list = [int(v) for v in ids]
    ns.segmentation_object_center_point_x = torch.zeros(N, 1, dtype=torch.int)
    ns.segmentation_object_center_point_y = torch.zeros(N, 1, dtype=torch.int)
    ns.segmentation_object_point_num = torch.zeros(N, 1, dtype=torch.int)
    se.BlockAssemblySearch.compute_contact_observations(ns, False)
    out = dict(dof=dof.numpy(), actions=ns.actions.numpy(), seg=seg.numpy().astype(np.int16), ids=ids.numpy(), lower=lo.numpy(), upper=hi.numpy(),
               obs_buf=ns.obs_buf.numpy().copy(), center_x=ns.segmentation_object_center_point_x.numpy().copy(),
               center_y=ns.segmentation_object_center_point_y.numpy().copy(), point_num=ns.segmentation_object_point_num.numpy().copy())
    # asymmetric states: every attribute it reads is an input
    rb = torch.randn(N, 13 * 6, generator=g)
    names = ["arm_hand_ff_pos", "arm_hand_rf_pos", "arm_hand_mf_pos", "arm_hand_th_pos"]
    for k, nm in enumerate(names):
        setattr(ns, nm, torch.randn(N, 3, generator=g))
    ns.hand_base_pose = torch.randn(N, 7, generator=g)
    ns.segmentation_target_pose = torch.randn(N, 7, generator=g)
    for k in range(8):
        setattr(ns, "hand_pos_history_%d" % k, torch.randn(N, 3, generator=g))
    ns.hand_base_linvel, ns.hand_base_angvel = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    for f in ("ff", "mf", "rf", "th"):
        setattr(ns, "arm_hand_%s_rot" % f, G.rand_quat(g, N))
        setattr(ns, "arm_hand_%s_linvel" % f, torch.randn(N, 3, generator=g))
        setattr(ns, "arm_hand_%s_angvel" % f, torch.randn(N, 3, generator=g))
    ns.segmentation_target_linvel, ns.segmentation_target_angvel = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    se.BlockAssemblySearch.compute_contact_asymmetric_observations(ns)
    for nm in names + ["hand_base_pose", "segmentation_target_pose", "hand_base_linvel", "hand_base_angvel", "segmentation_target_linvel",
                       "segmentation_target_angvel"] + ["hand_pos_history_%d" % k for k in range(8)] + \
            ["arm_hand_%s_%s" % (f, w) for f in ("ff", "mf", "rf", "th") for w in ("rot", "linvel", "angvel")]:
        out["in_" + nm] = getattr(ns, nm).numpy().copy()
    out["states_buf"] = ns.states_buf.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "S3_observations.npz"), **out)


def s5_reward(se, g):
    M = 64
    tgt = torch.rand(M, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.7])
    init = tgt + torch.randn(M, 3, generator=g) * 0.05
    spread = torch.cat([torch.full((M // 2,), 0.03), torch.full((M // 2,), 0.25)])[:, None]
    tips = [tgt + torch.randn(M, 3, generator=g) * spread for _ in range(4)]
    progress = torch.tensor(([3, 73, 74, 75, 60, 10, 74, 20] * (M // 8)), dtype=torch.long)
    reset_buf = torch.zeros(M, dtype=torch.long)
    reset_buf[5::16] = 1
    successes = torch.zeros(M)
    successes[::7] = 1.0
    cons = torch.tensor([0.21])
    arm_contacts = (torch.rand(M, 6, generator=g) > 0.8).float()
    palm_z = torch.randn(M, generator=g) * 30
    actions = torch.rand(M, 23, generator=g) * 2 - 1
    emergence = torch.randn(M, generator=g) * 50
    heap = torch.randint(0, 20, (M,), generator=g).float()
    init_heap = torch.randint(0, 5, (M,), generator=g).float()
    pix = torch.randint(0, 200, (M,), generator=g).int()
    tvalue = torch.rand(M, generator=g)
    emergence_in = emergence.clone()
    rew, resets, rgoal, prog, succ, cons_out = se.compute_hand_reward(
        torch.tensor(1.0), torch.zeros(M), reset_buf, torch.zeros(M, dtype=torch.long), progress, successes, cons, 45, arm_contacts, palm_z,
        pix, init, 75.0, torch.zeros(M, 3), G.rand_quat(g, M), torch.zeros(M, 3), torch.zeros(M, 3), G.rand_quat(g, M), tgt,
        torch.zeros(M, 3), emergence, tips[0], tips[1], tips[2], tips[3], heap, -1.0, 1.0, 0.1, actions, -0.0, 0.1, 250.0, 0.4, 0.0, 1, 0, 0.1,
        False, init_heap, tvalue)
    # emergence reward / heap movement (SE:1640-1652) on small inputs
    ns = types.SimpleNamespace()
    ns.num_envs = 8
    seg = torch.randint(0, 6, (8, 32, 32), generator=g).int()
    ids = [int(v) for v in torch.arange(8) % 5 + 1]
    ns.emergence_pixel = torch.zeros(8)
    ns.last_emergence_pixel = torch.randint(0, 300, (8,), generator=g).float()
    last_in = ns.last_emergence_pixel.clone()
    se.BlockAssemblySearch.compute_emergence_reward(ns, None, [seg[i] for i in range(8)], segmentation_id_list=ids)
    pos = torch.randn(8, 132, 3, generator=g) * torch.tensor([0.4, 0.4, 0.1]) + torch.tensor([1.0, 0.0, 0.6])
    ns.all_lego_brick_pos = pos
    se.BlockAssemblySearch.compute_heap_movement_penalty(ns, pos)
    np.savez(os.path.join(OUT, "S5_reward.npz"), target_pos=tgt.numpy(), init_pos=init.numpy(), ff=tips[0].numpy(), rf=tips[1].numpy(),
             mf=tips[2].numpy(), th=tips[3].numpy(), progress=progress.numpy(), reset_buf=reset_buf.numpy(), successes=successes.numpy(),
             cons_in=cons.numpy(), arm_contacts=arm_contacts.numpy(), actions=actions.numpy(), emergence_in=emergence_in.numpy(),
             emergence_after=emergence.numpy(), reward=rew.numpy(), resets=resets.numpy(), cons_out=cons_out.numpy(),
             max_episode_length=np.array(75.0), em_seg=seg.numpy().astype(np.int16), em_ids=np.array(ids), em_last=last_in.numpy(),
             em_pixel=ns.emergence_pixel.numpy().copy(), em_reward=ns.emergence_reward.numpy().copy(), heap_pos=pos.numpy(),
             heap_penalty=ns.heap_movement_penalty.numpy().copy())


def main():
    os.makedirs(OUT, exist_ok=True)
    se = import_search()
    scene = G.load_scene()
    g = torch.Generator().manual_seed(31)
    s2_pre_physics(se, g, scene)
    s3_observations(se, g, scene)
    s5_reward(se, g)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("S"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
