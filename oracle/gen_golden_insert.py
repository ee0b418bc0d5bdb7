#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of the reference's BlockAssemblyInsertSim per-step tensor code
(BASELINE.json configs[2], second task of the chain), produced by RUNNING THE REFERENCE'S OWN
functions in this container.  Same harness as oracle/gen_golden.py / gen_golden_orient.py.

  python oracle/gen_golden_insert.py      # needs /root/reference; writes tests/golden/I*.npz

IS = dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_insert_sim.py.  Fixtures (data only):
  I2 pre_physics.npz     IS:1496-1575  position-only action + fixed wrist orientation IK, rot_err kept for the reward
  I3 observations.npz    IS:1087-1218 + compute_contact_observations IS:1280-1298 + asymmetric states IS:1220-1278, 3 calls
  I5 reward.npz          IS:1640-1695
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


def import_insert():
    import importlib
    gs, vr, tv = G.import_reference()
    ins = importlib.import_module("tasks.block_assembly.allegro_hand_block_assembly_insert_sim")
    ins.gymtorch.unwrap_tensor = lambda t: t
    return ins, tv


def offset_index_sets():
    """IS:779-812 (use_unseen False): which envs get which placement offsets of the base-plate target."""
    xn = [i for i in range(N) if i % 8 in (0, 1, 2, 6, 3, 4, 7)]
    x1 = [i for i in range(N) if i % 8 == 5]
    h = [[i for i in range(N) if i % 3 == k] for k in range(3)]
    return xn, x1, h


def i2_pre_physics(ins, g, scene):
    lo, hi = G.limits(scene)
    ns = types.SimpleNamespace()
    ns.gym, ns.sim, ns.device, ns.num_envs = G.FakeGym(), None, "cpu", N
    ns.reset_buf = torch.zeros(N, dtype=torch.long)
    ns.reset_goal_buf = torch.zeros(N, dtype=torch.long)
    ns.test_robot_controller = False
    ns.use_teleoperation = False
    ns.apply_teleoper_perturbation = False
    ns.actuated_dof_indices = torch.arange(7, 23)
    ns.arm_hand_dof_lower_limits, ns.arm_hand_dof_upper_limits = lo, hi
    ns.act_moving_average = 1.0
    q = lo + (hi - lo) * torch.rand(N, 23, generator=g)
    ns.arm_hand_dof_pos = q.clone()
    ns.prev_targets = lo + (hi - lo) * torch.rand(N, 23, generator=g)
    ns.cur_targets = torch.zeros(N, 23)
    ns.rigid_body_states = torch.randn(N, NB, 13, generator=g) * 0.3
    ns.rigid_body_states[:, :, 3:7] = G.rand_quat(g, N * NB).view(N, NB, 4)
    ns.hand_base_rigid_body_index = 7
    ns.jacobian_tensor = torch.randn(N, 23, 6, 23, generator=g)
    te = [l for l in open(ins.__file__) if "self.target_euler = to_torch" in l]
    ns.target_euler = torch.tensor(eval(te[0].split("to_torch(")[1].split("]")[0] + "]")).repeat(N, 1)
    actions = torch.rand(N, 23, generator=g) * 2 - 1
    out = dict(actions=actions.numpy(), q=q.numpy(), prev_targets=ns.prev_targets.numpy().copy(),
               hand_rot=ns.rigid_body_states[:, 7, 3:7].numpy().copy(), J=ns.jacobian_tensor[:, 6, :, :7].numpy().copy(),
               target_euler=ns.target_euler.numpy().copy())
    ins.BlockAssemblyInsertSim.pre_physics_step(ns, actions)
    name, args = ns.gym.calls[-1]
    assert name == "set_dof_position_target_tensor"
    out.update(cur_targets=ns.cur_targets.numpy().copy(), prev_targets_out=ns.prev_targets.numpy().copy(),
               sim_targets=args[0].numpy().copy(), rot_err=ns.rot_err.numpy().copy(), lower=lo.numpy(), upper=hi.numpy())
    np.savez(os.path.join(OUT, "I2_pre_physics.npz"), **out)


def i3_observations(ins, tv, g, scene):
    lo, hi = G.limits(scene)
    ns = G.make_obs_namespace(ins, tv, g, scene)
    ns.one_frame_num_obs, ns.one_frame_num_states = 75, 188
    ns.obs_buf = torch.zeros(N, 75)                       # stack_obs = 1 (IS:172)
    ns.states_buf = torch.zeros(N, 188)
    ns.obs_buf_stack_frames = [torch.zeros(N, 75)]
    ns.state_buf_stack_frames = [torch.zeros(N, 188)]
    ns.use_temporal_tvalue = False
    ns.max_episode_length = 125
    xn, x1, h = offset_index_sets()
    ns.extra_1xn_lego_pos_offset_indices, ns.extra_1x1_lego_pos_offset_indices = xn, x1
    ns.extra_height_lego_pos_offset_indices_0, ns.extra_height_lego_pos_offset_indices_1 = h[0], h[1]
    ns.extra_height_lego_pos_offset_indices_2 = h[2]
    ns.compute_contact_observations = lambda full=False: ins.BlockAssemblyInsertSim.compute_contact_observations(ns, full)
    ns.compute_contact_asymmetric_observations = lambda: ins.BlockAssemblyInsertSim.compute_contact_asymmetric_observations(ns)
    out = {}
    ns.segmentation_target_init_pos = torch.rand(N, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.65])
    ns.segmentation_target_init_rot = G.rand_quat(g, N)
    out["init_pos"], out["init_rot"] = ns.segmentation_target_init_pos.numpy(), ns.segmentation_target_init_rot.numpy()
    out["seg_index_in_env"] = (ns.lego_segmentation_indices - torch.arange(N) * A).numpy()
    for c in range(3):
        root = torch.randn(N * A, 13, generator=g) * 0.2
        root[:, 3:7] = G.rand_quat(g, N * A)
        root[ns.hand_indices, 0:3] = torch.tensor(scene["robot"]["base_pos"])
        root[ns.hand_indices, 3:7] = torch.tensor([0.0, 0, 0, 1])
        root[ns.lego_segmentation_indices, 0:3] += torch.tensor([0.2, 0.2, 0.7])
        root[ns.extra_object_indices, 0:3] = torch.tensor([0.1, -0.2, 0.618]) + torch.randn(N, 3, generator=g) * 0.01
        rb = torch.randn(N, NB, 13, generator=g) * 0.3
        rb[:, :, 3:7] = G.rand_quat(g, N * NB).view(N, NB, 4)
        rb[:, :, 0:3] += torch.tensor([0.2, 0.2, 0.8])
        dof = torch.stack([lo + (hi - lo) * torch.rand(N, 23, generator=g), torch.randn(N, 23, generator=g) * 3.0], dim=-1)
        contact = torch.randn(N, NB * 3, generator=g) * 0.08
        actions = torch.rand(N, 23, generator=g) * 2 - 1
        root_in = root.clone()
        ns.root_state_tensor, ns.rigid_body_states, ns.contact_tensor, ns.actions = root, rb, contact, actions
        ns.arm_hand_dof_pos, ns.arm_hand_dof_vel = dof[..., 0], dof[..., 1]
        ns.progress_buf = torch.randint(0, 125, (N,), generator=g)
        with torch.no_grad():
            ins.BlockAssemblyInsertSim.compute_observations(ns)
        pre = "c%d_" % c
        out[pre + "root"], out[pre + "rb"], out[pre + "dof"] = root_in.numpy(), rb.numpy(), dof.numpy()
        out[pre + "root_after"] = root.numpy().copy()          # the extra-target offsets are written back into the root tensor view
        out[pre + "contact"], out[pre + "actions"] = contact.numpy(), actions.numpy()
        out[pre + "obs_buf"], out[pre + "states_buf"] = ns.obs_buf.numpy().copy(), ns.states_buf.numpy().copy()
        out[pre + "extra_target_pos"] = ns.extra_target_pos.numpy().copy()
        out[pre + "symmetry_extra_target_rot"] = ns.symmetry_extra_target_rot.numpy().copy()
        out[pre + "finger_dist"] = ns.arm_hand_finger_dist.numpy().copy()
    out["lower"], out["upper"] = lo.numpy(), hi.numpy()
    np.savez_compressed(os.path.join(OUT, "I3_observations.npz"), **out)


def i5_reward(ins, g):
    M = 64
    tgt = torch.rand(M, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.7])
    extra = tgt + torch.randn(M, 3, generator=g) * torch.cat([torch.full((M // 2,), 0.01), torch.full((M // 2,), 0.1)])[:, None]
    rot = G.rand_quat(g, M)
    extra_rot = rot.clone()
    extra_rot[M // 4:] = G.rand_quat(g, M - M // 4)
    zq = torch.tensor([0.0, 0.0, 1.0, 0.0]).repeat(M, 1)
    extra_rot[M // 4:M // 2] = ins.quat_mul(rot, zq)[M // 4:M // 2]                          # plate turned by 180 degrees: the twin counts as aligned
    sym = ins.quat_mul(extra_rot, zq)                                                        # as compute_observations builds it, IS:1167
    spread = torch.cat([torch.full((M // 2,), 0.03), torch.full((M // 2,), 0.25)])[:, None]
    tips = [tgt + torch.randn(M, 3, generator=g) * spread for _ in range(4)]
    progress = torch.tensor(([3, 74, 123, 124, 125, 60, 10, 90] * (M // 8)), dtype=torch.long)
    reset_buf = torch.zeros(M, dtype=torch.long)
    reset_buf[5::16] = 1
    successes = torch.zeros(M)
    successes[::7] = 1.0
    cons = torch.tensor([0.37])
    rot_err = torch.randn(M, 3, generator=g) * 0.1
    angvel = torch.randn(M, 3, generator=g) * 20
    rew, resets, rgoal, prog, succ, cons_out = ins.compute_hand_reward(
        torch.tensor(1.0), torch.zeros(M), reset_buf, torch.zeros(M, dtype=torch.long), progress, successes, cons, 0,
        torch.rand(M, 6, generator=g), rot, extra, extra_rot, sym, rot_err,
        125.0, torch.zeros(M, 3), rot, angvel, torch.zeros(M, 3), rot, tgt, torch.zeros(M, 3), torch.zeros(M),
        tips[0], tips[1], tips[2], tips[3], torch.zeros(M), tgt.clone(), -1.0, 1.0, 0.1, torch.zeros(M, 23), -0.0,
        0.1, 250.0, 0.4, 0.0, 1, 0, 0.1, False, torch.zeros(M, 3), angvel)
    np.savez(os.path.join(OUT, "I5_reward.npz"), target_pos=tgt.numpy(), target_rot=rot.numpy(), extra_pos=extra.numpy(),
             extra_rot=extra_rot.numpy(), symmetry_rot=sym.numpy(), rot_err=rot_err.numpy(), ff=tips[0].numpy(), rf=tips[1].numpy(),
             mf=tips[2].numpy(), th=tips[3].numpy(), progress=progress.numpy(), reset_buf=reset_buf.numpy(),
             successes=successes.numpy(), cons_in=cons.numpy(), reward=rew.numpy(), resets=resets.numpy(), cons_out=cons_out.numpy(),
             max_episode_length=np.array(125.0))


def main():
    os.makedirs(OUT, exist_ok=True)
    ins, tv = import_insert()
    scene = G.load_scene()
    g = torch.Generator().manual_seed(24)
    i2_pre_physics(ins, g, scene)
    i3_observations(ins, tv, g, scene)
    i5_reward(ins, g)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("I"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
