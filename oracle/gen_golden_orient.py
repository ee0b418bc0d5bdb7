#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of the reference's BlockAssemblyOrient per-step tensor code
(BASELINE.json configs[2]), produced by RUNNING THE REFERENCE'S OWN functions in this container.

  python oracle/gen_golden_orient.py      # needs /root/reference; writes tests/golden/O*.npz

Same harness as oracle/gen_golden.py (MagicMock stubs for isaacgym & co, isaacgym.torch_utils shim, unbound methods on a
SimpleNamespace that holds exactly the attributes each method reads).  OR = dexteroushandenvs/tasks/block_assembly/
allegro_hand_block_assembly_orient.py.  Fixtures are data only (synthetic inputs under fixed seeds + the reference's outputs):
  O2 pre_physics.npz     OR:1697-1778  object-tracking IK targets, orientation_error OR:1922-1925, progress phases 10 / 76
  O3 observations.npz    OR:1087-1243 + compute_real_observations OR:1308-1326 + asymmetric states OR:1244-1306, 3 calls
  O5 reward.npz          OR:1843-1907
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


def import_orient():
    import importlib
    gs, vr, tv = G.import_reference()
    orr = importlib.import_module("tasks.block_assembly.allegro_hand_block_assembly_orient")
    orr.gymtorch.unwrap_tensor = lambda t: t
    return orr, tv


def o2_pre_physics(orr, g, scene):
    lo, hi = G.limits(scene)
    idx = G.actor_indices()
    out = {}
    for phase, prog in enumerate([10, 76]):
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
        p = torch.full((N,), prog, dtype=torch.long)
        p[::3] = torch.tensor([5, 75, 76, 100, 149, 3][:len(p[::3])])
        ns.progress_buf = p
        ns.segmentation_target_init_pos = torch.rand(N, 3, generator=g) * 0.3 + torch.tensor([0.1, 0.1, 0.65])
        ns.rigid_body_states = torch.randn(N, NB, 13, generator=g) * 0.3
        ns.rigid_body_states[:, :, 3:7] = G.rand_quat(g, N * NB).view(N, NB, 4)
        ns.rigid_body_states[:, 7, 2] += 0.9
        ns.hand_base_rigid_body_index = 7
        ns.root_state_tensor = torch.randn(N * A, 13, generator=g) * 0.2
        ns.root_state_tensor[idx["seg"], 0:3] += torch.tensor([0.25, 0.19, 0.7])
        ns.lego_segmentation_indices = idx["seg"]
        ns.jacobian_tensor = torch.randn(N, 23, 6, 23, generator=g)
        ns.target_euler = torch.tensor([0.0, 3.1415, 1.571]).repeat(N, 1)          # OR:477
        actions = torch.rand(N, 23, generator=g) * 2 - 1
        pre = dict(actions=actions, q=q, prev_targets=ns.prev_targets.clone(), progress=p.clone(),
                   init_pos=ns.segmentation_target_init_pos.clone(), hand_pos=ns.rigid_body_states[:, 7, 0:3].clone(),
                   hand_rot=ns.rigid_body_states[:, 7, 3:7].clone(), target_pos=ns.root_state_tensor[idx["seg"], 0:3].clone(),
                   J=ns.jacobian_tensor[:, 6, :, :7].clone(), target_euler=ns.target_euler.clone())
        orr.BlockAssemblyOrient.pre_physics_step(ns, actions)
        name, args = ns.gym.calls[-1]
        assert name == "set_dof_position_target_tensor"
        for k, v in pre.items():
            out["p%d_%s" % (phase, k)] = v.numpy()
        out["p%d_cur_targets" % phase] = ns.cur_targets.numpy().copy()
        out["p%d_prev_targets_out" % phase] = ns.prev_targets.numpy().copy()
        out["p%d_sim_targets" % phase] = args[0].numpy().copy()
        out["p%d_bc_act_label" % phase] = ns.bc_act_label.numpy().copy()
    # orientation_error on its own (OR:1922-1925)
    d, c = G.rand_quat(g, 64), G.rand_quat(g, 64)
    out["oe_desired"], out["oe_current"], out["oe_err"] = d.numpy(), c.numpy(), orr.orientation_error(d, c).numpy()
    out["lower"], out["upper"] = lo.numpy(), hi.numpy()
    np.savez(os.path.join(OUT, "O2_pre_physics.npz"), **out)


def o3_observations(orr, tv, g, scene):
    lo, hi = G.limits(scene)
    ns = G.make_obs_namespace(orr, tv, g, scene)      # same attribute set; only the method bindings differ
    ns.one_frame_num_obs, ns.one_frame_num_states = 62, 188
    ns.obs_buf = torch.zeros(N, 186)
    ns.obs_buf_stack_frames = [torch.zeros(N, 62) for _ in range(3)]
    ns.use_temporal_tvalue = False
    ns.compute_real_observations = lambda: orr.BlockAssemblyOrient.compute_real_observations(ns)
    ns.compute_contact_asymmetric_observations = lambda: orr.BlockAssemblyOrient.compute_contact_asymmetric_observations(ns)
    out = {}
    for k, v in ns.t_value.state_dict().items():
        out["tv_" + k.replace(".", "_")] = v.numpy().copy()
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
        rb = torch.randn(N, NB, 13, generator=g) * 0.3
        rb[:, :, 3:7] = G.rand_quat(g, N * NB).view(N, NB, 4)
        rb[:, :, 0:3] += torch.tensor([0.2, 0.2, 0.8])
        dof = torch.stack([lo + (hi - lo) * torch.rand(N, 23, generator=g), torch.randn(N, 23, generator=g) * 3.0], dim=-1)
        contact = torch.randn(N, NB * 3, generator=g) * 0.08
        actions = torch.rand(N, 23, generator=g) * 2 - 1
        ns.root_state_tensor, ns.rigid_body_states, ns.contact_tensor, ns.actions = root, rb, contact, actions
        ns.arm_hand_dof_pos, ns.arm_hand_dof_vel = dof[..., 0], dof[..., 1]
        ns.progress_buf = torch.randint(0, 150, (N,), generator=g)
        with torch.no_grad():
            orr.BlockAssemblyOrient.compute_observations(ns)
        pre = "c%d_" % c
        out[pre + "root"], out[pre + "rb"], out[pre + "dof"] = root.numpy(), rb.numpy(), dof.numpy()
        out[pre + "contact"], out[pre + "actions"] = contact.numpy(), actions.numpy()
        out[pre + "obs_buf"], out[pre + "states_buf"] = ns.obs_buf.numpy().copy(), ns.states_buf.numpy().copy()
        out[pre + "tvalue"] = ns.tvalue.detach().numpy().copy()                      # thresholded at 0.99 (OR:1203)
        out[pre + "tvalue_confident"] = torch.sigmoid(ns.tvalue_predict_confident)[:, 1].detach().numpy().copy()
        out[pre + "z_align"] = ns.lego_z_align_reward.numpy().copy()
        out[pre + "finger_dist"] = ns.arm_hand_finger_dist.numpy().copy()
    out["lower"], out["upper"] = lo.numpy(), hi.numpy()
    np.savez_compressed(os.path.join(OUT, "O3_observations.npz"), **out)


def o5_reward(orr, g):
    M = 64
    z = torch.tensor([0, 0, 1], dtype=torch.float).repeat(M, 1)
    x = torch.tensor([1, 0, 0], dtype=torch.float).repeat(M, 1)
    tgt = torch.rand(M, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.7])
    rot = G.rand_quat(g, M)
    rot[:8] = torch.tensor([0.0, 0.0, 0.0, 1.0])                  # a few bricks exactly face up
    spread = torch.cat([torch.full((M // 2,), 0.03), torch.full((M // 2,), 0.25)])[:, None]
    tips = [tgt + torch.randn(M, 3, generator=g) * spread for _ in range(4)]
    progress = torch.tensor(([3, 74, 148, 149, 150, 176, 200, 10] * (M // 8)), dtype=torch.long)
    reset_buf = torch.zeros(M, dtype=torch.long)
    reset_buf[5::16] = 1
    successes = torch.zeros(M)
    successes[::7] = 1.0
    cons = torch.tensor([0.37])
    angvel = torch.randn(M, 3, generator=g) * 20
    init_z = torch.rand(M, generator=g) * 2 - 1
    rew, resets, rgoal, prog, succ, cons_out = orr.compute_hand_reward(
        torch.tensor(1.0), torch.zeros(M), reset_buf, torch.zeros(M, dtype=torch.long), progress, successes, cons, 0,
        torch.rand(M, 6, generator=g), rot, torch.zeros(M, 3), rot, 150.0, torch.zeros(M, 3), rot, angvel,
        torch.zeros(M, 3), rot, tgt, torch.zeros(M, 3), torch.zeros(M), tips[0], tips[1], tips[2], tips[3],
        torch.zeros(M), tgt.clone(), -1.0, 1.0, 0.1, torch.zeros(M, 23), -0.0, 0.1, 250.0, 0.4, 0.0, 1, 0, 0.1, False,
        torch.zeros(M, 3), M, z, rot, x, rot, torch.zeros(M), init_z)
    np.savez(os.path.join(OUT, "O5_reward.npz"), target_pos=tgt.numpy(), target_rot=rot.numpy(),
             ff=tips[0].numpy(), rf=tips[1].numpy(), mf=tips[2].numpy(), th=tips[3].numpy(),
             progress=progress.numpy(), reset_buf=reset_buf.numpy(), successes=successes.numpy(), cons_in=cons.numpy(),
             reward=rew.numpy(), resets=resets.numpy(), cons_out=cons_out.numpy(), max_episode_length=np.array(150.0),
             fall_penalty=np.array(0.0), max_consecutive_successes=np.array(0))


def main():
    os.makedirs(OUT, exist_ok=True)
    orr, tv = import_orient()
    scene = G.load_scene()
    g = torch.Generator().manual_seed(23)
    o2_pre_physics(orr, g, scene)
    o3_observations(orr, tv, g, scene)
    o5_reward(orr, g)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("O"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
