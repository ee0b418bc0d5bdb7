#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): generate tests/golden/*.npz by RUNNING THE
REFERENCE'S OWN pure-torch functions in this container.

  python oracle/gen_golden.py          # needs /root/reference; writes tests/golden/

The reference module
  dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_grasp_sim.py
is imported in place (sys.path; nothing is copied; bytecode writing disabled) with
MagicMock stubs for isaacgym/cv2/... and oracle/isaacgym_torch_utils_shim.py
registered as `isaacgym.torch_utils`.  Instance methods are driven *unbound* on a
types.SimpleNamespace holding exactly the attributes each method reads.  The saved
fixtures are data only: synthetic inputs (fixed seeds) and the reference's outputs.

Fixtures (reference file:line of what produced the expected values):
  F1 control_ik.npz          GS:1796-1804
  F2 pre_physics.npz         GS:1555-1638  (progress phases 10/76/101/126)
  F3 observations.npz        GS:1090-1218 + GS:1299-1332 + GS:1220-1280, 4 consecutive calls
  F5 reward.npz              GS:1706-1776
  F7 tvalue.npz              policy_sequencing/terminal_value_function.py:30-46
  F8 reset_idx.npz           GS:1361-1553
  F9 vectask.npz             tasks/hand_base/vec_task_rlgames.py:160-192
"""
import importlib
import importlib.util
import json
import os
import random
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference/dexteroushandenvs"
OUT = os.path.join(REPO, "tests", "golden")


def import_reference():
    for name in ["isaacgym", "isaacgym.gymapi", "isaacgym.gymtorch", "isaacgym.gymutil", "cv2", "pyquaternion",
                 "pytorch3d", "pytorch3d.transforms", "h5py", "gym", "gym.spaces", "torchvision",
                 "torchvision.models", "torchvision.transforms"]:
        sys.modules[name] = MagicMock()
    spec = importlib.util.spec_from_file_location("isaacgym.torch_utils",
                                                  os.path.join(HERE, "isaacgym_torch_utils_shim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    sys.modules["isaacgym.torch_utils"] = shim
    sys.modules["isaacgym"].torch_utils = shim
    sys.path.insert(0, REF)
    gs = importlib.import_module("tasks.block_assembly.allegro_hand_block_assembly_grasp_sim")
    gs.gymtorch.unwrap_tensor = lambda t: t

    class FakeQuat:  # stands in for gymapi.Quat at GS:1491 (values are overwritten at GS:1511 anyway)
        def __init__(self, x=0.0, y=0.0, z=0.0, w=1.0):
            self.x, self.y, self.z, self.w = x, y, z, w

        def from_euler_zyx(self, a, b, c):
            return FakeQuat(0.0, 0.0, 0.0, 1.0)

    gs.gymapi.Quat = FakeQuat
    vr = importlib.import_module("tasks.hand_base.vec_task_rlgames")
    tv = importlib.import_module("policy_sequencing.terminal_value_function")
    return gs, vr, tv


class FakeGym:
    """records set_* calls, ignores refresh_*/colour calls."""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def f(*a, **k):
            if name.startswith("set_") and "color" not in name:
                self.calls.append((name, [x.clone() if torch.is_tensor(x) else x for x in a[1:]]))
            return None
        return f


def rand_quat(g, n):
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def load_scene():
    with open(os.path.join(REPO, "seqdex_amd", "scene_data", "grasp_sim_scene.json")) as f:
        return json.load(f)


def limits(scene):
    b = scene["robot"]["bodies"]
    lo = torch.tensor([b[i + 1]["lower"] for i in range(23)], dtype=torch.float)
    hi = torch.tensor([b[i + 1]["upper"] for i in range(23)], dtype=torch.float)
    return lo, hi


N = 16          # envs in every fixture
A = 142         # actors per env (hand, object, goal, table, 5 bin boxes, 132 bricks, base plate)
NB = 165        # rigid bodies per env


def actor_indices():
    base = torch.arange(N, dtype=torch.long) * A
    seg_id = torch.tensor([(i % 8 if (i % 8) not in (3, 4, 7) else 0) for i in range(N)])  # GS:962-965
    lego = base[:, None] + 9 + torch.arange(132)[None, :]
    return dict(hand=base + 0, obj=base + 1, goal=base + 2, lego=lego, seg=base + 9 + seg_id, extra=base + 141)


def f1_control_ik(gs, g):
    J = torch.randn(N, 6, 7, generator=g)
    e = torch.randn(N, 6, 1, generator=g) * 0.3
    u = gs.control_ik(J, "cpu", e, N)
    np.savez(os.path.join(OUT, "F1_control_ik.npz"), J=J.numpy(), dpose=e.numpy(), u=u.numpy())


def f2_pre_physics(gs, g, scene):
    lo, hi = limits(scene)
    out = {}
    for phase, prog in enumerate([10, 76, 101, 126]):
        ns = types.SimpleNamespace()
        ns.gym, ns.sim, ns.device, ns.num_envs = FakeGym(), None, "cpu", N
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
        # mixed progress: half the envs at `prog`, the rest spread so masks differ per env
        p = torch.full((N,), prog, dtype=torch.long)
        p[::3] = torch.tensor([5, 75, 76, 100, 101, 125, 126, 149][:len(p[::3])])
        ns.progress_buf = p
        ns.segmentation_target_init_pos = torch.rand(N, 3, generator=g) * 0.3 + torch.tensor([0.1, 0.1, 0.65])
        ns.rigid_body_states = torch.randn(N, NB, 13, generator=g) * 0.3
        ns.rigid_body_states[:, 7, 2] += 0.9
        ns.hand_base_rigid_body_index = 7
        ns.jacobian_tensor = torch.randn(N, 23, 6, 23, generator=g)
        ns.arm_hand_insertion_prepare_dof_pos_list = [
            torch.tensor([-0.1560, -0.2140, -0.2795, -2.1806, -0.0681, 1.9730, 1.1735]),
            torch.tensor([-0.1800, -0.1604, -0.2770, -2.2674, -0.0533, 2.1049, 1.1696])]
        actions = torch.rand(N, 23, generator=g) * 2 - 1
        pre = dict(actions=actions, q=q, prev_targets=ns.prev_targets.clone(), progress=p.clone(),
                   init_pos=ns.segmentation_target_init_pos.clone(), hand_pos=ns.rigid_body_states[:, 7, 0:3].clone(),
                   J=ns.jacobian_tensor[:, 6, :, :7].clone())
        gs.BlockAssemblyGraspSim.pre_physics_step(ns, actions)
        name, args = ns.gym.calls[-1]
        assert name == "set_dof_position_target_tensor"
        for k, v in pre.items():
            out["p%d_%s" % (phase, k)] = v.numpy()
        out["p%d_cur_targets" % phase] = ns.cur_targets.numpy().copy()
        out["p%d_prev_targets_out" % phase] = ns.prev_targets.numpy().copy()
        out["p%d_sim_targets" % phase] = args[0].numpy().copy()
        out["p%d_bc_act_label" % phase] = ns.bc_act_label.numpy().copy()
    out["lower"], out["upper"] = lo.numpy(), hi.numpy()
    np.savez(os.path.join(OUT, "F2_pre_physics.npz"), **out)


def make_obs_namespace(gs, tv, g, scene):
    lo, hi = limits(scene)
    idx = actor_indices()
    ns = types.SimpleNamespace()
    ns.gym, ns.sim, ns.device, ns.num_envs = FakeGym(), None, "cpu", N
    ns.envs = [None]
    ns.hand_indices, ns.object_indices, ns.goal_object_indices = idx["hand"], idx["obj"], idx["goal"]
    ns.lego_segmentation_indices, ns.extra_object_indices = idx["seg"], idx["extra"]
    ns.hand_base_rigid_body_index = 7
    ns.mount_rigid_body_index = 7
    ns.fingertip_handles = torch.tensor(scene["robot"]["fingertip_bodies"], dtype=torch.long)
    ns.sensor_handle_indices = torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.int64)
    ns.camera_offset_quat = torch.tensor(scene["camera_offset_quat"], dtype=torch.float)
    ns.camera_offset_pos = torch.tensor(scene["camera_offset_pos"], dtype=torch.float)
    ns.z_unit_tensor = torch.tensor([0, 0, 1], dtype=torch.float).repeat(N, 1)
    ns.arm_hand_dof_lower_limits, ns.arm_hand_dof_upper_limits = lo, hi
    ns.vel_obs_scale = 0.2
    ns.obs_type = "partial_contact"
    ns.save_hdf5 = False
    ns.one_frame_num_obs, ns.one_frame_num_states = 132, 188
    ns.obs_buf = torch.zeros(N, 396)
    ns.states_buf = torch.zeros(N, 564)
    ns.obs_buf_stack_frames = [torch.zeros(N, 132) for _ in range(3)]
    ns.state_buf_stack_frames = [torch.zeros(N, 188) for _ in range(3)]
    ns.goal_states = torch.zeros(N, 13)
    ns.perturb_direction = torch.zeros(N, 6)
    ns.perturb_steps = torch.zeros(N)
    torch.manual_seed(1234)
    ns.t_value = tv.GraspInsertTValue(input_dim=4, output_dim=2)
    ns.compute_sim_observations = lambda: gs.BlockAssemblyGraspSim.compute_sim_observations(ns)
    ns.compute_contact_asymmetric_observations = \
        lambda: gs.BlockAssemblyGraspSim.compute_contact_asymmetric_observations(ns)
    return ns


def f3_observations(gs, tv, g, scene):
    lo, hi = limits(scene)
    ns = make_obs_namespace(gs, tv, g, scene)
    out = {}
    sd = ns.t_value.state_dict()
    for k, v in sd.items():
        out["tv_" + k.replace(".", "_")] = v.numpy().copy()
    ns.segmentation_target_init_pos = torch.rand(N, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.65])
    ns.segmentation_target_init_rot = rand_quat(g, N)
    out["init_pos"], out["init_rot"] = ns.segmentation_target_init_pos.numpy(), ns.segmentation_target_init_rot.numpy()
    out["seg_index_in_env"] = (ns.lego_segmentation_indices - torch.arange(N) * A).numpy()
    for c in range(4):
        root = torch.randn(N * A, 13, generator=g) * 0.2
        root[:, 3:7] = rand_quat(g, N * A)
        root[ns.hand_indices, 0:3] = torch.tensor(scene["robot"]["base_pos"])
        root[ns.hand_indices, 3:7] = torch.tensor([0.0, 0, 0, 1])
        root[ns.lego_segmentation_indices, 0:3] += torch.tensor([0.2, 0.2, 0.7])
        rb = torch.randn(N, NB, 13, generator=g) * 0.3
        rb[:, :, 3:7] = rand_quat(g, N * NB).view(N, NB, 4)
        rb[:, :, 0:3] += torch.tensor([0.2, 0.2, 0.8])
        if c == 3:  # provoke the +-5 clamp seen through VecTask and large values
            rb[:, 11, 7:13] *= 40.0
        dof = torch.stack([lo + (hi - lo) * torch.rand(N, 23, generator=g),
                           torch.randn(N, 23, generator=g) * 3.0], dim=-1)
        contact = torch.randn(N, NB * 3, generator=g) * 0.08
        actions = torch.rand(N, 23, generator=g) * 2 - 1
        ns.root_state_tensor, ns.rigid_body_states, ns.contact_tensor, ns.actions = root, rb, contact, actions
        ns.arm_hand_dof_pos, ns.arm_hand_dof_vel = dof[..., 0], dof[..., 1]
        ns.progress_buf = torch.randint(0, 150, (N,), generator=g)
        with torch.no_grad():
            gs.BlockAssemblyGraspSim.compute_observations(ns)
        pre = "c%d_" % c
        out[pre + "root"], out[pre + "rb"], out[pre + "dof"] = root.numpy(), rb.numpy(), dof.numpy()
        out[pre + "contact"], out[pre + "actions"] = contact.numpy(), actions.numpy()
        out[pre + "obs_buf"], out[pre + "states_buf"] = ns.obs_buf.numpy().copy(), ns.states_buf.numpy().copy()
        out[pre + "contacts"] = ns.contacts.numpy().copy()
        out[pre + "finger_dist"] = ns.arm_hand_finger_dist.numpy().copy()
        out[pre + "tvalue"] = ns.tvalue.detach().numpy().copy()
        out[pre + "z_align"] = ns.lego_z_align_reward.numpy().copy()
        out[pre + "hand_view_pos"] = ns.hand_base_view_hand_pos.numpy().copy()
        out[pre + "hand_view_rot"] = ns.hand_base_view_hand_rot.numpy().copy()
        out[pre + "cam_target_pos"] = ns.camera_view_segmentation_target_pos.numpy().copy()
        out[pre + "cam_target_rot"] = ns.camera_view_segmentation_target_rot.numpy().copy()
        out[pre + "ff_pos"], out[pre + "rf_pos"] = ns.arm_hand_ff_pos.numpy().copy(), ns.arm_hand_rf_pos.numpy().copy()
        out[pre + "mf_pos"], out[pre + "th_pos"] = ns.arm_hand_mf_pos.numpy().copy(), ns.arm_hand_th_pos.numpy().copy()
    out["lower"], out["upper"] = lo.numpy(), hi.numpy()
    np.savez_compressed(os.path.join(OUT, "F3_observations.npz"), **out)


def f5_reward(gs, g):
    M = 64
    z = torch.tensor([0, 0, 1], dtype=torch.float).repeat(M, 1)
    x = torch.tensor([1, 0, 0], dtype=torch.float).repeat(M, 1)
    tgt = torch.rand(M, 3, generator=g) * 0.2 + torch.tensor([0.1, 0.1, 0.7])
    init = tgt.clone()
    init[:, 2] -= torch.rand(M, generator=g) * 0.3 - 0.05  # lifts in [-0.05, 0.25]
    spread = torch.cat([torch.full((M // 2,), 0.03), torch.full((M // 2,), 0.25)])[:, None]
    tips = [tgt + torch.randn(M, 3, generator=g) * spread for _ in range(4)]
    progress = torch.tensor(([3, 74, 75, 76, 100, 148, 149, 150] * (M // 8)), dtype=torch.long)
    reset_buf = torch.zeros(M, dtype=torch.long)
    reset_buf[5::16] = 1
    successes = torch.zeros(M)
    cons = torch.tensor([0.37])
    rot = rand_quat(g, M)
    rew, resets, rgoal, prog, succ, cons_out = gs.compute_hand_reward(
        torch.tensor(1.0), torch.zeros(M), reset_buf, torch.zeros(M, dtype=torch.long), progress, successes, cons, 0,
        torch.zeros(M, 6), rot, torch.zeros(M, 3), rot, 150.0, torch.zeros(M, 3), rot, torch.zeros(M, 3),
        torch.zeros(M, 3), rot, tgt, torch.zeros(M, 3), torch.zeros(M), tips[0], tips[1], tips[2], tips[3],
        torch.zeros(M), init, -1.0, 1.0, 0.1, torch.zeros(M, 23), -0.0, 0.1, 250.0, 0.4, 0.0, 1, 0, 0.1, False,
        torch.zeros(M, 3), M, z, rot, x, rot, torch.zeros(M))
    np.savez(os.path.join(OUT, "F5_reward.npz"), target_pos=tgt.numpy(), init_pos=init.numpy(),
             ff=tips[0].numpy(), rf=tips[1].numpy(), mf=tips[2].numpy(), th=tips[3].numpy(),
             progress=progress.numpy(), reset_buf=reset_buf.numpy(), cons_in=cons.numpy(),
             reward=rew.numpy(), resets=resets.numpy(), cons_out=cons_out.numpy())


def f7_tvalue(tv, g):
    torch.manual_seed(77)
    net = tv.GraspInsertTValue(input_dim=4, output_dim=2)
    x = rand_quat(g, 32)
    with torch.no_grad():
        y = net(x)
        p = torch.sigmoid(y)[:, 1]
    out = {"x": x.numpy(), "y": y.numpy(), "tvalue": p.numpy()}
    for k, v in net.state_dict().items():
        out["tv_" + k.replace(".", "_")] = v.numpy().copy()
    np.savez(os.path.join(OUT, "F7_tvalue.npz"), **out)


def f8_reset_idx(gs, g, scene):
    lo, hi = limits(scene)
    idx = actor_indices()
    K = 6  # synthetic saved piles per brick type (the reference's file holds >=5000; GS:1507 samples range(0,5000))
    ns = types.SimpleNamespace()
    ns.gym, ns.sim, ns.device, ns.num_envs = FakeGym(), None, "cpu", N
    ns.record_completion_time = False
    ns.save_hdf5 = False
    ns.randomize = False
    ns.total_steps = 0
    ns.up_axis_idx = 2
    ns.reset_position_noise = 0.0
    ns.num_arm_hand_dofs = 23
    ns.z_unit_tensor = torch.tensor([0, 0, 1], dtype=torch.float).repeat(N, 1)
    ns.x_unit_tensor = torch.tensor([1, 0, 0], dtype=torch.float).repeat(N, 1)
    ns.y_unit_tensor = torch.tensor([0, 1, 0], dtype=torch.float).repeat(N, 1)
    ns.segmentation_target_rot = rand_quat(g, N)
    ns.hand_indices, ns.object_indices, ns.goal_object_indices = idx["hand"], idx["obj"], idx["goal"]
    ns.lego_indices, ns.lego_segmentation_indices = idx["lego"], idx["seg"]
    ns.root_state_tensor = torch.randn(N * A, 13, generator=g)
    root_before = ns.root_state_tensor.clone()
    ns.rigid_body_states = torch.randn(N, NB, 13, generator=g)
    ns.base_pos = torch.zeros(N, 3)
    ns.perturb_steps = torch.zeros(N)
    ns.perturb_direction = torch.zeros(N, 6)
    ns.rb_forces = torch.ones(N, NB, 3)
    ns.object_init_state = torch.zeros(N, 13)
    ns.object_init_state[:, 0:3] = torch.tensor(scene["vestigial_object_pos"])
    ns.object_init_state[:, 6] = 1.0
    ns.object_pose_for_open_loop = torch.zeros(N, 7)
    ns.goal_states = ns.object_init_state.clone()
    ns.goal_init_state = ns.object_init_state.clone()
    ns.goal_displacement_tensor = torch.tensor([-0.2, -0.06, -10.12])
    ns.reset_goal_buf = torch.ones(N, dtype=torch.long)
    ns.lego_init_states = torch.randn(N, 132, 13, generator=g)
    ns.force_prob_range = torch.tensor([0.001, 0.1])
    ns.random_force_prob = torch.zeros(N)
    ns.arm_hand_prepare_dof_poses = torch.zeros(N, 23)
    ns.end_effector_rotation = torch.zeros(N, 4)
    prep = torch.tensor([0.0, -0.49826458111314524, -0.01990020486871322, -2.4732269941140346, -0.01307073642274261,
                         2.00396583422025, 1.5480939705504309] + [0.0] * 16)
    prep[7:] = 0.5 * (torch.zeros(16) + 1.0) * (hi[7:] - lo[7:]) + lo[7:]  # scale(0, lo, hi), GS:271-272
    ns.arm_hand_prepare_dof_pos_list = [prep]
    ns.end_effector_rot_list = [torch.tensor([0, 0.707, 0, 0.707])]
    ns.dof_state = torch.randn(N * 23, 2, generator=g)
    ahs = ns.dof_state.view(N, -1, 2)[:, :23]
    ns.arm_hand_dof_state = ahs
    ns.arm_hand_dof_pos, ns.arm_hand_dof_vel = ahs[..., 0], ahs[..., 1]
    ns.arm_hand_dof_default_vel = torch.zeros(23)
    ns.arm_hand_dof_lower_limits, ns.arm_hand_dof_upper_limits = lo, hi
    ns.prev_targets = torch.randn(N, 23, generator=g)
    ns.cur_targets = torch.randn(N, 23, generator=g)
    ns.segmentation_target_init_pos = torch.zeros(N, 3)
    ns.segmentation_target_init_rot = torch.zeros(N, 4)
    ns.progress_buf = torch.randint(0, 150, (N,), generator=g)
    ns.reset_buf = torch.zeros(N, dtype=torch.long)
    env_ids = torch.tensor([0, 2, 3, 7, 8, 13, 15])
    ns.reset_buf[env_ids] = 1
    ns.successes = torch.ones(N)
    ns.meta_rew_buf = torch.ones(N)
    piles = [torch.randn(5000 if False else K, 132, 13, generator=g) for _ in range(8)]
    # the reference indexes saved[object_i][random.sample(range(0,5000),1)]; give it K piles and patch the range
    ns.saved_searching_ternimal_states_list = piles
    ns.reset_target_pose = lambda ids, apply_reset=False: gs.BlockAssemblyGraspSim.reset_target_pose(ns, ids, apply_reset)
    # drive python `random` so that sample(range(0,5000),1) lands inside [0,K): wrap random.sample
    chosen = []
    real_sample = random.sample

    def fake_sample(pop, k):
        if isinstance(pop, range) and len(pop) == 5000:
            v = [int(torch.randint(0, K, (1,), generator=g))]
            chosen.append(v[0])
            return v
        return real_sample(pop, k)

    gs.random.sample = fake_sample
    random.seed(5)
    torch.manual_seed(5)
    progress_before, dof_before = ns.progress_buf.clone(), ns.dof_state.clone()
    prev_before, cur_before = ns.prev_targets.clone(), ns.cur_targets.clone()
    gs.BlockAssemblyGraspSim.reset_idx(ns, env_ids, ns.reset_goal_buf.nonzero(as_tuple=False).squeeze(-1))
    gs.random.sample = real_sample
    calls = {n: a for n, a in ns.gym.calls}
    np.savez_compressed(
        os.path.join(OUT, "F8_reset_idx.npz"), env_ids=env_ids.numpy(), pile_choice=np.array(chosen),
        piles=torch.stack(piles).numpy(), root_before=root_before.numpy(), root_after=ns.root_state_tensor.numpy(),
        dof_before=dof_before.numpy(), dof_after=ns.dof_state.numpy(), prev_before=prev_before.numpy(),
        cur_before=cur_before.numpy(), prev_after=ns.prev_targets.numpy(), cur_after=ns.cur_targets.numpy(),
        progress_before=progress_before.numpy(), progress_after=ns.progress_buf.numpy(),
        reset_after=ns.reset_buf.numpy(), successes_after=ns.successes.numpy(), meta_rew_after=ns.meta_rew_buf.numpy(),
        init_pos_after=ns.segmentation_target_init_pos.numpy(), init_rot_after=ns.segmentation_target_init_rot.numpy(),
        seg_index_in_env=(ns.lego_segmentation_indices - torch.arange(N) * A).numpy(),
        root_set_indices=calls["set_actor_root_state_tensor_indexed"][1].numpy(),
        dof_set_indices=calls["set_dof_state_tensor_indexed"][1].numpy(), lower=lo.numpy(), upper=hi.numpy())


def f9_vectask(vr, g):
    task = types.SimpleNamespace()
    task.num_envs, task.num_obs, task.num_states, task.num_actions, task.device = N, 396, 564, 23, "cpu"
    task.obs_buf = torch.randn(N, 396, generator=g) * 4
    task.states_buf = torch.randn(N, 564, generator=g) * 4
    task.rew_buf = torch.randn(N, generator=g)
    task.reset_buf = torch.randint(0, 2, (N,), generator=g)
    task.extras = {}
    seen = []
    task.step = lambda a: seen.append(a.clone())
    vr.spaces.Box = lambda lo, hi: (np.asarray(lo).copy(), np.asarray(hi).copy())
    if not hasattr(np, "Inf"):
        np.Inf = np.inf  # the reference predates NumPy 2 (VR:27-28)
    env = vr.RLgamesVecTaskPython(task, "cpu")
    actions = torch.randn(N, 23, generator=g) * 2
    obs_dict, rew, reset, extras = env.step(actions)
    torch.manual_seed(9)
    od2 = env.reset()
    np.savez(os.path.join(OUT, "F9_vectask.npz"), obs_buf=task.obs_buf.numpy(), states_buf=task.states_buf.numpy(),
             actions=actions.numpy(), stepped_actions=seen[0].numpy(), obs=obs_dict["obs"].numpy(),
             states=obs_dict["states"].numpy(), rew=rew.numpy(), reset=reset.numpy(),
             reset_actions_absmax=np.array(float(seen[1].abs().max())), reset_obs=od2["obs"].numpy(),
             info_agents=np.array(env.get_env_info()["agents"]), num_agents=np.array(env.get_number_of_agents))


def main():
    os.makedirs(OUT, exist_ok=True)
    gs, vr, tv = import_reference()
    scene = load_scene()
    g = torch.Generator().manual_seed(22)
    f1_control_ik(gs, g)
    f2_pre_physics(gs, g, scene)
    f3_observations(gs, tv, g, scene)
    f5_reward(gs, g)
    f7_tvalue(tv, g)
    f8_reset_idx(gs, g, scene)
    f9_vectask(vr, g)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
