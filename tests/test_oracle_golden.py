"""oracle/task_oracle.py (numpy restatement) vs golden vectors produced by the REFERENCE's own functions
(oracle/gen_golden.py).  This is what pins the oracle for SURVEY.md §8(a) rows T2-T10."""
import os

import numpy as np
import pytest

from oracle import task_oracle as T

TOL = dict(rtol=2e-5, atol=2e-5)  # fp32 restatement vs torch fp32 (different op order)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_control_ik(golden_dir):
    g = load(golden_dir, "F1_control_ik.npz")
    u = T.control_ik(g["J"], g["dpose"][..., 0])
    np.testing.assert_allclose(u, g["u"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("phase", [0, 1, 2, 3])
def test_pre_physics_targets(golden_dir, phase):
    g = load(golden_dir, "F2_pre_physics.npz")
    p = "p%d_" % phase
    cur = T.pre_physics_targets(g[p + "actions"], g[p + "q"], g[p + "prev_targets"], g[p + "progress"],
                                g[p + "init_pos"], g[p + "hand_pos"], g[p + "J"], g["lower"], g["upper"])
    np.testing.assert_allclose(cur, g[p + "cur_targets"], rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(g[p + "cur_targets"], g[p + "prev_targets_out"])
    np.testing.assert_array_equal(g[p + "cur_targets"], g[p + "sim_targets"])


def test_observations_four_calls(golden_dir, scene):
    g = load(golden_dir, "F3_observations.npz")
    w = {k[3:]: g[k] for k in g.files if k.startswith("tv_")}
    n = g["c0_rb"].shape[0]
    obs_buf = np.zeros((n, 396), dtype=np.float32)
    st_buf = np.zeros((n, 564), dtype=np.float32)
    for c in range(4):
        p = "c%d_" % c
        root_env = g[p + "root"].reshape(n, 142, 13)
        o, s, d = T.compute_observation_frames(
            root_env, g[p + "rb"], g[p + "dof"], g[p + "contact"].reshape(n, -1, 3), g[p + "actions"],
            g["seg_index_in_env"], g["init_pos"], g["init_rot"], g["lower"], g["upper"],
            np.array(scene.camera_offset_quat, dtype=np.float32), np.array(scene.camera_offset_pos, dtype=np.float32),
            scene.fingertip_bodies, tv_weights=w)
        obs_buf = T.stack_frames(obs_buf, o)
        st_buf = T.stack_frames(st_buf, s)
        np.testing.assert_allclose(obs_buf, g[p + "obs_buf"], **TOL)
        np.testing.assert_allclose(st_buf, g[p + "states_buf"], **TOL)
        for k in ["contacts", "finger_dist", "z_align", "hand_view_pos", "hand_view_rot", "cam_target_pos",
                  "cam_target_rot", "ff_pos", "rf_pos", "mf_pos", "th_pos"]:
            np.testing.assert_allclose(d[k], g[p + k], **TOL, err_msg=k)
        np.testing.assert_allclose(d["tvalue"], g[p + "tvalue"], rtol=1e-4, atol=1e-5)


def test_reward(golden_dir):
    g = load(golden_dir, "F5_reward.npz")
    rew, resets, cons, _ = T.compute_hand_reward(g["target_pos"], g["init_pos"], g["ff"], g["rf"], g["mf"], g["th"],
                                                 g["progress"], g["reset_buf"], g["cons_in"])
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(resets, g["resets"])
    np.testing.assert_allclose(cons, g["cons_out"], rtol=1e-6)
    assert resets.sum() > 0 and (resets == 0).sum() > 0 and (rew > 1).sum() > 0  # all branches were hit


def test_tvalue(golden_dir):
    g = load(golden_dir, "F7_tvalue.npz")
    w = {k[3:]: g[k] for k in g.files if k.startswith("tv_")}
    y, p = T.tvalue_forward(g["x"], w)
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p, g["tvalue"], rtol=1e-4, atol=1e-5)


def test_reset_idx(golden_dir, scene):
    g = load(golden_dir, "F8_reset_idx.npz")
    n = g["progress_before"].shape[0]
    mask = np.zeros(n, dtype=bool)
    mask[g["env_ids"]] = True
    choice = np.zeros(n, dtype=np.int64)
    choice[g["env_ids"]] = g["pile_choice"]
    root_b = g["root_before"].reshape(n, 142, 13)
    out = T.reset_idx(root_b, g["dof_before"].reshape(n, 23, 2), g["prev_before"], g["cur_before"],
                      g["progress_before"], mask.astype(np.int64), np.zeros((n, 3), np.float32),
                      np.zeros((n, 4), np.float32), mask, g["piles"], choice, g["seg_index_in_env"],
                      g["lower"], g["upper"],
                      # the fixture's object_init_state / goal_init_state (gen_golden.f8_reset_idx)
                      np.array([0, 0, -10.78, 0, 0, 0, 1] + [0] * 6, dtype=np.float32),
                      np.array([-0.2, -0.06, -10.78 - 10.12], dtype=np.float32))
    root, dof, prev, cur, prog, rb, ipos, irot = out
    ra = g["root_after"].reshape(n, 142, 13)
    np.testing.assert_allclose(root[:, 9:141], ra[:, 9:141], rtol=0, atol=0)   # bricks: exact copy of the saved pile
    np.testing.assert_allclose(root[:, 1], ra[:, 1], atol=1e-6)               # vestigial object
    np.testing.assert_allclose(root[:, 2, 0:3], ra[:, 2, 0:3], atol=1e-5)     # vestigial goal position
    np.testing.assert_allclose(dof, g["dof_after"].reshape(n, 23, 2), atol=1e-6)
    np.testing.assert_allclose(prev, g["prev_after"], atol=1e-6)
    np.testing.assert_allclose(cur, g["cur_after"], atol=1e-6)
    np.testing.assert_array_equal(prog, g["progress_after"])
    np.testing.assert_array_equal(rb, g["reset_after"])
    np.testing.assert_allclose(ipos[mask], g["init_pos_after"][mask], atol=0)
    np.testing.assert_allclose(irot[mask], g["init_rot_after"][mask], atol=0)
    assert (g["successes_after"][mask] == 0).all() and (g["meta_rew_after"][mask] == 0).all()


def test_vectask_clamps(golden_dir):
    g = load(golden_dir, "F9_vectask.npz")
    np.testing.assert_array_equal(T.vectask_clamp_actions(g["actions"]), g["stepped_actions"])
    np.testing.assert_array_equal(T.vectask_clamp_obs(g["obs_buf"]), g["obs"])
    np.testing.assert_array_equal(T.vectask_clamp_obs(g["states_buf"]), g["states"])
    assert float(g["reset_actions_absmax"]) <= 0.01 + 1e-7     # reset() steps with 0.01*(1-2U) actions, VR:180
    assert int(g["info_agents"]) == 1 and int(g["num_agents"]) == 1


def test_seg_index_rule(golden_dir):
    g = load(golden_dir, "F3_observations.npz")
    for i, v in enumerate(g["seg_index_in_env"]):
        assert T.seg_index_for_env(i) == int(v)
