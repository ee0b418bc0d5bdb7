"""CPU: the numpy restatement of the BlockAssemblyOrient per-step tensor code (oracle/task_oracle.py, OR:*) against the golden
vectors that oracle/gen_golden_orient.py captured from the reference's own functions (tests/golden/O*.npz).
configs[2] of BASELINE.json / SURVEY.md section 8(f) rank 1 - the HIP side of this task is not built yet."""
import os

import numpy as np

from oracle import task_oracle as T

TOL = dict(rtol=2e-5, atol=2e-5)


def test_orientation_error_and_euler_quaternion(golden_dir):
    g = np.load(os.path.join(golden_dir, "O2_pre_physics.npz"))
    np.testing.assert_allclose(T.orientation_error(g["oe_desired"], g["oe_current"]), g["oe_err"], **TOL)
    q = T.quat_from_euler_xyz(*[np.array([v], np.float32) for v in T.ORIENT_TARGET_EULER])
    assert abs(np.linalg.norm(q) - 1) < 1e-6


def test_orient_pre_physics_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "O2_pre_physics.npz"))
    for ph in range(2):
        p = "p%d_" % ph
        cur = T.orient_pre_physics_targets(g[p + "actions"], g[p + "q"], g[p + "prev_targets"], g[p + "progress"], g[p + "init_pos"],
                                           g[p + "hand_pos"], g[p + "hand_rot"], g[p + "target_pos"], g[p + "J"], g["lower"],
                                           g["upper"], g[p + "target_euler"][0])
        np.testing.assert_allclose(cur, g[p + "cur_targets"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cur, g[p + "sim_targets"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cur, g[p + "prev_targets_out"], rtol=1e-4, atol=1e-4)
        # bc_act_label is taken BEFORE the limit clamp (OR:1756-1758): clipped to [-1, 1] it is the unscaled target
        np.testing.assert_allclose(T.unscale(cur, g["lower"], g["upper"]), np.clip(g[p + "bc_act_label"], -1, 1), rtol=1e-4, atol=3e-4)
        hold = g[p + "progress"] > 75
        assert hold.any() and (~hold).any()


def test_orient_observations_golden(golden_dir, scene):
    g = np.load(os.path.join(golden_dir, "O3_observations.npz"))
    tv = {k[3:]: g[k] for k in g.files if k.startswith("tv_")}
    n = g["c0_rb"].shape[0]
    st_prev = np.zeros((n, 564), np.float32)
    for c in range(3):
        p = "c%d_" % c
        root = g[p + "root"].reshape(n, 142, 13)
        o132, s188, d = T.compute_observation_frames(
            root, g[p + "rb"], g[p + "dof"], g[p + "contact"].reshape(n, 165, 3), g[p + "actions"], g["seg_index_in_env"],
            g["init_pos"], g["init_rot"], g["lower"], g["upper"], np.array(scene.camera_offset_quat, np.float32),
            np.array(scene.camera_offset_pos, np.float32), scene.fingertip_bodies, tv_weights=tv)
        obs = np.zeros((n, 186), np.float32)
        obs[:, :62] = T.orient_obs_frame(g[p + "dof"], g[p + "actions"], g["lower"], g["upper"])
        np.testing.assert_allclose(obs, g[p + "obs_buf"], **TOL)                                   # columns 62.. stay zero (no stacking)
        st_prev = T.stack_frames(st_prev, s188)
        np.testing.assert_allclose(st_prev, g[p + "states_buf"], **TOL)                            # asymmetric states as in GraspSim
        np.testing.assert_allclose(d["tvalue"], g[p + "tvalue_confident"], rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(T.orient_tvalue_gate(g[p + "tvalue_confident"]), g[p + "tvalue"])
        np.testing.assert_allclose(d["z_align"], g[p + "z_align"], **TOL)
        np.testing.assert_allclose(d["finger_dist"], g[p + "finger_dist"], **TOL)


def test_orient_reward_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "O5_reward.npz"))
    rew, resets, cons, z = T.orient_hand_reward(g["target_pos"], g["target_rot"], g["ff"], g["rf"], g["mf"], g["th"], g["progress"],
                                                g["reset_buf"], g["cons_in"], g["successes"],
                                                max_episode_length=float(g["max_episode_length"]),
                                                max_consecutive_successes=int(g["max_consecutive_successes"]),
                                                fall_penalty=float(g["fall_penalty"]))
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(resets, g["resets"])
    np.testing.assert_allclose(cons, g["cons_out"], rtol=1e-6)
    assert (g["resets"] == 1).any() and (g["resets"] == 0).any() and (g["progress"] > 175).any()
