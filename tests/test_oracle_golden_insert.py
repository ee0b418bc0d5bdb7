"""CPU: the numpy restatement of the BlockAssemblyInsertSim per-step tensor code (oracle/task_oracle.py, IS:*) against the golden
vectors that oracle/gen_golden_insert.py captured from the reference's own functions (tests/golden/I*.npz).
Second policy of the grasp -> insert chain (BASELINE.json configs[2] family / SURVEY.md section 8(f)); the HIP side of this task
is not built - this pins the oracle that a later round's kernels will be checked against."""
import os

import numpy as np

from oracle import task_oracle as T

TOL = dict(rtol=2e-5, atol=2e-5)


def test_insert_pre_physics_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "I2_pre_physics.npz"))
    cur, rot_err = T.insert_pre_physics_targets(g["actions"], g["q"], g["prev_targets"], g["hand_rot"], g["J"], g["lower"],
                                                g["upper"], g["target_euler"][0])
    np.testing.assert_allclose(rot_err, g["rot_err"], **TOL)
    for k in ("cur_targets", "sim_targets", "prev_targets_out"):
        np.testing.assert_allclose(cur, g[k], rtol=1e-4, atol=1e-4)
    assert (np.abs(cur - g["lower"]) < 1e-6).any() and (np.abs(cur - g["upper"]) < 1e-6).any()   # the limit clamp is exercised


def test_insert_offset_sets_cover_every_env():
    s = T.insert_offset_sets(48)
    assert (s["xn"] ^ s["x1"]).all() and s["x1"].sum() == 6
    assert sorted(set(s["height"].tolist())) == [0, 1, 2]


def test_insert_observations_golden(golden_dir, scene):
    g = np.load(os.path.join(golden_dir, "I3_observations.npz"))
    n = g["c0_rb"].shape[0]
    for c in range(3):
        p = "c%d_" % c
        root = g[p + "root"].reshape(n, 142, 13)
        # the reference needs the progress counter only for states[141]; recover it from the golden states themselves
        progress = np.rint(g[p + "states_buf"][:, 141] * 125.0)
        o, s, d = T.insert_observation_frames(root, g[p + "rb"], g[p + "dof"], g[p + "actions"], g["seg_index_in_env"],
                                              g["init_pos"], g["lower"], g["upper"], np.array(scene.camera_offset_quat, np.float32),
                                              np.array(scene.camera_offset_pos, np.float32), scene.fingertip_bodies, progress)
        np.testing.assert_allclose(d["extra_target_pos"], g[p + "extra_target_pos"], **TOL)
        np.testing.assert_allclose(d["symmetry_rot"], g[p + "symmetry_extra_target_rot"], **TOL)
        np.testing.assert_allclose(d["finger_dist"], g[p + "finger_dist"], **TOL)
        np.testing.assert_allclose(o, g[p + "obs_buf"], **TOL)
        np.testing.assert_allclose(s, g[p + "states_buf"], **TOL)
        assert not o[:, 16:23].any() and not o[:, 60].any()                                       # unused columns stay zero
        # advanced indexing returns copies: the offsets never reach the simulator's root tensor
        np.testing.assert_array_equal(g[p + "root_after"], g[p + "root"])


def test_insert_reward_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "I5_reward.npz"))
    rew, resets, cons, rot_dist = T.insert_hand_reward(
        g["target_pos"], g["target_rot"], g["extra_pos"], g["extra_rot"], g["symmetry_rot"], g["rot_err"], g["ff"], g["rf"],
        g["mf"], g["th"], g["progress"], g["reset_buf"], g["cons_in"], g["successes"], float(g["max_episode_length"]))
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(resets, g["resets"])
    np.testing.assert_allclose(cons, g["cons_out"], **TOL)
    assert (g["reward"] > 1.0).any() and 0 < resets.sum() < resets.size                          # bonus and both reset outcomes occur
