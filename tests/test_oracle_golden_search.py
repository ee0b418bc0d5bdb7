"""CPU: numpy restatement of the pure per-step tensor code of BlockAssemblySearch (oracle/task_oracle.py, SE:*) against the golden vectors
that oracle/gen_golden_search.py captured from the reference's own functions (tests/golden/S*.npz).  The Search TASK is not built
(it needs a segmentation rasteriser); this pins the oracle for the round that builds it."""
import os

import numpy as np

from oracle import task_oracle as T

TOL = dict(rtol=2e-5, atol=2e-5)


def test_search_pre_physics_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "S2_pre_physics.npz"))
    np.testing.assert_allclose(g["euler"], T.SEARCH_EULER)
    cur = T.search_pre_physics_targets(g["actions"], g["q"], g["prev_targets"], g["hand_pos"], g["hand_rot"], g["target_pos"], g["J"],
                                       g["lower"], g["upper"])
    np.testing.assert_allclose(cur, g["cur_targets"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cur, g["sim_targets"], rtol=1e-4, atol=1e-4)


def test_search_observations_and_pixel_statistics_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "S3_observations.npz"))
    cx, cy, num = T.segmentation_pixel_stats(g["seg"], g["ids"])
    np.testing.assert_array_equal(num, g["point_num"].reshape(-1))
    np.testing.assert_array_equal(cx, g["center_x"].reshape(-1))
    np.testing.assert_array_equal(cy, g["center_y"].reshape(-1))
    assert num[3] == 0 and cx[3] == 0 and (num > 0).sum() >= 10
    obs = np.zeros_like(g["obs_buf"])
    obs[:, :62] = T.search_obs_frame(g["dof"], g["actions"], g["lower"], g["upper"])
    np.testing.assert_allclose(obs, g["obs_buf"], **TOL)
    a = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    a.update(center_x=g["center_x"], center_y=g["center_y"], point_num=g["point_num"])
    s = T.search_state_frame(g["dof"], g["actions"], g["lower"], g["upper"], a)
    np.testing.assert_allclose(s, g["states_buf"][:, :188], **TOL)
    assert not g["states_buf"][:, 188:].any() and not s[:, 95].any() and not s[:, 175:].any()


def test_search_reward_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "S5_reward.npz"))
    rew, resets, cons, up = T.search_hand_reward(g["target_pos"], g["init_pos"], g["ff"], g["rf"], g["mf"], g["th"], g["progress"],
                                                 g["reset_buf"], g["cons_in"], g["successes"], g["arm_contacts"], g["actions"],
                                                 float(g["max_episode_length"]))
    np.testing.assert_allclose(rew, g["reward"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(resets, g["resets"])
    np.testing.assert_allclose(cons, g["cons_out"], **TOL)
    np.testing.assert_allclose(g["emergence_after"], g["emergence_in"] * up / 10, rtol=1e-4, atol=1e-4)   # scaled in place, never used
    pix, em = T.search_emergence_reward(g["em_seg"], g["em_ids"], g["em_last"])
    np.testing.assert_array_equal(pix, g["em_pixel"])
    np.testing.assert_allclose(em, g["em_reward"], **TOL)
    np.testing.assert_array_equal(T.search_heap_movement(g["heap_pos"]), g["heap_penalty"])
    assert g["heap_penalty"].max() > 0


def test_retri_tvalue_forward_golden(golden_dir):
    """RetriGraspTValue(650, 2) of the reference (terminal_value_function.py:12-28) with the formula-defined parameters: the numpy
    restatement reproduces the module's output and the sigmoid the task takes of it (SE:1133-1134)"""
    f = np.load(os.path.join(golden_dir, "S7_retri_tvalue.npz"))
    out, tv = T.retri_tvalue_forward(f["x"], T.retri_tvalue_formula_weights())
    np.testing.assert_allclose(out, f["out"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(tv, f["tvalue"], rtol=2e-5, atol=2e-5)
    assert sum(w.size for w in T.retri_tvalue_formula_weights().values()) == 1257346


def test_search_tvalue_buffer_shift():
    rng = np.random.default_rng(0)
    buf = rng.normal(size=(4, 650)).astype(np.float32)
    obs = rng.normal(size=(4, 62)).astype(np.float32)
    q = rng.normal(size=(4, 4)).astype(np.float32)
    nb = T.search_tvalue_buffer_update(buf, obs, q, np.full(4, 64.0, np.float32), np.full(4, 32.0, np.float32), np.full(4, 250.0, np.float32))
    np.testing.assert_array_equal(nb[:, :585], buf[:, 65:])
    np.testing.assert_array_equal(nb[:, 585:611], obs[:, :26]); np.testing.assert_array_equal(nb[:, 611:615], q)
    np.testing.assert_array_equal(nb[:, 615:647], obs[:, 30:62])
    np.testing.assert_allclose(nb[:, 647:650], np.tile([0.5, 0.25, 2.5], (4, 1)))
