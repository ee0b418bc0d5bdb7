"""TEST INFRASTRUCTURE ONLY (oracle/): numpy CPU restatement of the reference-owned
per-step tensor code of BlockAssemblyGraspSim.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product path is seqdex_amd/csrc (HIP).

Pinning: every function here is checked against golden vectors produced by running
the reference's own functions (oracle/gen_golden.py -> tests/golden/F*.npz) in
tests/test_oracle_golden.py.  Abbreviation: GS = reference file
dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_grasp_sim.py,
VR = tasks/hand_base/vec_task_rlgames.py, TV = policy_sequencing/terminal_value_function.py.

All arithmetic is float32 (the reference runs torch.float32); quaternions are xyzw.
"""
import numpy as np

F = np.float32


# ------------------------------------------------------------------ T10: isaacgym.torch_utils restatement
def quat_mul(a, b):
    """Hamilton product, xyzw (torch_utils.quat_mul; used GS:1179,1182,1263)."""
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    return np.stack([x, y, z, w], axis=-1).astype(F)


def quat_conjugate(a):
    return np.concatenate([-a[..., :3], a[..., 3:]], axis=-1).astype(F)


def quat_apply(q, v):
    """v + w t + u x t, t = 2 u x v  (torch_utils.quat_apply; used GS:1154-1157,1195)."""
    u = q[..., :3]
    t = np.cross(u, v) * F(2)
    return (v + q[..., 3:] * t + np.cross(u, t)).astype(F)


def tf_inverse(q, t):
    qi = quat_conjugate(q)
    return qi, -quat_apply(qi, t)


def tf_combine(q1, t1, q2, t2):
    return quat_mul(q1, q2), quat_apply(q1, t2) + t1


def scale(x, lo, hi):
    return (F(0.5) * (x + F(1.0)) * (hi - lo) + lo).astype(F)


def unscale(x, lo, hi):
    return ((F(2.0) * x - hi - lo) / (hi - lo)).astype(F)


def quat_from_angle_axis(angle, axis):
    th = (angle / F(2))[..., None]
    ax = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    q = np.concatenate([ax * np.sin(th), np.cos(th)], axis=-1)
    return (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(F)


# ------------------------------------------------------------------ T3: control_ik, GS:1796-1804
def control_ik(J, dpose, damping=0.05):
    """u = J^T (J J^T + damping^2 I)^-1 dpose.  J [N,6,7], dpose [N,6] -> [N,7]."""
    J = J.astype(F)
    A = J @ np.swapaxes(J, 1, 2) + (F(damping) ** 2) * np.eye(6, dtype=F)
    y = np.linalg.solve(A.astype(np.float64), dpose.astype(np.float64)[..., None])
    return (np.swapaxes(J, 1, 2).astype(np.float64) @ y)[..., 0].astype(F)


# ------------------------------------------------------------------ T2: pre_physics_step, GS:1570-1638
INSERT_POSE_A = np.array([-0.1560, -0.2140, -0.2795, -2.1806, -0.0681, 1.9730, 1.1735], dtype=F)  # GS:278
INSERT_POSE_B = np.array([-0.1800, -0.1604, -0.2770, -2.2674, -0.0533, 2.1049, 1.1696], dtype=F)  # GS:281


def pre_physics_targets(actions, q, prev_targets, progress, init_pos, hand_pos, J, lower, upper,
                        act_moving_average=1.0):
    """action -> joint position targets (after any reset has been applied).

    actions [N,23] (already clamped by VR:166), q [N,23], prev_targets [N,23], progress int [N],
    init_pos [N,3] = segmentation_target_init_pos, hand_pos [N,3] = link-7 position, J [N,6,7].
    Returns cur_targets [N,23] (== new prev_targets, GS:1636).
    """
    a = actions.astype(F)
    cur = np.zeros_like(prev_targets, dtype=F)
    cur[:, 7:23] = scale(a[:, 7:23], lower[7:23], upper[7:23])                                   # GS:1585-1587
    cur[:, 7:23] = F(act_moving_average) * cur[:, 7:23] + F(1.0 - act_moving_average) * prev_targets[:, 7:23]
    m0, m1, m2 = progress > 75, progress > 100, progress > 125                                   # GS:1590-1592
    pos_err = a[:, 0:3] * F(0.64)                                                                # GS:1594
    rot_err = a[:, 3:6] * F(0.2)                                                                 # GS:1595
    lift = F(0.2) + F(0.22) + (init_pos - hand_pos)[:, 2]                                        # GS:1596
    pos_err[m0, 2] = lift[m0]
    pos_err[m0, 0] = 0
    pos_err[m0, 1] = 0
    dpose = np.concatenate([pos_err, rot_err], axis=-1)
    cur[:, :7] = q[:, :7] + control_ik(J, dpose)                                                 # GS:1600-1602
    cur[m1, :7] = INSERT_POSE_A                                                                  # GS:1604
    cur[m2, :7] = INSERT_POSE_B                                                                  # GS:1605
    cur[m0, 7:23] = prev_targets[m0, 7:23]                                                       # GS:1606
    return np.maximum(np.minimum(cur, upper), lower).astype(F)                                   # GS:1633-1635


# ------------------------------------------------------------------ TV: GraspInsertTValue, TV:30-46
def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(F)


def tvalue_forward(x, w):
    """4 -> 256 -> 128 -> 64 -> 2 with ELU on every layer including the output; w = dict of
    linear{1,2,3}_{weight,bias}, output_layer_{weight,bias}.  Returns (logits[N,2], sigmoid(logits)[:,1])
    (GS:1200-1201)."""
    h = x.astype(F)
    for name in ["linear1", "linear2", "linear3", "output_layer"]:
        h = elu(h @ w[name + "_weight"].T + w[name + "_bias"])
    return h, (F(1) / (F(1) + np.exp(-h[:, 1]))).astype(F)


# ------------------------------------------------------------------ T5-T7: compute_observations, GS:1090-1280,1299-1332
FT_OFFSET = np.array([0, 0, 0.04], dtype=F)  # GS:1154-1157


def compute_observation_frames(root_env, rb, dof, contact, actions, seg_idx, init_pos, init_rot, lower, upper,
                               cam_q, cam_p, fingertips, tv_weights=None, hand_body=7):
    """One un-stacked frame of obs (132) and states (188) plus the derived quantities.

    root_env [N,142,13] actor root states of each env; rb [N,165,13]; dof [N,23,2]; contact [N,165,3];
    actions [N,23]; seg_idx [N] target-brick actor index within the env (9 + brick id);
    fingertips = body indices of link_3.0, link_7.0, link_11.0, link_15.0 (ff, mf, rf, th; GS:183-186,1130-1147).
    """
    N = rb.shape[0]
    ar = np.arange(N)
    base_pos, base_rot = root_env[:, 0, 0:3], root_env[:, 0, 3:7]                                # GS:1097-1098
    tgt = root_env[ar, seg_idx]                                                                  # GS:1116-1121
    tpos, trot, tlin, tang = tgt[:, 0:3], tgt[:, 3:7], tgt[:, 7:10], tgt[:, 10:13]
    hb = rb[:, hand_body]                                                                        # GS:1110-1114
    hpos, hrot = hb[:, 0:3], hb[:, 3:7]
    ff, mf, rf, th = (rb[:, fingertips[i]] for i in range(4))                                    # GS:1130-1152
    tip = lambda s: (s[:, 0:3] + quat_apply(s[:, 3:7], np.broadcast_to(FT_OFFSET, (N, 3)))).astype(F)
    ffp, mfp, rfp, thp = tip(ff), tip(mf), tip(rf), tip(th)                                      # GS:1154-1157
    cn = np.linalg.norm(contact[:, 1:7, :], axis=-1)                                             # GS:1159-1161
    contacts = np.where(cn >= 0.1, F(1), F(0))                                                   # GS:1162
    nrm = lambda v: np.linalg.norm(v, axis=-1).astype(F)
    finger_dist = nrm(tpos - ffp) + nrm(tpos - mfp) + nrm(tpos - rfp) + nrm(tpos - thp)          # GS:1164-1165
    qbi, pbi = tf_inverse(base_rot, base_pos)                                                    # GS:1172
    hv_rot, hv_pos = tf_combine(qbi, pbi, hrot, hpos)                                            # GS:1173
    qc, pc = tf_combine(hrot, hpos, np.broadcast_to(cam_q, (N, 4)), np.broadcast_to(cam_p, (N, 3)))  # GS:1176-1179
    qci, pci = tf_inverse(qc, pc)                                                                # GS:1180
    ct_rot, ct_pos = tf_combine(qci, pci, trot, tpos)                                            # GS:1182
    axis1 = quat_apply(trot, np.broadcast_to(np.array([0, 0, 1], dtype=F), (N, 3)))              # GS:1195-1198
    dot1 = axis1[:, 2]
    z_align = (np.sign(dot1) * dot1 ** 2).astype(F)
    tvalue = tvalue_forward(ct_rot, tv_weights)[1] if tv_weights is not None else np.zeros(N, dtype=F)

    q, qd = dof[..., 0], dof[..., 1]
    o = np.zeros((N, 132), dtype=F)                                                              # GS:1299-1326
    o[:, 0:16] = unscale(q[:, 7:23], lower[7:23], upper[7:23])
    o[:, 16:19], o[:, 19:23] = hv_pos, hv_rot
    o[:, 23:26], o[:, 26:30] = ct_pos, ct_rot
    o[:, 30:46] = F(0.2) * qd[:, 7:23]
    o[:, 46:59], o[:, 59:72], o[:, 72:85], o[:, 85:98] = ff, rf, mf, th
    o[:, 98:111] = tgt
    o[:, 111:114], o[:, 114:118] = hpos, hrot
    o[:, 118:121], o[:, 121:125] = init_pos, init_rot
    o[:, 125:128] = tpos - init_pos
    o[:, 128:131] = hpos - tpos

    s = np.zeros((N, 188), dtype=F)                                                              # GS:1220-1276
    s[:, 0:23] = unscale(q, lower, upper)
    s[:, 23:46] = F(0.2) * qd
    s[:, 46:49], s[:, 49:52], s[:, 52:55], s[:, 55:58] = ffp, rfp, mfp, thp
    s[:, 58:81] = actions
    s[:, 81:88] = hb[:, 0:7]
    s[:, 88:95] = tgt[:, 0:7]
    s[:, 95:98], s[:, 98:101] = hb[:, 7:10], hb[:, 10:13]
    for k, st in enumerate([ff, mf, rf, th]):                                                    # GS:1239-1253
        s[:, 101 + 10 * k:105 + 10 * k] = st[:, 3:7]
        s[:, 105 + 10 * k:108 + 10 * k] = st[:, 7:10]
        s[:, 108 + 10 * k:111 + 10 * k] = st[:, 10:13]
    s[:, 142:145], s[:, 145:148] = tlin, tang
    s[:, 148:151] = init_pos
    s[:, 151:154] = tpos - init_pos
    s[:, 154:157] = hpos - tpos
    s[:, 157:161] = quat_mul(hrot, quat_conjugate(trot))
    s[:, 161:164], s[:, 164:167] = tpos - ffp, tpos - rfp
    s[:, 167:170], s[:, 170:173] = tpos - mfp, tpos - thp
    s[:, 173] = finger_dist
    s[:, 174:177], s[:, 177:181] = ct_pos, ct_rot
    s[:, 181:184], s[:, 184:188] = ct_pos, ct_rot
    derived = dict(contacts=contacts, finger_dist=finger_dist, tvalue=tvalue, z_align=z_align, hand_view_pos=hv_pos,
                   hand_view_rot=hv_rot, cam_target_pos=ct_pos, cam_target_rot=ct_rot, ff_pos=ffp, rf_pos=rfp,
                   mf_pos=mfp, th_pos=thp, target_pos=tpos)
    return o, s, derived


def stack_frames(buf_prev, frame):
    """3-frame stacking, GS:1330-1332 / GS:1278-1280: new = [frame, prev[0:w], prev[w:2w]]."""
    w = frame.shape[1]
    return np.concatenate([frame, buf_prev[:, 0:w], buf_prev[:, w:2 * w]], axis=1).astype(F)


# ------------------------------------------------------------------ T8: compute_hand_reward, GS:1706-1776
def compute_hand_reward(target_pos, init_pos, ff, rf, mf, th, progress, reset_buf, cons_successes,
                        max_episode_length=150.0, av_factor=0.1, successes=None):
    nrm = lambda v: np.linalg.norm(v.astype(F), axis=-1).astype(F)
    d = nrm(target_pos - ff) + nrm(target_pos - mf) + nrm(target_pos - rf) + F(3) * nrm(target_pos - th)  # GS:1740
    resets = np.where(d <= -1, 1, reset_buf)                                                     # GS:1727
    timed_out = progress >= max_episode_length - 1                                               # GS:1729
    resets = np.where(timed_out, 1, resets)
    dist_rew = (np.exp(F(-2) * np.clip(d - F(0.5), 0, None)) * F(0.1)).astype(F)                 # GS:1742
    up = np.clip(target_pos[:, 2] - init_pos[:, 2], 0, 0.2).astype(F) * F(100)                   # GS:1744
    up = np.minimum(np.where(d < 0.5, up, F(0)), F(20))                                          # GS:1745
    reward = (dist_rew + up).astype(F)                                                           # GS:1751
    resets = np.where((progress >= 75) & (d >= 0.6), 1, resets)                                  # GS:1754-1755
    succ = np.zeros_like(reward) if successes is None else successes
    num_resets = resets.sum()                                                                    # GS:1771-1774
    fin = (succ * resets.astype(F)).sum()
    cons = np.where(num_resets > 0, F(av_factor) * fin / max(num_resets, 1) + F(1.0 - av_factor) * cons_successes,
                    cons_successes).astype(F)
    return reward, resets.astype(np.int64), cons, d


# ------------------------------------------------------------------ T4: reset_idx, GS:1361-1553
FINGER_RESET_UNSCALED = np.array([0, 0, -1, 0.5, 1, 0, -1, 0.5, 0, 0, -1, 0.5, 0, 0, -1, 0.5], dtype=F)   # GS:1531
ARM_PREPARE_POSE0 = np.array([0.0, -0.49826458111314524, -0.01990020486871322, -2.4732269941140346,
                              -0.01307073642274261, 2.00396583422025, 1.5480939705504309], dtype=F)        # GS:267


def reset_idx(root_env, dof, prev_targets, cur_targets, progress, reset_buf, init_pos, init_rot, env_mask,
              piles, pile_choice, seg_idx, lower, upper, object_init, goal_pos):
    """Masked re-initialisation.  root_env [N,142,13]; piles [8,K,132,13]; pile_choice [N] int (per env, only
    entries under env_mask are used; the reference draws it with python random, GS:1510).
    Returns updated copies.  Rows the reference leaves to RNG-only bookkeeping (perturb_*, random_force_prob,
    goal rotation GS:1338-1347) are not modelled: nothing on the hot path reads them."""
    root_env, dof = root_env.copy(), dof.copy()
    prev_targets, cur_targets = prev_targets.copy(), cur_targets.copy()
    progress, reset_buf = progress.copy(), reset_buf.copy()
    init_pos, init_rot = init_pos.copy(), init_rot.copy()
    hand_pose = np.concatenate([ARM_PREPARE_POSE0, scale(FINGER_RESET_UNSCALED, lower[7:23], upper[7:23])])
    prep_full = np.concatenate([ARM_PREPARE_POSE0, scale(np.zeros(16, dtype=F), lower[7:23], upper[7:23])])
    for e in np.nonzero(env_mask)[0]:
        root_env[e, 1] = object_init                                                             # GS:1475-1482
        root_env[e, 2, 0:3] = goal_pos                                                           # GS:1348-1350
        root_env[e, 2, 7:13] = 0
        root_env[e, 9:141] = piles[e % 8, pile_choice[e]]                                        # GS:1508-1511
        root_env[e, 9:141, 7:13] = 0                                                             # GS:1513
        dof[e, :, 0] = hand_pose                                                                 # GS:1526,1531
        dof[e, :, 1] = 0                                                                         # GS:1529
        prev_targets[e] = hand_pose                                                              # GS:1527,1533
        cur_targets[e] = hand_pose                                                               # GS:1528,1535
        init_pos[e] = root_env[e, seg_idx[e], 0:3]                                               # GS:1547
        init_rot[e] = root_env[e, seg_idx[e], 3:7]                                               # GS:1548
        progress[e] = 0                                                                          # GS:1550
        reset_buf[e] = 0                                                                         # GS:1551
    del prep_full
    return root_env, dof, prev_targets, cur_targets, progress, reset_buf, init_pos, init_rot


# ------------------------------------------------------------------ T9: VecTask clamps, VR:165-192
def vectask_clamp_actions(a, clip=1.0):
    return np.clip(a, -clip, clip).astype(F)


def vectask_clamp_obs(o, clip=5.0):
    return np.clip(o, -clip, clip).astype(F)


def seg_index_for_env(i):
    """target brick = brick (i % 8) with {3,4,7} -> 0 (GS:962-965,974-975); actor index inside env = 9 + brick."""
    b = i % 8
    return 9 + (0 if b in (3, 4, 7) else b)


# =================================================================== BlockAssemblyOrient (configs[2])
# OR = dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_orient.py.  The scene, the intermediate quantities of
# compute_observations (OR:1087-1200 == GS:1090-1198) and the asymmetric state frame (OR:1244-1306 == GS:1220-1280) are shared
# with GraspSim; what differs is restated here and pinned by tests/golden/O*.npz (oracle/gen_golden_orient.py).
ORIENT_TARGET_EULER = np.array([0.0, 3.1415, 1.571], dtype=F)                                    # OR:477


def quat_from_euler_xyz(roll, pitch, yaw):
    """isaacgym.torch_utils.quat_from_euler_xyz (package absent; published half-angle product form): xyzw."""
    cy, sy, cr, sr = np.cos(yaw * F(0.5)), np.sin(yaw * F(0.5)), np.cos(roll * F(0.5)), np.sin(roll * F(0.5))
    cp, sp = np.cos(pitch * F(0.5)), np.sin(pitch * F(0.5))
    return np.stack([cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
                     cy * cr * cp + sy * sr * sp], axis=-1).astype(F)


def orientation_error(desired, current):
    """OR:1922-1925: vector part of desired * conj(current), sign-flipped into the w >= 0 hemisphere."""
    q_r = quat_mul(desired, quat_conjugate(current))
    return (q_r[:, 0:3] * np.sign(q_r[:, 3])[:, None]).astype(F)


def orient_pre_physics_targets(actions, q, prev_targets, progress, init_pos, hand_pos, hand_rot, target_pos, J, lower, upper,
                               target_euler=ORIENT_TARGET_EULER, act_moving_average=1.0):
    """OR:1720-1778 (after any reset): fingers from the action, arm by damped-least-squares IK that TRACKS the target brick
    (hand base held 0.22 above and 0.18 behind it, fixed wrist orientation); after step 75 the hand lifts towards
    z_init + 0.39 and the fingers hold their previous targets."""
    a = actions.astype(F)
    N = a.shape[0]
    cur = np.zeros_like(prev_targets, dtype=F)
    cur[:, 7:23] = scale(a[:, 7:23], lower[7:23], upper[7:23])                                   # OR:1726-1728
    cur[:, 7:23] = F(act_moving_average) * cur[:, 7:23] + F(1.0 - act_moving_average) * prev_targets[:, 7:23]
    m0 = progress > 75                                                                           # OR:1731
    pos_err = (target_pos - hand_pos).astype(F)                                                  # OR:1734
    pos_err[:, 2] += F(0.22)                                                                     # OR:1735
    pos_err[:, 0] -= F(0.18)                                                                     # OR:1736
    lift = (init_pos[:, 2] - hand_pos[:, 2] + F(0.15) + F(0.24)).astype(F)                       # OR:1737
    pos_err[m0, 2] = lift[m0]
    te = np.broadcast_to(np.asarray(target_euler, dtype=F), (N, 3))
    rot_err = orientation_error(quat_from_euler_xyz(te[:, 0], te[:, 1], te[:, 2]), hand_rot)     # OR:1740-1741
    cur[:, :7] = q[:, :7] + control_ik(J, np.concatenate([pos_err, rot_err], axis=-1))           # OR:1743-1745
    cur[m0, 7:23] = prev_targets[m0, 7:23]                                                       # OR:1746
    return np.maximum(np.minimum(cur, upper), lower).astype(F)                                   # OR:1773-1776


def orient_obs_frame(dof, actions, lower, upper):
    """compute_real_observations OR:1308-1326: 62 numbers - finger joint positions (unscaled), 14 unused zeros, the action
    minus the unscaled finger positions, the finger action.  NOT stacked: columns 62..185 of obs_buf stay zero in the reference."""
    q = dof[..., 0]
    o = np.zeros((q.shape[0], 62), dtype=F)
    u = unscale(q[:, 7:23], lower[7:23], upper[7:23])
    o[:, 0:16] = u
    o[:, 30:46] = actions[:, 7:23] - u
    o[:, 46:62] = actions[:, 7:23]
    return o


def orient_tvalue_gate(tvalue):
    """OR:1203: the T-value is thresholded at 0.99 before anything uses it."""
    return np.where(tvalue > F(0.99), F(1), F(0)).astype(F)


def orient_hand_reward(target_pos, target_rot, ff, rf, mf, th, progress, reset_buf, cons_successes, successes,
                       max_episode_length=150.0, av_factor=0.1, max_consecutive_successes=0, fall_penalty=0.0):
    """compute_hand_reward OR:1843-1907: exp(-5 (1 - (z_align + 1)/2) - 5 max(d - 0.4, 0)) with d the thumb-weighted fingertip
    distance (dropped after step 175); reset only on time-out."""
    nrm = lambda v: np.linalg.norm(v.astype(F), axis=-1).astype(F)
    N = target_pos.shape[0]
    d = nrm(target_pos - ff) + nrm(target_pos - mf) + nrm(target_pos - rf) + F(3) * nrm(target_pos - th)   # OR:1853-1854
    dot1 = quat_apply(target_rot, np.broadcast_to(np.array([0, 0, 1], dtype=F), (N, 3)))[:, 2]             # OR:1856-1859
    z_align = (np.sign(dot1) * dot1 ** 2).astype(F)
    resets = np.where(d <= -1, 1, reset_buf)                                                               # OR:1866
    timed_out = progress >= max_episode_length - 1                                                         # OR:1868-1869
    resets = np.where(timed_out, 1, resets)
    d_rew = np.clip(d - F(0.4), 0, None).astype(F)                                                         # OR:1878
    d_rew = np.where(progress > 175, F(0), d_rew)                                                          # OR:1879
    z_rew = (F(1) - (z_align + F(1)) / F(2)).astype(F)                                                     # OR:1884
    reward = np.exp(-(F(5) * z_rew + F(5) * d_rew)).astype(F)                                              # OR:1886
    if max_consecutive_successes > 0:
        reward = np.where(timed_out, reward + F(0.5 * fall_penalty), reward)                               # OR:1900-1901
    num_resets = resets.sum()                                                                              # OR:1903-1906
    fin = (successes * resets.astype(F)).sum()
    cons = np.where(num_resets > 0, F(av_factor) * fin / max(num_resets, 1) + F(1.0 - av_factor) * cons_successes,
                    cons_successes).astype(F)
    return reward, resets.astype(np.int64), cons, z_align


# ================================================================== BlockAssemblyInsertSim (second policy of the chain)
# IS = dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_insert_sim.py.  Pinned by tests/golden/I*.npz
# (oracle/gen_golden_insert.py).  The HIP side of this task is not built: oracle groundwork only.
INSERT_POS_SCALE = 0.64                                                                          # IS:1537


def insert_offset_sets(n):
    """IS:779-812 (use_unseen False): envs whose insertion target is shifted along the base plate's y by one stud pitch (1xn
    bricks), by one stud in x and y (the 1x1 brick, i % 8 == 5), and the plate-height class i % 3."""
    i = np.arange(n)
    return dict(xn=i % 8 != 5, x1=i % 8 == 5, height=i % 3)


def insert_extra_target(extra_pos, extra_rot):
    """IS:1121-1132,1165: the insertion site = root pose of the base plate actor shifted, in the plate's own frame, by
    0.0375 (1 + i % 3) in z, 0.015 in y (1xn) or 0.015 in x and y (1x1); plus the 180-degree-about-z symmetric orientation."""
    n = extra_pos.shape[0]
    s = insert_offset_sets(n)
    ax = lambda v: quat_apply(extra_rot, np.broadcast_to(np.array(v, dtype=F), (n, 3)))
    p = extra_pos.astype(F).copy()
    p = p + ax([0, 0, 1]) * (F(0.0375) * (s["height"] + 1).astype(F))[:, None]
    p = p + ax([0, 1, 0]) * np.where(s["xn"], F(0.015), F(0))[:, None]
    p = p + ax([1, 0, 0]) * np.where(s["x1"], F(0.015), F(0))[:, None]
    p = p + ax([0, 1, 0]) * np.where(s["x1"], F(0.015), F(0))[:, None]
    sym = quat_mul(extra_rot, np.broadcast_to(np.array([0, 0, 1, 0], dtype=F), (n, 4)))
    return p.astype(F), sym


def insert_pre_physics_targets(actions, q, prev_targets, hand_rot, J, lower, upper, target_euler, act_moving_average=1.0):
    """IS:1526-1572: fingers from a[7:23]; the arm moves the hand base by a[0:3] * 0.64 with the wrist orientation servoed to
    target_euler through the damped-least-squares IK.  Returns (targets, rot_err); rot_err feeds the reward's reset rule."""
    a = actions.astype(F)
    n = a.shape[0]
    cur = np.zeros_like(prev_targets, dtype=F)
    cur[:, 7:23] = scale(a[:, 7:23], lower[7:23], upper[7:23])
    cur[:, 7:23] = F(act_moving_average) * cur[:, 7:23] + F(1.0 - act_moving_average) * prev_targets[:, 7:23]
    pos_err = a[:, 0:3] * F(INSERT_POS_SCALE)
    te = np.broadcast_to(np.asarray(target_euler, dtype=F), (n, 3))
    rot_err = orientation_error(quat_from_euler_xyz(te[:, 0], te[:, 1], te[:, 2]), hand_rot)
    cur[:, :7] = q[:, :7] + control_ik(J, np.concatenate([pos_err, rot_err], axis=-1))
    return np.maximum(np.minimum(cur, upper), lower).astype(F), rot_err


def insert_observation_frames(root_env, rb, dof, actions, seg_idx, init_pos, lower, upper, cam_q, cam_p, fingertips, progress,
                              extra_actor=141, hand_body=7, max_episode_length=125.0):
    """compute_contact_observations IS:1280-1298 (75 numbers, NOT stacked) and the asymmetric state IS:1220-1278 (188, not
    stacked): poses are expressed relative to the insertion site instead of the camera."""
    n = rb.shape[0]
    ar = np.arange(n)
    tgt = root_env[ar, seg_idx]
    tpos, trot = tgt[:, 0:3], tgt[:, 3:7]
    ext = root_env[:, extra_actor]
    epos, sym = insert_extra_target(ext[:, 0:3], ext[:, 3:7])
    erot = ext[:, 3:7]
    hb = rb[:, hand_body]
    hpos, hrot = hb[:, 0:3], hb[:, 3:7]
    ff, mf, rf, th = (rb[:, fingertips[i]] for i in range(4))
    tip = lambda s: (s[:, 0:3] + quat_apply(s[:, 3:7], np.broadcast_to(FT_OFFSET, (n, 3)))).astype(F)
    ffp, mfp, rfp, thp = tip(ff), tip(mf), tip(rf), tip(th)
    nrm = lambda v: np.linalg.norm(v, axis=-1).astype(F)
    finger_dist = nrm(tpos - ffp) + nrm(tpos - mfp) + nrm(tpos - rfp) + nrm(tpos - thp)          # IS:1180-1181
    qc, pc = tf_combine(hrot, hpos, np.broadcast_to(cam_q, (n, 4)), np.broadcast_to(cam_p, (n, 3)))
    qci, pci = tf_inverse(qc, pc)
    ct_rot, ct_pos = tf_combine(qci, pci, trot, tpos)
    q, qd = dof[..., 0], dof[..., 1]

    o = np.zeros((n, 75), dtype=F)
    o[:, 0:16] = unscale(q[:, 7:23], lower[7:23], upper[7:23])
    o[:, 23:46] = actions
    o[:, 46:49] = hpos - epos
    o[:, 49:53] = quat_mul(hrot, quat_conjugate(erot))
    o[:, 53:56] = hpos - tpos
    o[:, 56:60] = quat_mul(hrot, quat_conjugate(trot))
    o[:, 61:64], o[:, 64:68] = epos, erot
    o[:, 68:71] = tpos - epos
    o[:, 71:75] = quat_mul(trot, quat_conjugate(erot))

    s = np.zeros((n, 188), dtype=F)
    s[:, 0:23] = unscale(q, lower, upper)
    s[:, 23:46] = F(0.2) * qd
    s[:, 46:49], s[:, 49:52], s[:, 52:55], s[:, 55:58] = ffp, rfp, mfp, thp
    s[:, 58:81] = actions
    s[:, 81:88] = hb[:, 0:7]
    s[:, 88:95] = tgt[:, 0:7]
    s[:, 95:98], s[:, 98:101] = hb[:, 7:10], hb[:, 10:13]
    for k, st in enumerate([ff, mf, rf, th]):
        s[:, 101 + 10 * k:105 + 10 * k] = st[:, 3:7]
        s[:, 105 + 10 * k:108 + 10 * k] = st[:, 7:10]
        s[:, 108 + 10 * k:111 + 10 * k] = st[:, 10:13]
    s[:, 141] = progress.astype(F) / F(max_episode_length)                                       # IS:1255
    s[:, 142:145], s[:, 145:148] = tgt[:, 7:10], tgt[:, 10:13]
    s[:, 148:151] = init_pos
    s[:, 151:154] = tpos - init_pos
    s[:, 154:157] = hpos - tpos
    s[:, 157:161] = quat_mul(hrot, quat_conjugate(trot))
    s[:, 161:164], s[:, 164:167] = tpos - ffp, tpos - rfp
    s[:, 167:170], s[:, 170:173] = tpos - mfp, tpos - thp
    s[:, 173] = finger_dist
    s[:, 174:177], s[:, 177:181] = ct_pos, ct_rot
    s[:, 181:184], s[:, 184:188] = epos, erot                                                    # IS:1277-1278
    return o, s, dict(extra_target_pos=epos, symmetry_rot=sym, finger_dist=finger_dist)


def insert_hand_reward(target_pos, target_rot, extra_pos, extra_rot, symmetry_rot, rot_err, ff, rf, mf, th, progress, reset_buf,
                       cons_successes, successes, max_episode_length=125.0, av_factor=0.1, max_consecutive_successes=0,
                       fall_penalty=0.0):
    """compute_hand_reward IS:1640-1695: exp(-rot_dist - 20 |brick - site|) (+1 once within 2 cm and 0.2 rad, the site's
    180-degree twin counting as aligned); reset when the hand lets go (thumb-weighted distance >= 0.6), when the wrist servo error
    grows (sum rot_err^2 >= 0.03) or on time-out."""
    nrm = lambda v: np.linalg.norm(v.astype(F), axis=-1).astype(F)
    d = nrm(target_pos - ff) + nrm(target_pos - mf) + nrm(target_pos - rf) + F(3) * nrm(target_pos - th)   # IS:1650-1651
    ang = lambda a, b: (F(2) * np.arcsin(np.minimum(nrm(quat_mul(a, quat_conjugate(b))[:, 0:3]), F(1)))).astype(F)
    rot_dist = np.minimum(ang(target_rot, extra_rot), ang(target_rot, symmetry_rot))             # IS:1656-1660
    gap = nrm(target_pos - extra_pos)
    insert_reward = np.exp(-rot_dist - F(20) * gap).astype(F)                                    # IS:1664
    bonus = np.where((gap < F(0.02)) & (rot_dist < F(0.2)), F(1), F(0))                          # IS:1666-1668
    resets = np.where(d >= F(0.6), 1, reset_buf)                                                 # IS:1673
    resets = np.where((rot_err.astype(F) ** 2).sum(-1) >= F(0.03), 1, resets)                    # IS:1675
    timed_out = progress >= max_episode_length - 1                                               # IS:1677-1678
    resets = np.where(timed_out, 1, resets)
    reward = (bonus + insert_reward).astype(F)                                                   # IS:1680
    if max_consecutive_successes > 0:
        reward = np.where(timed_out, reward + F(0.5 * fall_penalty), reward)
    num_resets = resets.sum()
    fin = (successes * resets.astype(F)).sum()
    cons = np.where(num_resets > 0, F(av_factor) * fin / max(num_resets, 1) + F(1.0 - av_factor) * cons_successes,
                    cons_successes).astype(F)
    return reward, resets.astype(np.int64), cons, rot_dist


# ================================================================== BlockAssemblySearch (first policy of the chain) - pure functions
# SE = dexteroushandenvs/tasks/block_assembly/allegro_hand_block_assembly_search.py.  Pinned by tests/golden/S*.npz
# (oracle/gen_golden_search.py).  The task itself is NOT built (segmentation rasteriser, its reset with 60 settle steps and the
# hand-position history are missing): this is the oracle a later round's kernels will be checked against.
SEARCH_EULER = np.array([0.0, 3.14, 1.57], dtype=F)                                              # SE:1569
SEARCH_SUCCESS_PIXELS = np.array([20, 20, 15, 20, 20, 30, 30, 20])                               # SE:1289


def search_pre_physics_targets(actions, q, prev_targets, hand_pos, hand_rot, target_pos, J, lower, upper, act_moving_average=0.6):
    """SE:1557-1596: fingers = moving average of the scaled action, clamped; arm by the IK that keeps the hand base 0.24 above and
    0.18 behind the target brick with the wrist at quat_from_euler_xyz(0, 3.14, 1.57); everything clamped to the joint limits."""
    a = actions.astype(F)
    n = a.shape[0]
    cur = np.zeros_like(prev_targets, dtype=F)
    f = scale(a[:, 7:23], lower[7:23], upper[7:23])
    f = F(act_moving_average) * f + F(1.0 - act_moving_average) * prev_targets[:, 7:23]
    cur[:, 7:23] = np.maximum(np.minimum(f, upper[7:23]), lower[7:23])
    pos_err = (target_pos - hand_pos).astype(F)
    pos_err[:, 2] += F(0.24)
    pos_err[:, 0] -= F(0.18)
    te = np.broadcast_to(SEARCH_EULER, (n, 3))
    rot_err = orientation_error(quat_from_euler_xyz(te[:, 0], te[:, 1], te[:, 2]), hand_rot)
    cur[:, :7] = q[:, :7] + control_ik(J, np.concatenate([pos_err, rot_err], axis=-1))
    return np.maximum(np.minimum(cur, upper), lower).astype(F)


def segmentation_pixel_stats(seg, ids):
    """SE:1232-1241: pixels of the target's segmentation id in the [H, W] image: count and the truncated mean row / column (0, 0
    when the target is not visible)."""
    n = seg.shape[0]
    cx, cy, num = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    for i in range(n):
        rows, cols = np.nonzero(seg[i] == ids[i])
        num[i] = rows.size
        if rows.size:
            cx[i], cy[i] = int(rows.astype(np.float32).mean()), int(cols.astype(np.float32).mean())
    return cx, cy, num


def search_obs_frame(dof, actions, lower, upper):
    """compute_contact_observations SE:1220-1230: the same 62 numbers as Orient's compute_real_observations"""
    return orient_obs_frame(dof, actions, lower, upper)


def search_state_frame(dof, actions, lower, upper, a):
    """compute_contact_asymmetric_observations SE:1168-1218 (175 numbers used of the 188-wide frame); `a` maps the attribute names the
    reference reads (fingertip positions, hand base pose, target pose, eight hand-position history means, pixel statistics, twists)."""
    n = dof.shape[0]
    q, qd = dof[..., 0], dof[..., 1]
    s = np.zeros((n, 188), dtype=F)
    s[:, 0:23] = unscale(q, lower, upper)
    s[:, 23:46] = F(0.2) * qd
    s[:, 46:49], s[:, 49:52], s[:, 52:55], s[:, 55:58] = a["arm_hand_ff_pos"], a["arm_hand_rf_pos"], a["arm_hand_mf_pos"], a["arm_hand_th_pos"]
    s[:, 58:81] = actions
    s[:, 81:88] = a["hand_base_pose"]
    s[:, 88:95] = a["segmentation_target_pose"]
    for k in range(8):
        s[:, 96 + 3 * k:99 + 3 * k] = a["hand_pos_history_%d" % k]
    s[:, 120] = a["center_x"].reshape(-1) / F(128)
    s[:, 121] = a["center_y"].reshape(-1) / F(128)
    s[:, 122] = a["point_num"].reshape(-1) / F(100)
    s[:, 123:126], s[:, 126:129] = a["hand_base_linvel"], a["hand_base_angvel"]
    for k, f in enumerate(("ff", "mf", "rf", "th")):
        s[:, 129 + 10 * k:133 + 10 * k] = a["arm_hand_%s_rot" % f]
        s[:, 133 + 10 * k:136 + 10 * k] = a["arm_hand_%s_linvel" % f]
        s[:, 136 + 10 * k:139 + 10 * k] = a["arm_hand_%s_angvel" % f]
    s[:, 169:172], s[:, 172:175] = a["segmentation_target_linvel"], a["segmentation_target_angvel"]
    return s


def search_hand_reward(target_pos, init_pos, ff, rf, mf, th, progress, reset_buf, cons_successes, successes, arm_contacts, actions,
                       max_episode_length=75.0, av_factor=0.1):
    """compute_hand_reward SE:1660-1711: min(-0.2 d, -0.06) - arm contacts - 0.005 |a|^2 + lift term; the camera-derived emergence reward
    and the heap-movement count are computed by the task but do NOT enter the reward; reset on time-out only."""
    nrm = lambda v: np.linalg.norm(v.astype(F), axis=-1).astype(F)
    d = nrm(target_pos - ff) + nrm(target_pos - mf) + nrm(target_pos - rf) + nrm(target_pos - th)          # SE:1669-1670 (no thumb weight)
    dist_rew = np.minimum(F(-0.2) * d, F(-0.06))                                                           # SE:1671
    action_penalty = (actions.astype(F) ** 2).sum(-1) * F(0.005)
    dlt = (target_pos - init_pos).astype(F)
    up = (np.clip(dlt[:, 2], 0, 0.1) * F(1000) - np.clip(dlt[:, 0], 0, 0.1) * F(1000) - np.clip(dlt[:, 1], 0, 0.1) * F(1000)).astype(F)
    reward = (dist_rew - arm_contacts.sum(-1) - action_penalty + up).astype(F)                             # SE:1685
    resets = np.where(d <= -1, 1, reset_buf)
    resets = np.where(progress >= max_episode_length - 1, 1, resets)                                       # SE:1699-1700
    num_resets = resets.sum()
    fin = (successes * resets.astype(F)).sum()
    cons = np.where(num_resets > 0, F(av_factor) * fin / max(num_resets, 1) + F(1.0 - av_factor) * cons_successes,
                    cons_successes).astype(F)
    return reward, resets.astype(np.int64), cons, up


def search_emergence_reward(seg, ids, last_pixels):
    """SE:1640-1646: 5 x the change of the target's visible pixel count"""
    pix = np.array([(seg[i] == ids[i]).sum() for i in range(seg.shape[0])], dtype=F)
    return pix, (pix - last_pixels) * F(5)


def search_heap_movement(brick_pos):
    """SE:1648-1652: number of bricks thrown out of the bin region (|x - 1| > 0.25 and |y| > 0.35 in the env frame)"""
    out = (np.abs(brick_pos[:, :, 0] - 1) > 0.25) & (np.abs(brick_pos[:, :, 1]) > 0.35)
    return out.sum(axis=1).astype(F)


# ------------------------------------------------------------------------------------------------ Search: RetriGraspTValue (SE:395-410,1133-1166)
RETRI_LAYERS = (("linear1", 1024, 650), ("linear2", 512, 1024), ("linear3", 128, 512), ("output_layer", 2, 128))   # terminal_value_function.py:12-19


def retri_tvalue_formula_weights():
    """a fixed, formula-defined parameter set for RetriGraspTValue(650, 2) (1.26 M numbers are too many for a fixture file): the golden
    generator loads exactly these into the reference's module, the tests load them into the oracle and the GPU task"""
    sd = {}
    for li, (name, out, inn) in enumerate(RETRI_LAYERS):
        i = np.arange(out, dtype=np.float64)[:, None]
        j = np.arange(inn, dtype=np.float64)[None, :]
        sd[name + ".weight"] = (np.sin(0.37 * i + 0.11 * j + 0.5 * li) * np.cos(0.013 * i * (1 + li) - 0.029 * j) / np.sqrt(inn)).astype(F)
        sd[name + ".bias"] = (0.05 * np.cos(0.7 * np.arange(out, dtype=np.float64) + li)).astype(F)
    return sd


def retri_tvalue_forward(x, sd):
    """RetriGraspTValue.forward (terminal_value_function.py:21-27): ELU after every layer, the output layer included; returns the two
    activations and tvalue = sigmoid(out)[:, 1] (SE:1133-1134)"""
    h = x.astype(F)
    for name, _, _ in RETRI_LAYERS:
        h = elu(h @ sd[name + ".weight"].T.astype(F) + sd[name + ".bias"].astype(F)).astype(F)
    return h, (F(1) / (F(1) + np.exp(-h[:, 1]))).astype(F)


def search_tvalue_buffer_update(buf, obs62, cam_rot, center_x, center_y, point_num):
    """the ten-frame buffer of SE:1155-1166: frames 0..8 <- frames 1..9, frame 9 <- [obs_buf[:, 0:62] with columns 26..29 replaced by the
    camera-frame target quaternion, centroid x / 128, centroid y / 128, pixel count / 100].  buf [N, 650] (returns a new array)"""
    n = buf.shape[0]
    out = np.zeros_like(buf, dtype=F)
    out[:, :9 * 65] = buf[:, 65:]
    fr = np.zeros((n, 65), dtype=F)
    fr[:, 0:62] = obs62
    fr[:, 26:30] = cam_rot
    fr[:, 62] = center_x.reshape(-1) / F(128)
    fr[:, 63] = center_y.reshape(-1) / F(128)
    fr[:, 64] = point_num.reshape(-1) / F(100)
    out[:, 9 * 65:] = fr
    return out
