"""Known-answer tests that pin oracle/physics_oracle.c (the plain-C definition of the physics step; PARITY UNPINNED vs
PhysX).  Closed forms: free fall, resting contact, implicit-PD step response, FK/Jacobian by finite differences,
joint-space inertia = kinetic-energy Hessian, momentum exchange of a 2-body impact."""
import numpy as np
import pytest

from oracle import physics_oracle as po


@pytest.fixture(scope="module")
def desc(scene):
    return scene.to_desc()


def base_state(scene, n=1):
    root = np.zeros((n, 142, 13), np.float32)
    root[:, :, 6] = 1
    # park every free brick far apart in the air (no contacts), fixed bricks at their places
    for i in range(72):
        root[:, 9 + i, 0:3] = [5.0 + 0.3 * (i % 9), 5.0 + 0.3 * (i // 9), 3.0]
    for i, fb in enumerate(scene.raw["fixed_bricks"]):
        root[:, 81 + i, 0:3] = fb["pos"]
    lo, hi = scene.lower, scene.upper
    pose = np.concatenate([np.array(scene.arm_prepare_pose, np.float32), 0.5 * (lo[7:] + hi[7:])]).astype(np.float32)
    dof = np.zeros((n, 23, 2), np.float32)
    dof[:, :, 0] = pose
    return root, dof, np.tile(pose, (n, 1)).astype(np.float32)


def test_free_fall_is_semi_implicit_euler(scene, desc):
    root, dof, tg = base_state(scene)
    z0 = root[0, 9, 2]
    for k in range(1, 4):
        po.simulate(desc, root, dof, tg)
        h, g, m = 1 / 120.0, -9.81, 2 * k
        np.testing.assert_allclose(root[0, 9, 9], m * h * g, rtol=1e-5)
        np.testing.assert_allclose(root[0, 9, 2] - z0, h * h * g * m * (m + 1) / 2, rtol=2e-4)
    assert abs(root[0, 9, 7]) < 1e-7 and abs(root[0, 9, 10]) < 1e-7


def test_brick_rests_on_floor(scene, desc):
    root, dof, tg = base_state(scene)
    t0 = scene.brick_types[0]
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    rest_z = floor_top + t0["half"][2] - t0["center"][2]
    root[0, 9, 0:3] = [0.25, 0.19, rest_z + 0.004]
    for _ in range(120):
        rb, contact, jac, nc = po.simulate(desc, root, dof, tg)
    assert abs(root[0, 9, 2] - rest_z) < 1.5e-3                      # penetration well below contact_offset
    assert np.linalg.norm(root[0, 9, 7:13]) < 5e-3                   # at rest
    assert abs(root[0, 9, 0] - 0.25) < 2e-3 and abs(root[0, 9, 1] - 0.19) < 2e-3   # no drift


def test_implicit_pd_step_response(scene, desc):
    """fingers-only target step: monotone, no overshoot beyond 5%, settles to the target; arm holds still."""
    root, dof, tg = base_state(scene)
    tg2 = tg.copy()
    tg2[0, 8] += 0.4
    q = []
    for _ in range(60):
        po.simulate(desc, root, dof, tg2)
        q.append(dof[0, 8, 0])
    q = np.array(q)
    assert abs(q[-1] - tg2[0, 8]) < 2e-3
    assert q.max() < tg2[0, 8] + 0.05 * 0.4
    assert np.abs(dof[0, :7, 0] - tg[0, :7]).max() < 2e-3


def test_fk_jacobian_finite_difference(scene, desc):
    rng = np.random.default_rng(0)
    lo, hi = scene.lower, scene.upper
    q = (lo + (hi - lo) * rng.uniform(0.2, 0.8, 23)).astype(np.float32)
    dof = np.zeros((1, 23, 2), np.float32)
    dof[0, :, 0] = q
    rb0, jac = po.kinematics(desc, dof)
    eps = 1e-3
    for j in range(7):
        d = dof.copy(); d[0, j, 0] += eps
        rbp, _ = po.kinematics(desc, d)
        d[0, j, 0] -= 2 * eps
        rbm, _ = po.kinematics(desc, d)
        lin = (rbp[0, 7, 0:3] - rbm[0, 7, 0:3]) / (2 * eps)
        np.testing.assert_allclose(jac[0, 0:3, j], lin, atol=2e-3)
    # link velocities are J qd
    qd = rng.normal(size=23).astype(np.float32)
    dof[0, :, 1] = qd
    rb, jac = po.kinematics(desc, dof)
    np.testing.assert_allclose(rb[0, 7, 7:10], jac[0, 0:3] @ qd[:7], atol=1e-4)
    np.testing.assert_allclose(rb[0, 7, 10:13], jac[0, 3:6] @ qd[:7], atol=1e-4)
    # base link is the fixed robot base, fingertip bodies follow the tree
    np.testing.assert_allclose(rb[0, 0, 0:3], scene.base_pos, atol=1e-6)


def test_mass_matrix_is_kinetic_energy_hessian(scene, desc):
    rng = np.random.default_rng(1)
    lo, hi = scene.lower, scene.upper
    q = (lo + (hi - lo) * rng.uniform(0.2, 0.8, 23)).astype(np.float32)
    H, Hinv = po.mass_matrix(desc, q, 0.0)          # h = 0: pure M(q) (+ armature 0)
    np.testing.assert_allclose(H, H.T, atol=1e-6)
    assert np.linalg.eigvalsh(H.astype(np.float64)).min() > 0
    np.testing.assert_allclose(H.astype(np.float64) @ Hinv.astype(np.float64), np.eye(23), atol=5e-3)
    # T = 1/2 qd^T M qd must equal sum over links of 1/2 m |v_com|^2 + 1/2 w^T I w
    qd = rng.normal(size=23).astype(np.float32)
    dof = np.zeros((1, 23, 2), np.float32); dof[0, :, 0] = q; dof[0, :, 1] = qd
    rb, _ = po.kinematics(desc, dof)
    T = 0.0
    from oracle.task_oracle import quat_apply
    for k, b in enumerate(scene.raw["robot"]["bodies"]):
        quat, w, v0, p = rb[0, k, 3:7], rb[0, k, 10:13].astype(np.float64), rb[0, k, 7:10].astype(np.float64), rb[0, k, 0:3]
        com = quat_apply(quat[None], np.array([b["com"]], np.float32))[0].astype(np.float64)
        v = v0 + np.cross(w, com)
        R = np.stack([quat_apply(quat[None], np.eye(3, dtype=np.float32)[i:i + 1])[0] for i in range(3)], 1).astype(np.float64)
        Iw = R @ np.array(b["inertia"]) @ R.T
        T += 0.5 * b["mass"] * v @ v + 0.5 * w @ Iw @ w
    np.testing.assert_allclose(0.5 * qd @ H.astype(np.float64) @ qd, T, rtol=2e-4)


def test_contact_generation_face_on_face(scene, desc):
    root, dof, tg = base_state(scene)
    t0 = scene.brick_types[0]
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    root[0, 9, 0:3] = [0.25, 0.19, floor_top + t0["half"][2] - t0["center"][2] - 0.0005]
    c, total = po.contacts(desc, root[0], dof[0])
    assert total == 4                                         # the four bottom corners, corners come first
    np.testing.assert_allclose(c[:, 5:8], [[0, 0, 1]] * 4, atol=1e-6)    # normals out of the floor slab
    np.testing.assert_allclose(c[:, 8], -0.0005, atol=2e-6)
    assert set(c[:, 1].astype(int)) == {255} and set(c[:, 0].astype(int)) == {0}


def test_free_swinging_arm_conserves_kinetic_energy(scene):
    """known answer for the velocity-product (Coriolis / centrifugal) terms: with the drives switched off and nothing in reach
    the arm + hand is a conservative system, so T = 1/2 qd^T (M(q) + armature) qd stays constant while q changes by tenths of
    a radian.  Without C(q,qd)qd the same run gains 87 % in 20 steps; the bar here is the drift of semi-implicit Euler."""
    d = scene.to_desc()
    for j in range(23):
        d.kp[j] = 0.0; d.kd[j] = 0.0; d.vel_limit[j] = 100.0
    root = np.zeros((1, 142, 13), np.float32); root[..., 6] = 1.0
    root[:, :, 0] = 50.0 + np.arange(142)[None, :] * 2.0; root[:, :, 2] = 100.0     # bricks and statics out of reach
    rng = np.random.default_rng(0)
    q0 = ((scene.lower + scene.upper) / 2).astype(np.float32)
    dof = np.zeros((1, 23, 2), np.float32); dof[0, :, 0] = q0
    dof[0, :7, 1] = rng.uniform(-1.5, 1.5, 7); dof[0, 7:, 1] = rng.uniform(-2, 2, 16)
    tg = np.tile(q0, (1, 1)).astype(np.float32)

    def kinetic(dof):
        H, _ = po.mass_matrix(d, dof[0, :, 0], 0.0)
        v = dof[0, :, 1].astype(np.float64)
        return 0.5 * v @ H.astype(np.float64) @ v

    e0 = kinetic(dof)
    for _ in range(15):                       # 0.25 s; no joint reaches its limit before step 16 in this draw
        po.simulate(d, root, dof, tg)
        assert abs(kinetic(dof) / e0 - 1.0) < 0.02
    assert np.abs(dof[0, :, 0] - q0).max() > 0.3


def test_robot_angular_damping_scales_the_joint_velocities(scene):
    """GS:546 `asset_options.angular_damping = 0.01` on the arm-hand asset: every substep multiplies the joint
    velocities by (1 - h x 0.01).  One joint moving alone with the drives off (no velocity-product coupling onto itself, constant inertia
    about its axis) loses that factor per substep relative to the undamped run (to first order: the recoil of the links above it couples back)."""
    root = np.zeros((1, 142, 13), np.float32); root[..., 6] = 1.0
    root[:, :, 0] = 50.0 + np.arange(142)[None, :] * 2.0; root[:, :, 2] = 100.0
    q0 = ((scene.lower + scene.upper) / 2).astype(np.float32)
    res = {}
    for c in (0.0, 0.01, 2.0):
        d = scene.to_desc(robot_angular_damping=c)
        for j in range(23):
            d.kp[j] = 0.0; d.kd[j] = 0.0; d.vel_limit[j] = 100.0
        dof = np.zeros((1, 23, 2), np.float32); dof[0, :, 0] = q0
        dof[0, 10, 1] = 1.0                                   # the last joint of the first finger (link_3.0): nothing hangs below it
        tg = np.tile(q0, (1, 1)).astype(np.float32)
        for _ in range(5):
            po.simulate(d, root.copy(), dof, tg)
        res[c] = float(dof[0, 10, 1])
    h = 1.0 / 120.0
    assert 0.95 < res[0.0] < 1.0                                             # free: the joint keeps spinning (the links above it recoil a little)
    np.testing.assert_allclose(res[0.01] / res[0.0], (1 - h * 0.01) ** 10, rtol=5e-5)
    np.testing.assert_allclose(res[2.0] / res[0.0], (1 - h * 2.0) ** 10, rtol=1e-2)   # (an exaggerated value makes the factor visible)


def _momentum(scene, root, idx):
    """linear momentum and angular momentum about the world origin of the listed free bricks (COM = box centre, principal inertia)"""
    from oracle import task_oracle as T
    P, L = np.zeros(3), np.zeros(3)
    for i in idx:
        t = scene.brick_types[scene.brick_type[i]]
        m = t["mass"]
        s = root[0, 9 + i].astype(np.float64)
        q = s[3:7] / np.linalg.norm(s[3:7])
        c = s[0:3] + T.quat_apply(q[None].astype(np.float32), np.array([t["center"]], np.float32))[0]
        v, w = s[7:10], s[10:13]
        wl = T.quat_apply(T.quat_conjugate(q[None].astype(np.float32)), w[None].astype(np.float32))[0].astype(np.float64)
        Lb = T.quat_apply(q[None].astype(np.float32), (np.array(t["inertia_diag"]) * wl)[None].astype(np.float32))[0]
        P += m * v
        L += np.cross(c, m * v) + Lb
    return P, L


@pytest.mark.parametrize("warm_start", [0.0, 0.8])
def test_two_body_impact_conserves_momentum(scene, warm_start):
    """two free bricks collide in mid-air without gravity: every contact applies equal and opposite impulses at
    one point, so linear momentum and angular momentum about the origin are conserved through the impact (fp32 rounding), the
    bricks end up separating, and kinetic energy does not grow."""
    desc = scene.to_desc(gravity=[0.0, 0.0, 0.0], warm_start=warm_start)
    root, dof, tg = base_state(scene)
    a, b = 0, 5                                             # a 1x2 and a 1x3 brick (different masses)
    root[0, 9 + a, 0:3] = [5.00, 9.0, 3.0]
    root[0, 9 + b, 0:3] = [5.10, 9.004, 3.006]              # slightly off-centre: the impact also spins them
    root[0, 9 + a, 7:10] = [0.6, 0.0, 0.0]
    root[0, 9 + b, 7:10] = [-0.2, 0.0, 0.0]
    root[0, 9 + b, 10:13] = [0.0, 0.0, 1.5]
    P0, L0 = _momentum(scene, root, (a, b))
    def ke():
        from oracle import task_oracle as T
        e = 0.0
        for i in (a, b):
            t = scene.brick_types[scene.brick_type[i]]
            s = root[0, 9 + i].astype(np.float64)
            wl = T.quat_apply(T.quat_conjugate(s[None, 3:7].astype(np.float32)), s[None, 10:13].astype(np.float32))[0].astype(np.float64)
            e += 0.5 * t["mass"] * s[7:10] @ s[7:10] + 0.5 * (np.array(t["inertia_diag"]) * wl) @ wl
        return e
    ke0 = ke()
    touched = False
    warm = po.WarmState(1)
    for step in range(30):
        rb, contact, jac, nc = po.simulate(desc, root, dof, tg, warm)
        touched |= nc[0] > 0
        P, L = _momentum(scene, root, (a, b))
        np.testing.assert_allclose(P, P0, rtol=0, atol=2e-6 * (1 + np.abs(P0).max() * 100))
        # angular momentum: the contact impulses conserve it exactly; the integrator keeps the WORLD angular velocity of a free brick
        # constant (no gyroscopic term), which lets L of a tumbling non-spherical brick wander by ~1e-4 relative
        np.testing.assert_allclose(L, L0, rtol=0, atol=2e-4)
    assert touched                                          # they did meet
    assert ke() <= ke0 * (1 + 1e-3)                         # the impact does not create energy (Baumgarte only acts on penetration)
    assert abs(root[0, 9 + 10, 7:13]).max() == 0.0         # the 70 parked bricks never moved


def stacked_pair_state(scene, ia, ib, yaw=0.0, dx=0.0, dy=0.0):
    """brick ib dropped from 4 mm onto brick ia, which rests on the bin floor; returns (root, dof, targets, rest z of ia, rest z of ib)"""
    root, dof, tg = base_state(scene)
    ta, tb = scene.brick_types[scene.brick_type[ia]], scene.brick_types[scene.brick_type[ib]]
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    za = floor_top + ta["half"][2] - ta["center"][2]
    zb = za + ta["center"][2] + ta["half"][2] + tb["half"][2] - tb["center"][2]
    root[0, 9 + ia, 0:3] = [0.25, 0.19, za + 0.001]
    root[0, 9 + ib, 0:3] = [0.25 + dx, 0.19 + dy, zb + 0.004]
    root[0, 9 + ib, 3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
    return root, dof, tg, za, zb


# (lower brick, upper brick, yaw of the upper, x / y offset of the upper): flush stacks of equal bricks, a stack 1 mm off, a 1x1 on a 1x1,
# a stack shifted by a quarter of its length, crossed bricks (only edge samples meet), a small brick on a 2x2 ...
# (straight bricks only: since round 5 a brick collides as the slab compound of its convex hull, and a brick put on the half-studded
# wedge types 1, 2, 3, 7 rests on their true profile - test_bricks_rest_on_the_true_profile below)
STACKS = [(6, 14, 0.0, 0.0, 0.0), (6, 14, 0.0, 0.001, 0.0), (4, 12, 0.0, 0.0, 0.0), (5, 13, 0.0, 0.0, 0.0),
          (6, 14, np.pi / 2, 0.0, 0.0), (5, 13, np.pi / 2, 0.0, 0.0), (6, 14, np.pi / 4, 0.0, 0.0), (6, 4, 0.0, 0.005, 0.003)]
# ... and four tall stacks of 1-stud-wide bricks loaded off their axis (the upper brick stands on the 26 mm wide stud row of the lower
# one), which only the warm-started solver - the default - holds: shifted by a quarter length, 3 mm / 2 mm off, turned by 1 rad, a 1x1
# turned by 0.3 rad
STACKS_WARM = STACKS + [(6, 14, 0.0, 0.03, 0.0), (6, 14, 0.0, 0.003, 0.002), (6, 14, 1.0, 0.01, 0.0), (6, 4, 0.3, 0.005, 0.003)]
WARM = 0.8


def check_stack(root, nc, e, ia, ib, yaw, dx, dy, za, zb, sink_max, drift_max):
    sink_a = za - root[e, 9 + ia, 2]
    sink_b = zb - root[e, 9 + ib, 2] - sink_a
    assert -1e-4 < sink_a < sink_max and -1e-4 < sink_b < sink_max, (sink_a, sink_b)
    assert abs(root[e, 9 + ib, 0] - 0.25 - dx) < drift_max and abs(root[e, 9 + ib, 1] - 0.19 - dy) < drift_max   # friction holds it
    qz0 = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)], np.float32)
    assert abs(abs(float(root[e, 9 + ib, 3:7] @ qz0)) - 1) < 2e-3                                    # still upright, same yaw
    assert abs(abs(float(root[e, 9 + ia, 6])) - 1) < 2e-3
    assert nc[e] == 8                                                                                # 4 on the floor + 4 between the bricks


@pytest.mark.parametrize("ia,ib,yaw,dx,dy", STACKS)
def test_stacked_bricks_stay_stacked(scene, desc, ia, ib, yaw, dx, dy):
    """the contact manifold of a pair is built on the face the two boxes meet on (separating-axis choice), with the
    4 slots of a direction given to face samples first.  Two seconds after the drop the upper brick still stands on the lower one: it
    neither sank into it (the round-1 rule pushed flush equal bricks apart sideways) nor tipped over an edge (speculative samples beside
    the lower brick used up the slots).  Default solver (cold start): resting penetration below 3 mm per interface, creep below 6 mm."""
    root, dof, tg, za, zb = stacked_pair_state(scene, ia, ib, yaw, dx, dy)
    for _ in range(120):
        rb, contact, jac, nc = po.simulate(desc, root, dof, tg)
    check_stack(root, nc, 0, ia, ib, yaw, dx, dy, za, zb, 3e-3, 6e-3)


@pytest.mark.parametrize("ia,ib,yaw,dx,dy", STACKS_WARM)
def test_stacked_bricks_with_warm_start(scene, ia, ib, yaw, dx, dy):
    """DESIGN.md section 3.E, warm_start = 0.8 (optional): every solve starts from 0.8 x the impulses the same contacts ended the previous
    solve with.  Resting penetration drops from 2 mm to below 0.6 mm per interface, creep below 1.5 mm in two seconds, and the two tall
    off-axis stacks that creep over under the cold solver stand."""
    warm_desc = scene.to_desc(warm_start=WARM)
    root, dof, tg, za, zb = stacked_pair_state(scene, ia, ib, yaw, dx, dy)
    warm = po.WarmState(1)
    for _ in range(120):
        rb, contact, jac, nc = po.simulate(warm_desc, root, dof, tg, warm)
    check_stack(root, nc, 0, ia, ib, yaw, dx, dy, za, zb, 6e-4, 2.5e-3 if (dx, dy) == (0.003, 0.002) else 1.5e-3)


def test_cold_solver_lets_the_off_axis_stack_creep_over(scene, desc):
    """the known limit of the solver without its impulse cache (16 Jacobi iterations from zero impulses): a flush
    stack rests 2 mm deep per interface, and the tall stack loaded 3 mm off its axis is on the floor after four seconds"""
    root, dof, tg, za, zb = stacked_pair_state(scene, 6, 14)
    for _ in range(120):
        po.simulate(desc, root, dof, tg)
    assert 1.5e-3 < za - root[0, 9 + 6, 2] < 3e-3
    root, dof, tg, za, zb = stacked_pair_state(scene, 6, 14, 0.0, 0.003, 0.002)
    for _ in range(240):
        po.simulate(desc, root, dof, tg)
    assert zb - root[0, 9 + 14, 2] > 0.03                                                            # the upper brick is on the floor


def test_sliding_friction_decelerates_at_mu_g(scene, desc):
    """Coulomb friction, mu = 1: a brick sliding on the floor along its long axis loses mu g dt of speed per
    step until it stops, without turning (the friction pyramid's axes are x and y for a vertical normal)"""
    root, dof, tg = base_state(scene)
    t0 = scene.brick_types[0]
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    root[0, 9, 0:3] = [0.25, 0.19, floor_top + t0["half"][2] - t0["center"][2] - 0.0015]
    for _ in range(30):
        po.simulate(desc, root, dof, tg)                     # settle
    root[0, 9, 7:10] = [0.6, 0.0, 0.0]
    v = [0.6]
    for _ in range(5):
        po.simulate(desc, root, dof, tg)
        v.append(float(root[0, 9, 7]))
    np.testing.assert_allclose(np.diff(v[:4]), -9.81 / 60.0, rtol=0.02)      # 0.6 -> 0.436 -> 0.273 -> 0.110
    assert abs(v[4]) < 5e-3 and abs(v[5]) < 1e-3                             # stopped, and stays stopped (no friction overshoot)
    assert np.abs(root[0, 9, 10:13]).max() < 0.05 and abs(root[0, 9, 8]) < 1e-3


def test_joint_limit_holds(scene, desc):
    """joint limits as clamp + velocity projection (section 3.F): a finger joint driven 0.5 rad beyond its upper limit ends AT the limit, at rest"""
    root, dof, tg = base_state(scene)
    j = 8
    tg2 = tg.copy()
    tg2[0, j] = scene.upper[j] + 0.5
    for _ in range(60):
        po.simulate(desc, root, dof, tg2)
    assert dof[0, j, 0] == np.float32(scene.upper[j]) and dof[0, j, 1] == 0.0
    tg2[0, j] = scene.lower[j] - 0.5
    for _ in range(90):
        po.simulate(desc, root, dof, tg2)
    assert dof[0, j, 0] == np.float32(scene.lower[j]) and dof[0, j, 1] == 0.0


# ------------------------------------------------------------------------------------------------ compound shapes (round 5)
def _floor_top(scene):
    return scene.statics[6]["center"][2] + scene.statics[6]["half"][2]


def test_contacts_follow_the_hull_not_the_bounding_box(scene, desc):
    """a 1x1 brick hovering 3 mm above the SLOPE of the 1x3 wedge brick (type 3: stud on one third, then a ramp down to 6 mm; as a compound: a full-length slab up
    to the stud base and a slab over the studded end that thins towards the ramp) is deep inside the wedge's bounding box and touches nothing; the same brick 1 mm above the stud has four contacts with it"""
    root, dof, tg = base_state(scene)
    w, b = 3, 4                                         # brick 3 = type 3 (1x3_curve), brick 4 = type 4 (1x1)
    tw, tb = scene.brick_types[3], scene.brick_types[4]
    z0 = _floor_top(scene) + tw["half"][2] - tw["center"][2]
    root[0, 9 + w, 0:3] = [0.25, 0.19, z0]
    low = tw["sub"][0]["center"][2] + tw["sub"][0]["half"][2]            # top of the lower slab = the ramp's level in the compound
    top = tw["center"][2] + tw["half"][2]                               # stud tops = top of the bounding box
    assert top - low > 0.015
    bottom = tb["center"][2] - tb["half"][2]
    root[0, 9 + b, 0:3] = [0.25 + 0.028, 0.19, z0 + low + 0.003 - bottom]   # over the ramp end, 3 mm above the lower slab
    c, n = po.contacts(desc, root[0], dof[0])
    assert n == 4 and set(c[:, 1]) == {255.0}                           # only the wedge on the floor
    root[0, 9 + b, 0:3] = [0.25 - 0.03, 0.19, z0 + top + 0.001 - bottom]    # over the stud, 1 mm above it
    c, n = po.contacts(desc, root[0], dof[0])
    pair = c[((c[:, 0] == w) & (c[:, 1] == b)) | ((c[:, 0] == b) & (c[:, 1] == w))]        # (either direction of the pair)
    assert n == 8 and len(pair) == 4 and np.allclose(pair[:, 8], 0.001, atol=2e-5) and np.allclose(np.abs(pair[:, 7]), 1.0)


def test_bricks_rest_on_the_true_profile(scene):
    """dynamics on the compound shapes: (a) a 1x1 brick put on the stud of the 1x3 wedge stands there, one put over the ramp ends
    > 15 mm lower, leaning on the ramp; (b) a 1x2 brick lying on its side rests on its BODY's side face - 15 mm from its axis - with
    four contacts (the stud row is 2.1 mm narrower and stays clear of the floor)."""
    desc = scene.to_desc()
    tw, tb = scene.brick_types[3], scene.brick_types[4]
    z0 = _floor_top(scene) + tw["half"][2] - tw["center"][2]
    top = tw["center"][2] + tw["half"][2]
    bottom = tb["center"][2] - tb["half"][2]
    ends = []
    for dx in (-0.03, 0.03):
        root, dof, tg = base_state(scene)
        root[0, 9 + 3, 0:3] = [0.25, 0.19, z0 + 0.001]
        root[0, 9 + 4, 0:3] = [0.25 + dx, 0.19, z0 + top + 0.003 - bottom]
        warm = po.WarmState(1)
        for _ in range(150):
            po.simulate(desc, root, dof, tg, warm)
        ends.append(root[0, 9 + 4].copy())
        assert np.abs(root[0, 9 + 3, 0:2] - [0.25, 0.19]).max() < 3e-3 and np.abs(root[0, 9 + 4, 7:13]).max() < 0.05
    on_stud, on_ramp = ends
    assert abs(on_stud[2] - (z0 + top - bottom)) < 1.5e-3 and abs(on_stud[6]) > 0.999         # standing on the stud, upright
    assert on_stud[2] - on_ramp[2] > 0.015                                                     # the other one went down the ramp
    # (b)
    root, dof, tg = base_state(scene)
    t0 = scene.brick_types[0]
    s = np.sqrt(0.5)
    root[0, 9, 3:7] = [s, 0, 0, s]                                      # rolled 90 degrees about x: the -y side face down
    root[0, 9, 0:3] = [0.25, 0.19, _floor_top(scene) + t0["half"][1] + 0.002]
    warm = po.WarmState(1)
    for _ in range(90):
        rb, contact, jac, nc = po.simulate(desc, root, dof, tg, warm)
    assert nc[0] == 4 and abs(root[0, 9, 2] - (_floor_top(scene) + 0.015)) < 1e-3
    assert abs(abs(root[0, 9, 3:7] @ np.array([s, 0, 0, s], np.float32)) - 1) < 1e-3


def seated_brick_state(scene, desc, brick_type=7):
    """InsertSim scene of env 0 (plate 4x4x1): the target brick (brick 0, here a 2x2) 2 mm above its seat over the four centre studs"""
    root, dof, tg = base_state(scene)
    desc.brick_type[0] = brick_type
    site_z = 0.618 + 0.0375                                             # IS:1123-1125
    root[0, 9, 0:3] = [0.25, -0.2, site_z + 0.002]
    return root, dof, tg, site_z


def test_hollow_brick_engages_the_studs_of_the_base_plate(scene):
    """SURVEY.md section 8(a) rows A0 / P3 (GS:810-838, IS:698-709, 740-767): the target brick of BlockAssemblyInsertSim collides as the
    hollow compound of its mesh, the base plate as body + studs.  A 2x2 brick put over four studs drops onto the plate's BODY (its walls
    around the studs); pushed sideways with 1 N (gravity tilted by its weight) it moves by the clearance of 1.25 mm and stops; pulled upwards
    it comes off freely; as a plain hull (seg_hollow = 0) it would stand 20 mm higher, on the stud tops."""
    desc = scene.to_desc(task_kind=2)
    root, dof, tg, site_z = seated_brick_state(scene, desc)
    m = scene.brick_types[7]["mass"]
    warm = po.WarmState(1)
    for _ in range(40):
        rb, contact, jac, nc = po.simulate(desc, root, dof, tg, warm)
    assert abs(root[0, 9, 2] - site_z) < 1e-3 and np.abs(root[0, 9, 0:2] - [0.25, -0.2]).max() < 3e-4 and nc[0] >= 16
    desc.gravity[0] = 1.0 / m                                           # 1 N along +x
    for _ in range(90):
        po.simulate(desc, root, dof, tg, warm)
    assert 5e-4 < root[0, 9, 0] - 0.25 < 2.2e-3 and abs(root[0, 9, 1] + 0.2) < 5e-4            # against the studs, after the 1.25 mm of play
    assert abs(root[0, 9, 2] - site_z) < 1e-3 and np.abs(root[0, 9, 7:10]).max() < 5e-3 and abs(root[0, 9, 6]) > 0.9999
    desc.gravity[0] = 0.0
    desc.gravity[2] = 2.0                                               # pulled upwards: nothing holds it
    for _ in range(30):
        po.simulate(desc, root, dof, tg, warm)
    assert root[0, 9, 2] - site_z > 0.02 and np.abs(root[0, 9, 0:2] - [0.25, -0.2]).max() < 3e-3
    # the same brick as a plain hull stands on the stud tops
    hull = scene.to_desc(task_kind=2, seg_hollow=0)
    root, dof, tg, site_z = seated_brick_state(scene, hull)
    root[0, 9, 2] += 0.02
    warm = po.WarmState(1)
    for _ in range(60):
        po.simulate(hull, root, dof, tg, warm)
    assert abs(root[0, 9, 2] - (site_z + 0.0387 - 0.01875)) < 1e-3
