"""-m gpu: the HIP physics kernel (one wavefront per env) against oracle/physics_oracle.c, the plain-C CPU
restatement of the same step (PARITY UNPINNED vs PhysX - see the oracle header).  Teacher forcing: both sides
start every compared step from the same state.  SURVEY.md §8(a) rows P1-P5."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import physics_oracle as po  # noqa: E402


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def state(golden_dir):
    return np.load(os.path.join(golden_dir, "P1_settled_state.npz"))


def test_kinematics_and_jacobian(scene):
    from seqdex_amd.sim import SdxSim
    n = 256
    s = SdxSim(n)
    try:
        rng = np.random.default_rng(0)
        lo, hi = scene.lower, scene.upper
        dof = np.stack([lo + (hi - lo) * rng.uniform(size=(n, 23)), rng.normal(size=(n, 23))], -1).astype(np.float32)
        s.DOF.copy_(_dev(dof.reshape(-1, 2)))
        s.refresh_kinematics()
        torch.cuda.synchronize()
        rb, jac = po.kinematics(s._desc, dof)
        np.testing.assert_allclose(s.RB.cpu().numpy()[:, :24], rb[:, :24], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s.JAC_EEF.cpu().numpy(), jac, rtol=2e-5, atol=2e-5)
    finally:
        s.close()


def _one_step(s, root, dof, targets):
    n = root.shape[0]
    s.ROOT.copy_(_dev(root.reshape(-1, 13)))
    s.DOF.copy_(_dev(dof.reshape(-1, 2)))
    s.TARGETS.copy_(_dev(targets))
    s.simulate()
    torch.cuda.synchronize()
    return (s.ROOT.cpu().numpy().reshape(n, 142, 13), s.DOF.cpu().numpy().reshape(n, 23, 2), s.RB.cpu().numpy(),
            s.CONTACT.cpu().numpy().reshape(n, 165, 3), s.JAC_EEF.cpu().numpy(), s.NCONTACTS.cpu().numpy())


@pytest.mark.parametrize("warm_start", [0.0, 0.8])
def test_one_step_teacher_forcing(state, scene, warm_start):
    """same start state, one simulate() on each side.  Contact-rich piles amplify fp32 rounding through the
    discrete contact set (and, since the face manifold of DESIGN.md section 3.D, through its separating-axis choice), so the bar is: identical
    contact counts, robot pose to 1e-4 (velocities 5e-4 / 1e-3), brick poses to 2e-5 m for >= 99% and brick velocities to 2e-3 m/s for
    >= 98% of the bricks, at most 2 bricks further than 1e-4 m off and none further than 1 mm (measured over 16 steps: 99.6 - 100 % within
    2e-5 m, identical contact sets, one step in which a sample on the contact offset fell on the other side: 1 brick 0.12 mm off)."""
    from seqdex_amd.sim import SdxSim
    n = state["root"].shape[0]
    s = SdxSim(n, warm_start=warm_start)      # default: cold solver; 0.8: the optional warm start of DESIGN.md section 3.E
    try:
        root, dof = state["root"].copy(), state["dof"].copy()
        o_warm = po.WarmState(n)            # both sides keep their own impulse cache from step to step
        for it in range(3):
            g_root, g_dof, g_rb, g_contact, g_jac, g_nc = _one_step(s, root, dof, state["targets"])
            o_root, o_dof = root.copy(), dof.copy()
            o_rb, o_contact, o_jac, o_nc = po.simulate(s._desc, o_root, o_dof, state["targets"], o_warm)
            if warm_start > 0:
                np.testing.assert_array_equal(s.WARM_COUNT.cpu().numpy(), g_nc)       # the cache holds the contacts of the last solve
            # the counts are those of the SECOND substep, i.e. after one substep of fma-vs-separate rounding: a sample that sits on the
            # contact offset may fall on either side (1 of ~1100 contacts seen); the first substep's lists are identical by construction
            assert np.abs(g_nc - o_nc).max() <= 2 and (g_nc == o_nc).mean() >= 0.75, (g_nc, o_nc)
            np.testing.assert_allclose(g_dof[..., 0], o_dof[..., 0], rtol=1e-4, atol=1e-4)     # joint positions
            np.testing.assert_allclose(g_dof[..., 1], o_dof[..., 1], rtol=1e-3, atol=2e-3)     # joint velocities (fingers in contact: 1.2e-3 rad/s seen)
            np.testing.assert_allclose(g_rb[:, :24, :7], o_rb[:, :24, :7], rtol=1e-4, atol=1e-4)    # link poses
            np.testing.assert_allclose(g_rb[:, :24, 7:], o_rb[:, :24, 7:], rtol=1e-3, atol=4e-3)    # link twists (fingertips sum the joint velocity differences)
            np.testing.assert_allclose(g_jac, o_jac, rtol=1e-4, atol=1e-4)
            dp = np.abs(g_root[:, 9:81, 0:7] - o_root[:, 9:81, 0:7]).max(-1)       # a brick whose velocity differs by 4e-3 m/s is 3e-5 m off
            # round 5: a convex pair of compounds contributes the box pair with the smallest separation bound, and the manifold the face
            # samples that span the contact patch.  The first version of that rule broke ties by strict comparison: one box pair in the 8
            # golden envs (two samples of one box edge, equally far from the line p1 p2) chose another sample on the device than in the
            # oracle and 16 Jacobi iterations spread it over eight bricks, up to 1.9 mm (tests/helpers/parity_keys.py lists the differing
            # contacts).  With the tie margins of the rule (a later candidate must win by 1 um / 1e-8 m^2) the contact SETS of device and
            # oracle are identical and the bar is back near rounds 1-4's: 99 % of the 576 bricks within 2e-5 m, at most 2 beyond 1e-4, none beyond 1 mm
            assert (dp >= 1e-4).sum() <= 2 and dp.max() < 1e-3 and (dp < 2e-5).mean() >= 0.99, (float(dp.max()), float((dp < 2e-5).mean()), int((dp >= 1e-4).sum()))
            dv = np.abs(g_root[:, 9:81, 7:13] - o_root[:, 9:81, 7:13]).max(-1)
            assert (dv < 2e-3).mean() >= 0.98, float((dv < 2e-3).mean())
            np.testing.assert_allclose(g_contact[:, :24], o_contact[:, :24], rtol=5e-3, atol=5e-2)
            np.testing.assert_array_equal(g_root[:, 81:141], root[:, 81:141])     # fixed bricks untouched
            root, dof = o_root, o_dof                                              # teacher forcing
    finally:
        s.close()


def test_free_fall_and_invariants(scene):
    """a single brick row dropped from the spawn lattice: free fall is exact semi-implicit Euler until contact;
    afterwards nothing tunnels through the floor slab and the robot holds its targets."""
    from seqdex_amd.sim import SdxSim
    n = 64
    s = SdxSim(n)
    try:
        root0 = s.ROOT.cpu().numpy().reshape(n, 142, 13).copy()
        pose = np.concatenate([np.array(scene.arm_prepare_pose, np.float32),
                               0.5 * (np.array(scene.finger_reset_unscaled, np.float32) + 1)
                               * (scene.upper[7:] - scene.lower[7:]) + scene.lower[7:]]).astype(np.float32)
        dof = np.zeros((n, 23, 2), np.float32); dof[:, :, 0] = pose
        s.DOF.copy_(_dev(dof.reshape(-1, 2)))
        s.TARGETS.copy_(_dev(np.tile(pose, (n, 1))))
        s.simulate()
        torch.cuda.synchronize()
        r1 = s.ROOT.cpu().numpy().reshape(n, 142, 13)
        top = 9 + 64  # bricks of the highest spawn layer are in free fall during the first step
        h, g = 1.0 / 120.0, -9.81
        rbp = s.RB.cpu().numpy()[0, :24, 0:3]                                      # the parked hand reaches into the top layer
        clear = [b for b in range(top, top + 8) if np.linalg.norm(rbp - root0[0, b, 0:3], axis=-1).min() > 0.12]
        assert len(clear) >= 6
        np.testing.assert_allclose(r1[:, clear, 9], 2 * h * g, rtol=1e-5)
        np.testing.assert_allclose(r1[:, clear, 2] - root0[:, clear, 2], h * h * g * 3, rtol=1e-4, atol=1e-6)
        for _ in range(150):
            s.simulate()
        torch.cuda.synchronize()
        r = s.ROOT.cpu().numpy().reshape(n, 142, 13)
        assert np.isfinite(r).all()
        assert r[:, 9:81, 2].min() > 0.60                      # nothing below the bin bottom
        assert np.abs(r[:, 9:81, 0] - 0.25).max() < 0.31 and np.abs(r[:, 9:81, 1] - 0.19).max() < 0.22   # inside the bin
        assert np.abs(s.DOF.cpu().numpy().reshape(n, 23, 2)[:, :, 0] - pose).max() < 2e-3
        assert np.linalg.norm(r[:, 9:81, 7:10], axis=-1).mean() < 0.02
    finally:
        s.close()


@pytest.mark.parametrize("warm_start", [0.0, 0.8])
def test_stacked_bricks_stay_stacked_on_device(scene, warm_start):
    """the stacking cases of tests/test_physics_oracle.py (flush equal bricks, offsets, crossed bricks), one per env, through k_physics:
    two seconds after the drop every upper brick still stands on its lower brick, and the device trajectory ends where the oracle's
    does (these quiet scenes do not amplify rounding: 0.2 mm / 2e-3 in the quaternion)."""
    from seqdex_amd.sim import SdxSim
    from test_physics_oracle import STACKS, STACKS_WARM, check_stack, stacked_pair_state
    STACKS = STACKS_WARM if warm_start > 0 else STACKS
    n = len(STACKS)
    roots, rests = [], []
    for (ia, ib, yaw, dx, dy) in STACKS:
        root, dof, tg, za, zb = stacked_pair_state(scene, ia, ib, yaw, dx, dy)
        roots.append(root[0]); rests.append((za, zb))
    root = np.stack(roots).astype(np.float32)
    dof = np.repeat(dof, n, 0); tg = np.repeat(tg, n, 0)
    s = SdxSim(n, warm_start=warm_start)
    try:
        ref, refd = root.copy(), dof.copy()
        o_warm = po.WarmState(n)
        s.ROOT.copy_(_dev(root.reshape(-1, 13))); s.DOF.copy_(_dev(dof.reshape(-1, 2))); s.TARGETS.copy_(_dev(tg))
        for _ in range(120):
            s.simulate()
            po.simulate(s._desc, ref, refd, tg, o_warm)
        torch.cuda.synchronize()
        r = s.ROOT.cpu().numpy().reshape(n, 142, 13)
        nc = s.NCONTACTS.cpu().numpy()
        for e, (ia, ib, yaw, dx, dy) in enumerate(STACKS):
            za, zb = rests[e]
            check_stack(r, nc, e, ia, ib, yaw, dx, dy, za, zb, *((6e-4, 2.5e-3 if (dx, dy) == (0.003, 0.002) else 1.5e-3) if warm_start > 0 else (3e-3, 6e-3)))
            np.testing.assert_allclose(r[e, [9 + ia, 9 + ib], 0:3], ref[e, [9 + ia, 9 + ib], 0:3], rtol=0, atol=2e-4)
            np.testing.assert_allclose(np.abs((r[e, [9 + ia, 9 + ib], 3:7] * ref[e, [9 + ia, 9 + ib], 3:7]).sum(-1)), 1.0, rtol=0, atol=2e-3)
    finally:
        s.close()


def test_compound_shapes_on_device(scene):
    """round 5 (GS:717-731, 810-838, IS:698-709): the known-answer cases of the compound shapes through
    k_physics.  Env 0 of an InsertSim scene: the hollow 2x2 target brick drops over four studs of the base plate onto the plate's body,
    resists 1 N sideways after the 1.25 mm of play, and comes off upwards; GraspSim scene: a 1x1 brick stands on the stud of the 1x3 wedge,
    one put over its ramp ends more than 15 mm lower; a 1x2 brick on its side rests on its body's side face.  Device trajectories end where
    the oracle's do."""
    from seqdex_amd.sim import SdxSim
    from test_physics_oracle import base_state, seated_brick_state
    ins = scene.to_desc(task_kind=2)
    root, dof, tg, site_z = seated_brick_state(scene, ins)
    m = scene.brick_types[7]["mass"]
    s = SdxSim(1, desc=ins)
    try:
        ref, refd, o_warm = root.copy(), dof.copy(), po.WarmState(1)
        s.ROOT.copy_(_dev(root.reshape(-1, 13))); s.DOF.copy_(_dev(dof.reshape(-1, 2))); s.TARGETS.copy_(_dev(tg))
        for _ in range(40):
            s.simulate(); po.simulate(ins, ref, refd, tg, o_warm)
        torch.cuda.synchronize()
        r = s.ROOT.cpu().numpy().reshape(1, 142, 13)
        assert abs(r[0, 9, 2] - site_z) < 1e-3 and np.abs(r[0, 9, 0:2] - [0.25, -0.2]).max() < 3e-4 and s.NCONTACTS.cpu().numpy()[0] >= 16
        np.testing.assert_allclose(r[0, 9, 0:3], ref[0, 9, 0:3], atol=2e-4)
    finally:
        s.close()
    # sideways push and upward pull: fresh simulators on descriptors with the tilted gravity, started from the seated state
    for grav, steps, check in (([1.0 / m, 0.0, -9.81], 90, "push"), ([0.0, 0.0, 2.0], 30, "pull")):
        d2 = scene.to_desc(task_kind=2, gravity=grav)
        d2.brick_type[0] = 7
        s = SdxSim(1, desc=d2)
        try:
            s.ROOT.copy_(_dev(r.reshape(-1, 13))); s.DOF.copy_(_dev(dof.reshape(-1, 2))); s.TARGETS.copy_(_dev(tg))
            for _ in range(steps):
                s.simulate()
            torch.cuda.synchronize()
            q = s.ROOT.cpu().numpy().reshape(1, 142, 13)
            if check == "push":
                assert 5e-4 < q[0, 9, 0] - 0.25 < 2.2e-3 and abs(q[0, 9, 1] + 0.2) < 5e-4 and abs(q[0, 9, 2] - site_z) < 1e-3
                assert np.abs(q[0, 9, 7:10]).max() < 5e-3 and abs(q[0, 9, 6]) > 0.9999
            else:
                assert q[0, 9, 2] - site_z > 0.02 and np.abs(q[0, 9, 0:2] - [0.25, -0.2]).max() < 3e-3
        finally:
            s.close()
    # true profile
    desc = scene.to_desc()
    tw, tb, t0 = scene.brick_types[3], scene.brick_types[4], scene.brick_types[0]
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    z0 = floor_top + tw["half"][2] - tw["center"][2]
    top, bottom = tw["center"][2] + tw["half"][2], tb["center"][2] - tb["half"][2]
    parts = []
    for dx in (-0.03, 0.03):
        rr, dd, tt = base_state(scene)
        rr[0, 9 + 3, 0:3] = [0.25, 0.19, z0 + 0.001]
        rr[0, 9 + 4, 0:3] = [0.25 + dx, 0.19, z0 + top + 0.003 - bottom]
        parts.append((rr, dd, tt))
    rr, dd, tt = base_state(scene)
    sq = np.sqrt(0.5)
    rr[0, 9, 3:7] = [sq, 0, 0, sq]
    rr[0, 9, 0:3] = [0.25, 0.19, floor_top + t0["half"][1] + 0.002]
    parts.append((rr, dd, tt))
    root = np.concatenate([p[0] for p in parts]); dof = np.concatenate([p[1] for p in parts]); tg = np.concatenate([p[2] for p in parts])
    s = SdxSim(3, desc=desc)
    try:
        s.ROOT.copy_(_dev(root.reshape(-1, 13))); s.DOF.copy_(_dev(dof.reshape(-1, 2))); s.TARGETS.copy_(_dev(tg))
        for _ in range(150):
            s.simulate()
        torch.cuda.synchronize()
        q = s.ROOT.cpu().numpy().reshape(3, 142, 13)
        nc = s.NCONTACTS.cpu().numpy()
        assert abs(q[0, 13, 2] - (z0 + top - bottom)) < 1.5e-3 and abs(q[0, 13, 6]) > 0.999 and q[0, 13, 2] - q[1, 13, 2] > 0.015
        assert nc[2] == 4 and abs(q[2, 9, 2] - (floor_top + 0.015)) < 1e-3
    finally:
        s.close()
