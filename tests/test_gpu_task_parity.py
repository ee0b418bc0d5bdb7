"""-m gpu: the HIP task kernels, called through the C ABI, against (a) the golden vectors produced by the
reference's own functions and (b) the numpy oracle on larger seeded inputs.  SURVEY.md §8(a) rows T2-T9."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import task_oracle as T  # noqa: E402

TOL = dict(rtol=3e-5, atol=3e-5)


def _dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope="module")
def sim16():
    from seqdex_amd.sim import SdxSim
    s = SdxSim(16, device="cuda:0", seed=22)
    yield s
    s.close()


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("phase", [0, 1, 2, 3])
def test_pre_physics_golden(sim16, golden_dir, phase):
    f = g(golden_dir, "F2_pre_physics.npz")
    p = "p%d_" % phase
    s = sim16
    n = 16
    s.RESET.zero_()
    dof = torch.zeros(n, 23, 2)
    dof[:, :, 0] = torch.as_tensor(f[p + "q"])
    s.DOF.copy_(dof.view(-1, 2).cuda())
    s.PREV_TARGETS.copy_(_dev(f[p + "prev_targets"]))
    s.PROGRESS.copy_(_dev(f[p + "progress"]))
    s.INIT_POS.copy_(_dev(f[p + "init_pos"]))
    s.RB[:, 7, 0:3] = _dev(f[p + "hand_pos"])
    s.JAC_EEF.copy_(_dev(f[p + "J"]))
    actions = _dev(f[p + "actions"] * 1.3)          # also exercises the +-1 clamp of VR:166
    exp_in = np.clip(f[p + "actions"] * 1.3, -1, 1)
    s.pre_physics(actions)
    torch.cuda.synchronize()
    want = T.pre_physics_targets(exp_in, f[p + "q"], f[p + "prev_targets"], f[p + "progress"], f[p + "init_pos"],
                                 f[p + "hand_pos"], f[p + "J"], f["lower"], f["upper"])
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), want, rtol=2e-4, atol=5e-5)
    np.testing.assert_array_equal(s.TARGETS.cpu().numpy(), s.PREV_TARGETS.cpu().numpy())
    np.testing.assert_array_equal(s.ACTIONS.cpu().numpy(), exp_in.astype(np.float32))
    # and exactly the reference's numbers when the actions are un-scaled
    s.PREV_TARGETS.copy_(_dev(f[p + "prev_targets"]))
    s.pre_physics(_dev(f[p + "actions"]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), f[p + "cur_targets"], rtol=2e-4, atol=5e-5)


def test_observations_golden(sim16, golden_dir, scene):
    f = g(golden_dir, "F3_observations.npz")
    s = sim16
    s.set_tvalue_weights({k[3:]: f[k] for k in f.files if k.startswith("tv_")})
    s.OBS.zero_(); s.STATES.zero_()
    s.INIT_POS.copy_(_dev(f["init_pos"]))
    s.INIT_ROT.copy_(_dev(f["init_rot"]))
    for c in range(4):
        p = "c%d_" % c
        s.ROOT.copy_(_dev(f[p + "root"]))
        s.RB.copy_(_dev(f[p + "rb"]))
        s.DOF.copy_(_dev(f[p + "dof"]).view(-1, 2))
        s.CONTACT.copy_(_dev(f[p + "contact"]))
        s.ACTIONS.copy_(_dev(f[p + "actions"]))
        s.compute_observations()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.OBS.cpu().numpy(), f[p + "obs_buf"], **TOL)
        np.testing.assert_allclose(s.STATES.cpu().numpy(), f[p + "states_buf"], **TOL)
        np.testing.assert_array_equal(s.OBS_CLAMPED.cpu().numpy(), np.clip(s.OBS.cpu().numpy(), -5, 5))
        np.testing.assert_array_equal(s.STATES_CLAMPED.cpu().numpy(), np.clip(s.STATES.cpu().numpy(), -5, 5))
        np.testing.assert_allclose(s.FINGER_DIST.cpu().numpy(), f[p + "finger_dist"], **TOL)
        np.testing.assert_allclose(s.TVALUE.cpu().numpy(), f[p + "tvalue"], rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(s.ARM_CONTACTS.cpu().numpy(), f[p + "contacts"])
    assert (np.abs(f["c3_obs_buf"]) > 5).any()      # the clamp path was exercised


def test_reward_golden(golden_dir):
    from seqdex_amd.sim import SdxSim
    f = g(golden_dir, "F5_reward.npz")
    m = f["progress"].shape[0]
    s = SdxSim(m, device="cuda:0")
    try:
        root = s.ROOT.view(m, 142, 13)
        seg = torch.tensor([s.scene.seg_index(i) for i in range(m)]).cuda()
        root[torch.arange(m).cuda(), seg, 0:3] = _dev(f["target_pos"])
        tips = s.scene.fingertip_bodies
        for body, key in zip(tips, ["ff", "mf", "rf", "th"]):
            s.RB[:, body, 0:3] = _dev(f[key] - np.array([0, 0, 0.04], np.float32))
            s.RB[:, body, 3:7] = torch.tensor([0.0, 0, 0, 1]).cuda()
        s.INIT_POS.copy_(_dev(f["init_pos"]))
        s.PROGRESS.copy_(_dev(f["progress"] - 1))        # post_physics_step increments first (GS:1641)
        s.RESET.copy_(_dev(f["reset_buf"]))
        s.CONS_SUCCESSES.copy_(_dev(f["cons_in"]))
        s.post_physics()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.REW.cpu().numpy(), f["reward"], rtol=2e-5, atol=2e-6)
        np.testing.assert_array_equal(s.RESET.cpu().numpy(), f["resets"])
        np.testing.assert_array_equal(s.PROGRESS.cpu().numpy(), f["progress"])
        np.testing.assert_allclose(s.CONS_SUCCESSES.cpu().numpy(), f["cons_out"], rtol=1e-6)
        np.testing.assert_allclose(s.META_REW.cpu().numpy(), f["reward"], rtol=2e-5, atol=2e-6)
    finally:
        s.close()


def test_reset_idx_golden(sim16, golden_dir):
    f = g(golden_dir, "F8_reset_idx.npz")
    s = sim16
    n = 16
    s.load_initial_states(f["piles"])
    s.ROOT.copy_(_dev(f["root_before"]))
    s.DOF.copy_(_dev(f["dof_before"]))
    s.PREV_TARGETS.copy_(_dev(f["prev_before"]))
    s.TARGETS.copy_(_dev(f["cur_before"]))
    s.PROGRESS.copy_(_dev(f["progress_before"]))
    s.SUCCESSES.fill_(1.0); s.META_REW.fill_(1.0)
    mask = np.zeros(n, np.uint8); mask[f["env_ids"]] = 1
    choice = np.zeros(n, np.int32); choice[f["env_ids"]] = f["pile_choice"]
    s.RESET.copy_(_dev(mask.astype(np.int64)))
    s.reset_idx(_dev(mask), _dev(choice))
    torch.cuda.synchronize()
    ra = f["root_after"].reshape(n, 142, 13)
    got = s.ROOT.cpu().numpy().reshape(n, 142, 13)
    np.testing.assert_array_equal(got[:, 9:141], ra[:, 9:141])          # bricks: exact copy of the saved pile
    m = mask.astype(bool)
    np.testing.assert_allclose(got[m, 2, 0:3], np.tile(np.array(s.scene.goal_reset_pos, np.float32), (m.sum(), 1)), atol=1e-5)  # GS:1345-1348 with the scene's own goal_init_state
    np.testing.assert_array_equal(got[~m], f["root_before"].reshape(n, 142, 13)[~m])
    np.testing.assert_allclose(s.DOF.cpu().numpy(), f["dof_after"], atol=1e-6)
    np.testing.assert_allclose(s.PREV_TARGETS.cpu().numpy(), f["prev_after"], atol=1e-6)
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), f["cur_after"], atol=1e-6)
    np.testing.assert_array_equal(s.PROGRESS.cpu().numpy(), f["progress_after"])
    np.testing.assert_array_equal(s.RESET.cpu().numpy(), f["reset_after"])
    np.testing.assert_array_equal(s.INIT_POS.cpu().numpy()[m], f["init_pos_after"][m])
    np.testing.assert_array_equal(s.INIT_ROT.cpu().numpy()[m], f["init_rot_after"][m])
    assert (s.SUCCESSES.cpu().numpy()[m] == 0).all() and (s.META_REW.cpu().numpy()[m] == 0).all()
    assert (s.SUCCESSES.cpu().numpy()[~m] == 1).all()
    np.testing.assert_array_equal(s.PILE_CHOICE.cpu().numpy()[m], choice[m])


def test_task_kernels_vs_oracle_1024(scene):
    """full-size (N=1024) seeded comparison of obs/reward against the numpy oracle + stacking property."""
    from seqdex_amd.sim import SdxSim
    n = 1024
    s = SdxSim(n, device="cuda:0")
    try:
        rng = np.random.default_rng(3)
        lo, hi = scene.lower, scene.upper
        obs_prev = np.zeros((n, 396), np.float32); st_prev = np.zeros((n, 564), np.float32)
        seg = np.array([scene.seg_index(i) for i in range(n)])
        init_pos = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        init_rot = rng.normal(size=(n, 4)).astype(np.float32); init_rot /= np.linalg.norm(init_rot, axis=1, keepdims=True)
        s.INIT_POS.copy_(_dev(init_pos)); s.INIT_ROT.copy_(_dev(init_rot))
        for it in range(3):
            root = rng.normal(size=(n, 142, 13)).astype(np.float32) * 0.3
            rb = rng.normal(size=(n, 165, 13)).astype(np.float32) * 0.3
            for a in (root, rb):
                a[..., 3:7] /= np.linalg.norm(a[..., 3:7], axis=-1, keepdims=True)
            dof = np.stack([lo + (hi - lo) * rng.uniform(size=(n, 23)), rng.normal(size=(n, 23))], -1).astype(np.float32)
            contact = (rng.normal(size=(n, 165, 3)) * 0.08).astype(np.float32)
            actions = rng.uniform(-1, 1, (n, 23)).astype(np.float32)
            progress = rng.integers(0, 149, n)
            s.ROOT.copy_(_dev(root.reshape(-1, 13))); s.RB.copy_(_dev(rb)); s.DOF.copy_(_dev(dof.reshape(-1, 2)))
            s.CONTACT.copy_(_dev(contact.reshape(n, -1))); s.ACTIONS.copy_(_dev(actions)); s.PROGRESS.copy_(_dev(progress))
            s.RESET.zero_()
            s.post_physics()
            torch.cuda.synchronize()
            o, st, d = T.compute_observation_frames(root, rb, dof, contact, actions, seg, init_pos, init_rot, lo, hi,
                                                    np.array(scene.camera_offset_quat, np.float32),
                                                    np.array(scene.camera_offset_pos, np.float32), scene.fingertip_bodies)
            obs_prev = T.stack_frames(obs_prev, o); st_prev = T.stack_frames(st_prev, st)
            np.testing.assert_allclose(s.OBS.cpu().numpy(), obs_prev, **TOL)
            np.testing.assert_allclose(s.STATES.cpu().numpy(), st_prev, **TOL)
            rew, resets, _, _ = T.compute_hand_reward(d["target_pos"], init_pos, d["ff_pos"], d["rf_pos"], d["mf_pos"],
                                                      d["th_pos"], progress + 1, np.zeros(n, np.int64), np.zeros(1, np.float32))
            np.testing.assert_allclose(s.REW.cpu().numpy(), rew, rtol=1e-4, atol=1e-5)
            np.testing.assert_array_equal(s.RESET.cpu().numpy(), resets)
    finally:
        s.close()


def test_terminal_state_harvesting(scene):
    """GS:1398-1442: on reset, envs whose target brick sits at y < 0 with finger_dist < 0.6 and tvalue > 0.8 store their
    hand dof state and target root state in the ring buffer of their brick-type group (only when total_steps > 0)."""
    from seqdex_amd.sim import SdxSim
    n = 16
    s = SdxSim(n)
    try:
        a = torch.zeros(n, 23).cuda()
        s.step(a)                                    # total_steps becomes 1; every env was reset at step 0 (nothing harvested)
        torch.cuda.synchronize()
        assert int(s.HARVEST_COUNT.sum()) == 0
        root = s.ROOT.view(n, 142, 13)
        seg = torch.tensor([scene.seg_index(i) for i in range(n)]).cuda()
        idx = torch.arange(n).cuda()
        root[idx, seg, 1] = torch.where(idx % 2 == 0, -0.1, 0.2).float()     # even envs carried the brick to y < 0
        s.FINGER_DIST.fill_(0.3)
        s.TVALUE.copy_(torch.where(idx % 4 == 0, 0.9, 0.5).float())          # only envs 0,4,8,12 pass the T-value gate
        dof_before = s.DOF.view(n, 23, 2).clone()
        tgt_before = root[idx, seg].clone()
        s.RESET.fill_(1)
        s.pre_physics(a)
        torch.cuda.synchronize()
        cnt = s.HARVEST_COUNT.cpu().numpy()
        assert cnt.tolist() == [2, 0, 0, 0, 2, 0, 0, 0]                      # envs {0,8} -> group 0, {4,12} -> group 4
        hh, ho = s.HARVEST_HAND.cpu().numpy(), s.HARVEST_OBJ.cpu().numpy()
        for grp, envs in ((0, (0, 8)), (4, (4, 12))):
            got = {tuple(np.round(ho[grp, k], 5)) for k in range(2)}
            want = {tuple(np.round(tgt_before[e].cpu().numpy(), 5)) for e in envs}
            assert got == want
            got_h = {tuple(np.round(hh[grp, k].ravel(), 5)) for k in range(2)}
            want_h = {tuple(np.round(dof_before[e].cpu().numpy().ravel(), 5)) for e in envs}
            assert got_h == want_h
        # round 4: every append also writes (step << 24 | env); read in key order the rows are in serial (step, env) order whatever slot
        # the atomics handed out (SdxSim.ring_rows: what the hand-off lists and the T-value trainer consume)
        keys = s.HARVEST_KEYS.cpu().numpy()
        for grp, envs in ((0, (0, 8)), (4, (4, 12))):
            k = keys[grp, :2]
            assert sorted((k & 0xFFFFFF).tolist()) == list(envs) and len(set((k >> 24).tolist())) == 1 and int(k[0] >> 24) >= 1
            rows = s.ring_rows(s.HARVEST_OBJ[grp], s.HARVEST_KEYS[grp], 2).cpu().numpy()
            np.testing.assert_array_equal(rows[0], tgt_before[envs[0]].cpu().numpy())       # the lower env first
            np.testing.assert_array_equal(rows[1], tgt_before[envs[1]].cpu().numpy())
        tvc = s.TV_COUNT.cpu().numpy()
        assert tvc.tolist() == [4, 12]                                           # every finished episode logged its outcome: 4 harvested, 12 not
        tk = s.TV_KEYS.cpu().numpy()
        assert sorted((tk[0, :4] & 0xFFFFFF).tolist()) == [0, 4, 8, 12]
        assert sorted((tk[1, :12] & 0xFFFFFF).tolist()) == [e for e in range(16) if e % 4 != 0]
        succ = s.ring_rows(s.TV_SUCCESS, s.TV_KEYS[0], 4)
        assert succ.shape == (4, 4) and bool(torch.isfinite(succ).all())
    finally:
        s.close()
