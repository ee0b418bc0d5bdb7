"""-m gpu: BlockAssemblyInsertSim (second policy of the grasp -> insert chain) on the HIP path
(scene.task_kind = 2), called through the C ABI, against the golden vectors captured from the reference's own InsertSim module
(tests/golden/I*.npz), the numpy oracle and - for the per-env base plates - the C physics oracle."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import task_oracle as T  # noqa: E402


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def insert16():
    from seqdex_amd.sim import SdxSim
    s = SdxSim(16, device="cuda:0", seed=22, task_kind=2, max_episode_length=125.0)
    yield s
    s.close()


def test_insert_pre_physics_golden(insert16, golden_dir):
    f = np.load(os.path.join(golden_dir, "I2_pre_physics.npz"))
    s, n = insert16, 16
    s.RESET.zero_()
    dof = torch.zeros(n, 23, 2)
    dof[:, :, 0] = torch.as_tensor(f["q"])
    s.DOF.copy_(dof.view(-1, 2).cuda())
    s.PREV_TARGETS.copy_(_dev(f["prev_targets"]))
    s.PROGRESS.fill_(90)                                   # past GraspSim's step-75 / step-100 switches: InsertSim has none
    s.RB[:, 7, 3:7] = _dev(f["hand_rot"])
    s.JAC_EEF.copy_(_dev(f["J"]))
    s.pre_physics(_dev(f["actions"]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), f["cur_targets"], rtol=2e-4, atol=1e-4)          # the reference's numbers
    np.testing.assert_array_equal(s.TARGETS.cpu().numpy(), s.PREV_TARGETS.cpu().numpy())
    np.testing.assert_allclose(s.INSERT_AUX.cpu().numpy()[:, 0:3], f["rot_err"], rtol=3e-5, atol=3e-6)
    want, _ = T.insert_pre_physics_targets(f["actions"], f["q"], f["prev_targets"], f["hand_rot"], f["J"], f["lower"], f["upper"],
                                           f["target_euler"][0])
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), want, rtol=2e-4, atol=1e-4)                      # and the oracle's


def test_insert_observations_golden(insert16, golden_dir):
    f = np.load(os.path.join(golden_dir, "I3_observations.npz"))
    s, n = insert16, 16
    assert tuple(s.OBS.shape) == (n, 75) and tuple(s.STATES.shape) == (n, 564)           # 75 x 1 (IS:175); 188 real state columns
    s.OBS.zero_(); s.STATES.zero_()
    s.INIT_POS.copy_(_dev(f["init_pos"]))
    s.INIT_ROT.copy_(_dev(f["init_rot"]))
    for c in range(3):
        p = "c%d_" % c
        s.ROOT.copy_(_dev(f[p + "root"]))
        s.RB.copy_(_dev(f[p + "rb"]))
        s.DOF.copy_(_dev(f[p + "dof"]).view(-1, 2))
        s.CONTACT.copy_(_dev(f[p + "contact"]))
        s.ACTIONS.copy_(_dev(f[p + "actions"]))
        s.PROGRESS.copy_(_dev(np.rint(f[p + "states_buf"][:, 141] * 125.0).astype(np.int64)))
        s.compute_observations()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.OBS.cpu().numpy(), f[p + "obs_buf"], rtol=3e-5, atol=3e-5)
        st = s.STATES.cpu().numpy()
        np.testing.assert_allclose(st[:, :188], f[p + "states_buf"], rtol=3e-5, atol=3e-5)
        assert not st[:, 188:].any()                                                     # one frame: the rest of the row stays zero
        np.testing.assert_array_equal(s.OBS_CLAMPED.cpu().numpy(), np.clip(s.OBS.cpu().numpy(), -5, 5))
        root = f[p + "root"].reshape(n, 142, 13)
        tpos = root[np.arange(n), f["seg_index_in_env"], 0:3]
        np.testing.assert_allclose(s.INSERT_AUX.cpu().numpy()[:, 3], np.linalg.norm(tpos - f[p + "extra_target_pos"], axis=-1),
                                   rtol=3e-5, atol=3e-6)


def test_insert_reward_golden(golden_dir):
    from seqdex_amd.sim import SdxSim
    f = np.load(os.path.join(golden_dir, "I5_reward.npz"))
    m = f["progress"].shape[0]
    s = SdxSim(m, device="cuda:0", task_kind=2, max_episode_length=float(f["max_episode_length"]))
    try:
        root = s.ROOT.view(m, 142, 13)
        seg = torch.tensor([s.scene.seg_index(i) for i in range(m)]).cuda()
        ar = torch.arange(m).cuda()
        root[ar, seg, 0:3] = _dev(f["target_pos"])
        root[ar, seg, 3:7] = _dev(f["target_rot"])
        # the reference's function takes the insertion SITE; the plate actor sits below it by the env's offsets (IS:1123-1130)
        off, sym = T.insert_extra_target(np.zeros((m, 3), np.float32), f["extra_rot"])
        np.testing.assert_allclose(sym, f["symmetry_rot"], rtol=1e-6, atol=1e-6)
        root[:, 141, 0:3] = _dev(f["extra_pos"] - off)
        root[:, 141, 3:7] = _dev(f["extra_rot"])
        for body, key in zip(s.scene.fingertip_bodies, ["ff", "mf", "rf", "th"]):
            s.RB[:, body, 0:3] = _dev(f[key] - np.array([0, 0, 0.04], np.float32))
            s.RB[:, body, 3:7] = torch.tensor([0.0, 0, 0, 1]).cuda()
        s.PROGRESS.copy_(_dev(f["progress"] - 1))        # post_physics_step increments first
        s.RESET.copy_(_dev(f["reset_buf"]))
        s.SUCCESSES.copy_(_dev(f["successes"]))
        s.CONS_SUCCESSES.copy_(_dev(f["cons_in"]))
        s.INSERT_AUX[:, 0:3] = _dev(f["rot_err"])
        s.post_physics()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.REW.cpu().numpy(), f["reward"], rtol=2e-4, atol=2e-5)
        np.testing.assert_array_equal(s.RESET.cpu().numpy(), f["resets"])
        np.testing.assert_allclose(s.CONS_SUCCESSES.cpu().numpy(), f["cons_out"], rtol=1e-6)
        assert (f["reward"] > 1).any() and (f["reward"] < 1).any()
    finally:
        s.close()


def test_insert_reset_from_grasp_states_and_success_flag():
    """reset_idx IS:1416-1494: brick and hand come from a random slot of the env's brick-type ring with zero velocities, PD targets
    = restored joint positions, the plate returns to (0.25, -0.2, 0.618) with ONE yaw for the whole reset event, and the outcome of
    the finished episode lands in success_buf (IS:1352)."""
    from seqdex_amd.sim import SdxSim
    n, k = 32, 5
    s = SdxSim(n, device="cuda:0", seed=5, task_kind=2, max_episode_length=125.0)
    try:
        g = torch.Generator().manual_seed(1)
        obj = torch.randn(8, k, 13, generator=g)
        hand = torch.randn(8, k, 23, 2, generator=g) * 0.1
        s.HARVEST_OBJ[:, :k] = obj.cuda()
        s.HARVEST_HAND[:, :k] = hand.cuda()
        s.HARVEST_COUNT.fill_(k)
        # a first step so that total_steps > 0, then a forced reset of every other env
        s.step(torch.zeros(n, 23).cuda())
        aux = torch.zeros(n, 8)
        aux[:, 3] = torch.where(torch.arange(n) % 4 == 0, 0.01, 0.05)
        aux[:, 4] = 0.1
        s.INSERT_AUX.copy_(aux.cuda())
        mask = torch.zeros(n, dtype=torch.uint8)
        mask[::2] = 1
        root_before = s.ROOT.view(n, 142, 13).cpu().numpy().copy()
        s.reset_idx(mask.cuda())
        torch.cuda.synchronize()
        root = s.ROOT.view(n, 142, 13).cpu().numpy()
        dof = s.DOF.view(n, 23, 2).cpu().numpy()
        yaws = set()
        for e in range(n):
            seg = s.scene.seg_index(e)
            if e % 2:
                np.testing.assert_array_equal(root[e, seg], root_before[e, seg])         # untouched
                continue
            hit = [j for j in range(k) if np.array_equal(root[e, seg, :7], obj[e % 8, j, :7].numpy())]
            assert len(hit) == 1, e
            j = hit[0]
            assert not root[e, seg, 7:].any()
            np.testing.assert_array_equal(dof[e, :, 0], hand[e % 8, j, :, 0].numpy())
            assert not dof[e, :, 1].any()
            np.testing.assert_array_equal(s.TARGETS.cpu().numpy()[e], dof[e, :, 0])
            np.testing.assert_array_equal(s.PREV_TARGETS.cpu().numpy()[e], dof[e, :, 0])
            np.testing.assert_array_equal(s.INIT_POS.cpu().numpy()[e], root[e, seg, 0:3])
            np.testing.assert_array_equal(s.INIT_ROT.cpu().numpy()[e], root[e, seg, 3:7])
            np.testing.assert_allclose(root[e, 141, 0:3], [0.25, -0.2, 0.618], atol=1e-7)
            yaws.add(round(float(2 * np.arctan2(root[e, 141, 5], root[e, 141, 6])), 4))
            assert int(s.SUCCESS_BUF.cpu()[e]) == (1 if e % 4 == 0 else 0)
            assert int(s.PROGRESS.cpu()[e]) == 0 and int(s.RESET.cpu()[e]) == 0
        assert len(yaws) == 1 and list(yaws)[0] in (0.0, 1.57)
        slots = {tuple(root[e, s.scene.seg_index(e), :3]) for e in range(0, n, 2)}
        assert len(slots) > 4                                                            # the draw differs between envs
    finally:
        s.close()


def test_insert_plate_variants_match_c_oracle_and_seat_the_brick():
    """the base plate is 4x4x{1,2,4} by env % 3: (a) one simulator step with the target brick touching the plate agrees with the C
    oracle env by env; (b) a brick released just above the insertion site comes to rest on the plate body with its origin at the
    site height 0.618 + 0.0375 (1 + env % 3) (IS:1123-1125)."""
    from oracle import physics_oracle as po
    from seqdex_amd.sim import SdxSim
    n = 9
    s = SdxSim(n, device="cuda:0", seed=3, task_kind=2, max_episode_length=125.0)
    try:
        s.RESET.zero_()
        root = s.ROOT.view(n, 142, 13)
        zq = np.zeros((n, 4), np.float32); zq[:, 3] = 1
        site, _ = T.insert_extra_target(np.tile(np.array([[0.25, -0.2, 0.618]], np.float32), (n, 1)), zq)
        assert np.allclose(site[:, 2] - 0.618, 0.0375 * (1 + np.arange(n) % 3))
        seg = [s.scene.seg_index(e) for e in range(n)]
        for e in range(n):
            root[e, 9:81, 0] += 5.0                                                       # the pile out of the way (over no support:
            root[e, 9:81, 2] = 0.3 + 0.002 * torch.arange(72).cuda()                      # they fall freely, far from everything)
            root[e, seg[e], 0:3] = _dev(site[e] + np.array([0, 0, 0.004], np.float32))
            root[e, seg[e], 3:7] = torch.tensor([0.0, 0, 0, 1]).cuda()
            root[e, seg[e], 7:13] = 0
        s.TARGETS.copy_(s.DOF.view(n, 23, 2)[:, :, 0])
        tg = s.TARGETS.cpu().numpy().copy()
        for _ in range(3):                                                                # teacher forcing, as tests/test_gpu_physics_parity.py
            o_root = s.ROOT.cpu().numpy().reshape(n, 142, 13).copy()
            o_dof = s.DOF.cpu().numpy().reshape(n, 23, 2).copy()
            s.simulate()
            torch.cuda.synchronize()
            _, _, _, o_nc = po.simulate(s._desc, o_root, o_dof, tg)
            g_root = s.ROOT.cpu().numpy().reshape(n, 142, 13)
            np.testing.assert_array_equal(s.NCONTACTS.cpu().numpy(), o_nc)
            for e in range(n):
                np.testing.assert_allclose(g_root[e, seg[e], 0:7], o_root[e, seg[e], 0:7], atol=2e-5)
                np.testing.assert_allclose(g_root[e, seg[e], 7:13], o_root[e, seg[e], 7:13], atol=2e-3)
        for _ in range(60):
            s.simulate()
        torch.cuda.synchronize()
        r_gpu = s.ROOT.cpu().numpy().reshape(n, 142, 13)
        for e in range(n):
            assert abs(r_gpu[e, seg[e], 2] - site[e, 2]) < 0.004, (e, r_gpu[e, seg[e], 2], site[e, 2])
            assert np.abs(r_gpu[e, seg[e], 0:2] - site[e, 0:2]).max() < 0.01
    finally:
        s.close()


def test_insert_task_end_to_end(scene):
    """BlockAssemblyInsertSim through the VecTask surface with synthetic grasp states: shapes, finite numbers, episodes end by the
    reference's three rules, the reward stays in (0, 2]."""
    import yaml
    from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_insert_sim.yaml")))
    n = 24
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblyInsertSim(cfg, device_type="cuda", device_id=0, headless=True, seed=3, piles_per_type=2,
                                  synthetic_states_per_type=6)
    assert task.grasp_states_source == "synthetic" and task.num_obs == 75
    env = RLgamesVecTaskPython(task, "cuda:0")
    obs = env.reset()
    assert tuple(obs["obs"].shape) == (n, 75) and tuple(obs["states"].shape) == (n, 564)
    g = torch.Generator().manual_seed(0)
    resets_seen, rews = 0, []
    for t in range(140):
        a = (torch.rand(n, 23, generator=g) * 2 - 1).cuda() * 0.2
        obs, rew, reset, extras = env.step(a)
        resets_seen += int(reset.sum())
        rews.append(rew.cpu().numpy().copy())
    torch.cuda.synchronize()
    rews = np.stack(rews)
    assert np.isfinite(obs["obs"].cpu().numpy()).all() and np.isfinite(rews).all()
    assert (rews > 0).all() and (rews <= 2.0).all()
    assert resets_seen >= n                                      # every env finished at least one episode (time-out at 124 at the latest)
    assert resets_seen < 12 * n                                  # ... and episodes are not cut at once: the synthetic states start with
                                                                 # the wrist near its target orientation and the fingers around the brick
    assert not obs["states"][:, 188:].any()
    r = task.sim.ROOT.cpu().numpy().reshape(n, 142, 13)
    assert np.isfinite(r).all()
    np.testing.assert_allclose(r[:, 141, 0:3], np.tile([0.25, -0.2, 0.618], (n, 1)), atol=1e-6)
    assert set(extras["success_buf"].cpu().numpy().tolist()) <= {0, 1}
