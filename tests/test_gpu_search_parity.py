"""-m gpu: BlockAssemblySearch pieces on the HIP path (scene.task_kind = 3) through the C ABI: the segmentation camera against the
numpy ray caster oracle/camera_oracle.py (PARITY UNPINNED against Isaac Gym's renderer, see its header), the per-step tensor code
against the golden vectors captured from the reference's Search module (tests/golden/S*.npz)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import camera_oracle as CO  # noqa: E402
from oracle import task_oracle as T  # noqa: E402


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_segmentation_camera_matches_numpy_ray_caster(scene):
    from seqdex_amd.sim import SdxSim
    n = 8
    s = SdxSim(n, device="cuda:0", seed=2, task_kind=3)
    try:
        assert tuple(s.SEG_IMAGE.shape) == (n, 128, 128) and s.SEG_IMAGE.dtype == torch.int16
        g = torch.Generator().manual_seed(0)
        root = s.ROOT.view(n, 142, 13)
        for e in range(n):      # scatter the free bricks over the bin with random orientations; env 3's target brick is hidden far away
            root[e, 9:81, 0] = (0.05 + 0.4 * torch.rand(72, generator=g)).cuda()
            root[e, 9:81, 1] = (0.02 + 0.34 * torch.rand(72, generator=g)).cuda()
            root[e, 9:81, 2] = (0.63 + 0.12 * torch.rand(72, generator=g)).cuda()
            q = torch.randn(72, 4, generator=g)
            root[e, 9:81, 3:7] = (q / q.norm(dim=1, keepdim=True)).cuda()
        root[3, scene.seg_index(3), 0:3] = torch.tensor([3.0, 3.0, 0.3]).cuda()
        s.refresh_kinematics()
        s.render_segmentation()
        torch.cuda.synchronize()
        img = s.SEG_IMAGE.cpu().numpy()
        pix = s.SEG_PIXELS.cpu().numpy()
        r = s.ROOT.view(n, 142, 13).cpu().numpy()
        rb = s.RB.cpu().numpy()
        for e in (0, 3, 5):
            want = CO.render(s._desc, r[e], rb[e])
            assert (img[e] != want).mean() < 0.003, (e, float((img[e] != want).mean()))     # silhouette pixels may flip (fp32 orders)
            assert len(np.unique(want)) > 20                                                 # many bricks are in view
        for e in range(n):
            num, cx, cy = CO.pixel_stats(img[e], scene.seg_index(e) - 9 + 1)                 # statistics of the kernel's own image: exact
            assert (int(pix[e, 0]), int(pix[e, 1]), int(pix[e, 2])) == (num, cx, cy), (e, pix[e], num, cx, cy)
        assert pix[3, 0] == 0 and (pix[:, 0] > 0).sum() >= 4, pix[:, 0]                       # some targets are buried, most are in view
        first = pix[:, 0].copy()
        np.testing.assert_allclose(s.EMERGENCE.cpu().numpy(), 5.0 * first)                   # previous count was 0 (SE:1645)
        s.render_segmentation()
        torch.cuda.synchronize()
        assert not s.EMERGENCE.cpu().numpy().any()                                           # nothing moved: no emergence
    finally:
        s.close()


@pytest.fixture(scope="module")
def search16():
    from seqdex_amd.sim import SdxSim
    s = SdxSim(16, device="cuda:0", seed=22, task_kind=3, max_episode_length=75.0, act_moving_average=0.6, target_euler=[0.0, 3.14, 1.57])
    yield s
    s.close()


def test_search_pre_physics_golden(search16, golden_dir, scene):
    f = np.load(os.path.join(golden_dir, "S2_pre_physics.npz"))
    s, n = search16, 16
    s.RESET.zero_()
    dof = torch.zeros(n, 23, 2)
    dof[:, :, 0] = torch.as_tensor(f["q"])
    s.DOF.copy_(dof.view(-1, 2).cuda())
    s.PREV_TARGETS.copy_(_dev(f["prev_targets"]))
    s.PROGRESS.fill_(80)                                   # past Orient's step-75 lift: Search has none
    s.RB[:, 7, 0:3] = _dev(f["hand_pos"])
    s.RB[:, 7, 3:7] = _dev(f["hand_rot"])
    root = s.ROOT.view(n, 142, 13)
    for e in range(n):
        root[e, scene.seg_index(e), 0:3] = _dev(f["target_pos"][e])
    s.JAC_EEF.copy_(_dev(f["J"]))
    s.pre_physics(_dev(f["actions"]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), f["cur_targets"], rtol=2e-4, atol=1e-4)      # the reference's numbers
    want = T.search_pre_physics_targets(f["actions"], f["q"], f["prev_targets"], f["hand_pos"], f["hand_rot"], f["target_pos"], f["J"],
                                        f["lower"], f["upper"])
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), want, rtol=2e-4, atol=1e-4)


def test_search_observation_and_state_layout(search16, golden_dir, scene):
    """62-number observation and Search's own asymmetric frame (SE:1168-1218) against the numpy oracle (itself pinned to the reference
    on CPU), with the attribute values the reference would read derived from the same rigid-body / root / joint states."""
    f = np.load(os.path.join(golden_dir, "F3_observations.npz"))     # GraspSim's state fixtures serve as inputs
    s, n = search16, 16
    s.OBS.zero_(); s.STATES.zero_()
    s.ROOT.copy_(_dev(f["c0_root"])); s.RB.copy_(_dev(f["c0_rb"])); s.DOF.copy_(_dev(f["c0_dof"]).view(-1, 2))
    s.CONTACT.copy_(_dev(f["c0_contact"])); s.ACTIONS.copy_(_dev(f["c0_actions"]))
    pix = torch.zeros(n, 4)
    pix[:, 0] = torch.arange(n) * 7.0
    pix[:, 1] = torch.arange(n) + 20.0
    pix[:, 2] = 100.0 - torch.arange(n)
    s.SEG_PIXELS.copy_(pix.cuda())
    s.compute_observations()
    torch.cuda.synchronize()
    root = f["c0_root"].reshape(n, 142, 13)
    rb, dof, act = f["c0_rb"], f["c0_dof"], f["c0_actions"]
    obs = np.zeros((n, 186), np.float32)
    obs[:, :62] = T.search_obs_frame(dof, act, f["lower"], f["upper"])
    np.testing.assert_allclose(s.OBS.cpu().numpy(), obs, rtol=3e-5, atol=3e-5)
    tgt = root[np.arange(n), f["seg_index_in_env"]]
    tip = lambda b: (rb[:, b, 0:3] + T.quat_apply(rb[:, b, 3:7], np.broadcast_to(T.FT_OFFSET, (n, 3)))).astype(np.float32)
    ft = scene.fingertip_bodies
    a = dict(arm_hand_ff_pos=tip(ft[0]), arm_hand_mf_pos=tip(ft[1]), arm_hand_rf_pos=tip(ft[2]), arm_hand_th_pos=tip(ft[3]),
             hand_base_pose=rb[:, 7, 0:7], segmentation_target_pose=tgt[:, 0:7], hand_base_linvel=rb[:, 7, 7:10],
             hand_base_angvel=rb[:, 7, 10:13], segmentation_target_linvel=tgt[:, 7:10], segmentation_target_angvel=tgt[:, 10:13],
             center_x=pix[:, 1].numpy(), center_y=pix[:, 2].numpy(), point_num=pix[:, 0].numpy())
    for k in range(8):
        a["hand_pos_history_%d" % k] = np.zeros((n, 3), np.float32)
    for nm, b in zip(("ff", "mf", "rf", "th"), ft):
        a["arm_hand_%s_rot" % nm], a["arm_hand_%s_linvel" % nm], a["arm_hand_%s_angvel" % nm] = rb[:, b, 3:7], rb[:, b, 7:10], rb[:, b, 10:13]
    want = T.search_state_frame(dof, act, f["lower"], f["upper"], a)
    st = s.STATES.cpu().numpy()
    np.testing.assert_allclose(st[:, :188], want, rtol=3e-5, atol=3e-5)
    assert not st[:, 188:].any()


def test_search_reward_golden(golden_dir):
    from seqdex_amd.sim import SdxSim
    f = np.load(os.path.join(golden_dir, "S5_reward.npz"))
    m = f["progress"].shape[0]
    s = SdxSim(m, device="cuda:0", task_kind=3, max_episode_length=float(f["max_episode_length"]))
    try:
        root = s.ROOT.view(m, 142, 13)
        seg = torch.tensor([s.scene.seg_index(i) for i in range(m)]).cuda()
        ar = torch.arange(m).cuda()
        root[ar, seg, 0:3] = _dev(f["target_pos"])
        for body, key in zip(s.scene.fingertip_bodies, ["ff", "mf", "rf", "th"]):
            s.RB[:, body, 0:3] = _dev(f[key] - np.array([0, 0, 0.04], np.float32))
            s.RB[:, body, 3:7] = torch.tensor([0.0, 0, 0, 1]).cuda()
        s.INIT_POS.copy_(_dev(f["init_pos"]))
        s.ACTIONS.copy_(_dev(f["actions"]))
        cf = torch.zeros(m, 165, 3)
        cf[:, 1:7, 2] = torch.as_tensor(f["arm_contacts"]) * 1.0          # |force| >= 0.1 on the flagged arm links (SE: contacts >= 0.1)
        s.CONTACT.copy_(cf.view(m, -1).cuda())
        s.PROGRESS.copy_(_dev(f["progress"] - 1))                         # post_physics_step increments first
        s.RESET.copy_(_dev(f["reset_buf"]))
        s.SUCCESSES.copy_(_dev(f["successes"]))
        s.CONS_SUCCESSES.copy_(_dev(f["cons_in"]))
        s.post_physics()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.REW.cpu().numpy(), f["reward"], rtol=2e-4, atol=2e-3)      # the lift term scales metres by 1000
        np.testing.assert_array_equal(s.RESET.cpu().numpy(), f["resets"])
        np.testing.assert_allclose(s.CONS_SUCCESSES.cpu().numpy(), f["cons_out"], rtol=1e-6)
    finally:
        s.close()


def test_search_task_end_to_end(scene):
    """BlockAssemblySearch through the VecTask surface: the first step drops the pile (60 settling steps) and renders; the last step of
    the episode parks the hand and renders again; the reset event that follows labels every env success / failure by its pixel count
    and hands the successful piles on to BlockAssemblyOrient."""
    import yaml
    from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient
    from seqdex_amd.tasks.block_assembly_search import BlockAssemblySearch
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_search.yaml")))
    n = 16
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblySearch(cfg, device_type="cuda", device_id=0, headless=True, seed=5)
    env = RLgamesVecTaskPython(task, "cuda:0")
    obs = env.reset()
    torch.cuda.synchronize()
    assert tuple(obs["obs"].shape) == (n, 186) and tuple(obs["states"].shape) == (n, 564)
    r = task.sim.ROOT.view(n, 142, 13).cpu().numpy()
    # the lattice (up to z = 1.16 since it starts above the floor slab) has fallen into the bin; a brick in a few hundred goes over the
    # 10 cm wall on the way and keeps falling
    zz = r[:, 9:81, 2]
    assert zz.max() < 0.85 and (zz > 0.55).mean() > 0.99, (float(zz.max()), float((zz > 0.55).mean()))
    pix0 = task.sim.SEG_PIXELS.cpu().numpy().copy()
    assert (pix0[:, 0] >= 0).all() and pix0[:, 0].max() > 0                      # rendered after the settling steps
    g = torch.Generator().manual_seed(0)
    rews = []
    for t in range(76):
        obs, rew, reset, extras = env.step(((torch.rand(n, 23, generator=g) * 2 - 1) * 0.3).cuda())
        rews.append(rew.cpu().numpy().copy())
    torch.cuda.synchronize()
    assert np.isfinite(np.stack(rews)).all() and np.isfinite(obs["states"].cpu().numpy()).all()
    tvc = task.sim.TV_COUNT.cpu().numpy()
    pc = task.sim.PILE_HARVEST_COUNT.cpu().numpy()
    assert tvc.sum() == n and pc.sum() == tvc[0], (tvc, pc)                      # one labelled reset event; successes handed on
    assert int(task.extras["success_buf"].sum()) == tvc[0]
    st = obs["states"].cpu().numpy()
    assert not st[:, 96:120].any() and not st[:, 188:].any()
    # the ring keeps every success of a brick-type group (512 slots, as Orient's), not only the latest one: a second episode adds to it
    assert task.sim.PILE_HARVEST.shape[1] == 512
    for t in range(75):
        env.step(((torch.rand(n, 23, generator=g) * 2 - 1) * 0.3).cuda())
    torch.cuda.synchronize()
    tvc2, pc2 = task.sim.TV_COUNT.cpu().numpy(), task.sim.PILE_HARVEST_COUNT.cpu().numpy()
    assert tvc2.sum() == 2 * n and pc2.sum() == tvc2[0] and (pc2 >= pc).all(), (tvc2, pc2)
    ring = task.sim.PILE_HARVEST.cpu().numpy()
    for grp in range(8):
        for slot in range(min(int(pc2[grp]), 4)):
            assert np.abs(ring[grp, slot, :72, :3]).max() > 0.1, (grp, slot)     # a pile state was stored in every counted slot
        if pc2[grp] >= 2:
            assert np.abs(ring[grp, 0] - ring[grp, 1]).max() > 1e-4              # ... and the second success did not overwrite the first
    piles = task.pile_terminal_states()
    if piles is not None:                                                        # every brick-type group had a success
        ocfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_orient.yaml")))
        ocfg["env"]["numEnvs"] = n
        ot = BlockAssemblyOrient(ocfg, device_type="cuda", device_id=0, headless=True, seed=1, initial_piles=piles)
        ot.step(torch.zeros(n, 23).cuda())
        torch.cuda.synchronize()
        assert np.isfinite(ot.sim.ROOT.cpu().numpy()).all()


def test_search_retri_tvalue_and_temporal_buffer(search16, golden_dir, scene):
    """RetriGraspTValue(650, 2) on the ten-frame buffer (SE:395-410,1133-1166): the value of a step is the network applied to the buffer as
    it stood BEFORE the step's frame is appended (golden vectors from the reference's module class), and the append is the shift of
    SE:1155-1166 (numpy restatement)."""
    f = np.load(os.path.join(golden_dir, "S7_retri_tvalue.npz"))
    g3 = np.load(os.path.join(golden_dir, "F3_observations.npz"))
    s, n = search16, 16
    s.set_retri_tvalue_weights(T.retri_tvalue_formula_weights())
    assert tuple(s.TVALUE_OBS.shape) == (n, 652)
    buf = np.zeros((n, 652), np.float32)
    buf[:, :650] = f["x"]
    s.TVALUE_OBS.copy_(_dev(buf))
    s.ROOT.copy_(_dev(g3["c0_root"])); s.RB.copy_(_dev(g3["c0_rb"])); s.DOF.copy_(_dev(g3["c0_dof"]).view(-1, 2))
    s.CONTACT.copy_(_dev(g3["c0_contact"])); s.ACTIONS.copy_(_dev(g3["c0_actions"]))
    pix = np.stack([np.arange(n) * 9.0, np.arange(n) + 30.0, 90.0 - np.arange(n), np.zeros(n)], 1).astype(np.float32)
    s.SEG_PIXELS.copy_(_dev(pix))
    s.compute_observations()
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.TVALUE.cpu().numpy(), f["tvalue"], rtol=2e-4, atol=2e-4)        # fp32 MFMA chain over K = 650 / 1024 / 512
    # the frame appended by the same call
    obs = s.OBS.cpu().numpy()
    root = g3["c0_root"].reshape(n, 142, 13)
    want = T.search_tvalue_buffer_update(f["x"], obs[:, :62], np.zeros((n, 4), np.float32), pix[:, 1], pix[:, 2], pix[:, 0])
    got = s.TVALUE_OBS.cpu().numpy()
    assert not got[:, 650:].any()
    mask = np.ones(650, bool); mask[585 + 26:585 + 30] = False                                   # the camera-frame quaternion is checked below
    np.testing.assert_allclose(got[:, :650][:, mask], want[:, mask], rtol=1e-6, atol=1e-6)
    q = got[:, 585 + 26:585 + 30]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-4)                        # a unit quaternion: camera_view_segmentation_target_rot
    # a second step evaluates the network on the shifted buffer
    s.compute_observations()
    torch.cuda.synchronize()
    _, tv2 = T.retri_tvalue_forward(got[:, :650], T.retri_tvalue_formula_weights())
    np.testing.assert_allclose(s.TVALUE.cpu().numpy(), tv2, rtol=2e-4, atol=2e-4)
