"""-m gpu: BlockAssemblyOrient (BASELINE.json configs[2]) per-step tensor code on the HIP path (scene.task_kind = 1), called through
the C ABI, against the golden vectors captured from the reference's own Orient module (tests/golden/O*.npz) and the numpy oracle."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import task_oracle as T  # noqa: E402


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def orient16():
    from seqdex_amd.sim import SdxSim
    s = SdxSim(16, device="cuda:0", seed=22, task_kind=1)
    yield s
    s.close()


@pytest.mark.parametrize("phase", [0, 1])
def test_orient_pre_physics_golden(orient16, golden_dir, scene, phase):
    f = np.load(os.path.join(golden_dir, "O2_pre_physics.npz"))
    p = "p%d_" % phase
    s, n = orient16, 16
    s.RESET.zero_()
    dof = torch.zeros(n, 23, 2)
    dof[:, :, 0] = torch.as_tensor(f[p + "q"])
    s.DOF.copy_(dof.view(-1, 2).cuda())
    s.PREV_TARGETS.copy_(_dev(f[p + "prev_targets"]))
    s.PROGRESS.copy_(_dev(f[p + "progress"]))
    s.INIT_POS.copy_(_dev(f[p + "init_pos"]))
    s.RB[:, 7, 0:3] = _dev(f[p + "hand_pos"])
    s.RB[:, 7, 3:7] = _dev(f[p + "hand_rot"])
    root = s.ROOT.view(n, 142, 13)
    for e in range(n):
        root[e, scene.seg_index(e), 0:3] = _dev(f[p + "target_pos"][e])
    s.JAC_EEF.copy_(_dev(f[p + "J"]))
    s.pre_physics(_dev(f[p + "actions"]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), f[p + "cur_targets"], rtol=2e-4, atol=1e-4)      # the reference's numbers
    np.testing.assert_array_equal(s.TARGETS.cpu().numpy(), s.PREV_TARGETS.cpu().numpy())
    want = T.orient_pre_physics_targets(f[p + "actions"], f[p + "q"], f[p + "prev_targets"], f[p + "progress"], f[p + "init_pos"],
                                        f[p + "hand_pos"], f[p + "hand_rot"], f[p + "target_pos"], f[p + "J"], f["lower"], f["upper"])
    np.testing.assert_allclose(s.TARGETS.cpu().numpy(), want, rtol=2e-4, atol=1e-4)                       # and the oracle's


def test_orient_observations_golden(orient16, golden_dir):
    f = np.load(os.path.join(golden_dir, "O3_observations.npz"))
    s = orient16
    assert tuple(s.OBS.shape) == (16, 186) and tuple(s.STATES.shape) == (16, 564)        # 62 x 3 / 188 x 3 (OR:191-207)
    s.set_tvalue_weights({k[3:]: f[k] for k in f.files if k.startswith("tv_")})
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
        s.compute_observations()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.OBS.cpu().numpy(), f[p + "obs_buf"], rtol=3e-5, atol=3e-5)     # 62 numbers, never stacked
        np.testing.assert_allclose(s.STATES.cpu().numpy(), f[p + "states_buf"], rtol=3e-5, atol=3e-5)
        np.testing.assert_array_equal(s.OBS_CLAMPED.cpu().numpy(), np.clip(s.OBS.cpu().numpy(), -5, 5))
        np.testing.assert_allclose(s.FINGER_DIST.cpu().numpy(), f[p + "finger_dist"], rtol=3e-5, atol=3e-5)
        np.testing.assert_array_equal(s.TVALUE.cpu().numpy(), f[p + "tvalue"])                     # gated at 0.99 (OR:1203)


def test_orient_reward_golden(golden_dir):
    from seqdex_amd.sim import SdxSim
    f = np.load(os.path.join(golden_dir, "O5_reward.npz"))
    m = f["progress"].shape[0]
    s = SdxSim(m, device="cuda:0", task_kind=1, max_episode_length=float(f["max_episode_length"]))
    try:
        root = s.ROOT.view(m, 142, 13)
        seg = torch.tensor([s.scene.seg_index(i) for i in range(m)]).cuda()
        ar = torch.arange(m).cuda()
        root[ar, seg, 0:3] = _dev(f["target_pos"])
        root[ar, seg, 3:7] = _dev(f["target_rot"])
        for body, key in zip(s.scene.fingertip_bodies, ["ff", "mf", "rf", "th"]):
            s.RB[:, body, 0:3] = _dev(f[key] - np.array([0, 0, 0.04], np.float32))
            s.RB[:, body, 3:7] = torch.tensor([0.0, 0, 0, 1]).cuda()
        s.PROGRESS.copy_(_dev(f["progress"] - 1))        # post_physics_step increments first
        s.RESET.copy_(_dev(f["reset_buf"]))
        s.SUCCESSES.copy_(_dev(f["successes"]))
        s.CONS_SUCCESSES.copy_(_dev(f["cons_in"]))
        s.post_physics()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s.REW.cpu().numpy(), f["reward"], rtol=3e-5, atol=3e-6)
        np.testing.assert_array_equal(s.RESET.cpu().numpy(), f["resets"])
        np.testing.assert_allclose(s.CONS_SUCCESSES.cpu().numpy(), f["cons_out"], rtol=1e-6)
    finally:
        s.close()


def test_orient_task_end_to_end_with_scripted_reset(scene):
    """BlockAssemblyOrient through the VecTask surface for two episodes (episodeLength 75): every env times out together, the reset
    event runs its scripted pre-grasp (50 + 2 + 1 + 50 simulator steps with the tracking IK), and afterwards the hand base hovers
    0.22 above / 0.18 behind the target brick's initial position with the fixed wrist orientation (OR:1655-1695)."""
    import yaml
    from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient
    from seqdex_amd.vec_task_rlgames import RLgamesVecTaskPython
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_orient.yaml")))
    n = 16
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblyOrient(cfg, device_type="cuda", device_id=0, headless=True, seed=3, piles_per_type=2)
    env = RLgamesVecTaskPython(task, "cuda:0")
    obs = env.reset()
    assert tuple(obs["obs"].shape) == (n, 186) and tuple(obs["states"].shape) == (n, 564)
    g = torch.Generator().manual_seed(0)
    resets_seen = 0
    for t in range(160):
        a = (torch.rand(n, 23, generator=g) * 2 - 1).cuda() * 0.3
        obs, rew, reset, _ = env.step(a)
        resets_seen += int(reset.sum())
        if t == 0:      # the step right after the reset event of env.reset()/first step: the pre-grasp has just finished
            torch.cuda.synchronize()
            hb = task.sim.RB[:, 7, 0:3].cpu().numpy()
            ip = task.sim.INIT_POS.cpu().numpy()
            want = ip + np.array([-0.18, 0.0, 0.22], np.float32)
            err = hb - want            # x, y converge; z stops a few cm high where joint 4 reaches its limit (arm fully folded)
            assert np.abs(err[:, :2]).max() < 0.05 and -0.02 < err[:, 2].min() and err[:, 2].max() < 0.12, err
    torch.cuda.synchronize()
    assert resets_seen == 2 * n                                  # two time-outs per env in 160 steps of 75-step episodes
    assert np.isfinite(obs["obs"].cpu().numpy()).all() and np.isfinite(rew.cpu().numpy()).all()
    assert (obs["obs"][:, 62:] == 0).all()                       # the unstacked tail of the observation stays zero
    r = task.sim.ROOT.cpu().numpy().reshape(n, 142, 13)
    assert r[:, 9:81, 2].min() > 0.55 and np.isfinite(r).all()
    assert 0.0 < float(rew.mean()) <= 1.0                        # exp(-...) reward


def test_pile_ring_takes_the_reference_length_from_the_environment():
    """OR:1485 keeps 10 000 piles per brick-type group; the default ring here is SDX_PILE_HARVEST_SLOTS = 512, SDX_PILE_SLOTS in the
    environment of sdx_create selects the length (549 MB at 10 000).  The ring and its key tensor follow, appends still land in it."""
    import yaml
    from seqdex_amd import _abi
    from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_orient.yaml")))
    n = 16
    cfg["env"]["numEnvs"] = n
    os.environ["SDX_PILE_SLOTS"] = "10000"
    try:
        task = BlockAssemblyOrient(cfg, device_type="cuda", device_id=0, headless=True, seed=4, piles_per_type=2)
    finally:
        del os.environ["SDX_PILE_SLOTS"]
    assert tuple(task.sim.PILE_HARVEST.shape) == (8, 10000, 132, 13) and tuple(task.sim.PILE_HARVEST_KEYS.shape) == (8, 10000)
    flat = np.zeros(_abi.TV_PARAMS, np.float32)
    flat[-1] = 20.0                                   # the 0.99 gate accepts every state
    task.sim.set_tvalue_weights(flat)
    g = torch.Generator().manual_seed(0)
    for t in range(77):
        task.step(((torch.rand(n, 23, generator=g) * 2 - 1) * 0.1).cuda())
    torch.cuda.synchronize()
    pc = task.sim.PILE_HARVEST_COUNT.cpu().numpy()
    assert pc.sum() >= n // 2
    piles = task.pile_terminal_states()
    assert piles is not None and piles.shape[1] == int(pc.min()) and np.isfinite(piles.cpu().numpy()).all()


def test_orient_harvests_pile_states_for_grasp_sim(scene):
    """OR:1463-1488: at a reset event (after the first one) every env whose episode ended with the hand withdrawn, the target brick in
    the bin half and an accepting T-value stores its whole brick pile in the ring of its brick-type group and logs a T-value success;
    the others log failures.  The harvested piles are what BlockAssemblyGraspSim starts from (GS:412-413)."""
    import yaml
    from seqdex_amd import _abi
    from seqdex_amd.tasks.block_assembly_grasp_sim import BlockAssemblyGraspSim
    from seqdex_amd.tasks.block_assembly_orient import BlockAssemblyOrient
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_orient.yaml")))
    n = 16
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblyOrient(cfg, device_type="cuda", device_id=0, headless=True, seed=4, piles_per_type=2)
    flat = np.zeros(_abi.TV_PARAMS, np.float32)
    flat[-1] = 20.0                                   # output_layer.bias[1]: sigmoid(elu(20)) = 1 -> the 0.99 gate accepts every state
    task.sim.set_tvalue_weights(flat)
    assert task.pile_terminal_states() is None
    g = torch.Generator().manual_seed(0)
    for t in range(77):                               # episode length 75: one time-out of all envs, its reset event runs in step 76
        task.step(((torch.rand(n, 23, generator=g) * 2 - 1) * 0.1).cuda())
    torch.cuda.synchronize()
    tvc = task.sim.TV_COUNT.cpu().numpy()
    pc = task.sim.PILE_HARVEST_COUNT.cpu().numpy()
    assert tvc.sum() == n and pc.sum() == tvc[0] and tvc[0] >= n // 2, (tvc, pc)
    piles = task.pile_terminal_states()
    assert piles is not None and tuple(piles.shape[2:]) == (132, 13)
    p = piles.cpu().numpy()
    assert np.isfinite(p).all()
    for grp in range(8):
        seg = scene.seg_index(grp) - 9
        y = p[grp, :, seg, 1]
        assert ((y > 0) & (y < 0.5)).all()            # the harvest rule on the target brick of that group
    np.testing.assert_allclose(np.linalg.norm(p[..., 3:7], axis=-1), 1.0, atol=1e-4)
    # hand-off: a GraspSim instance starts its episodes from these piles
    gcfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_grasp_sim.yaml")))
    gcfg["env"]["numEnvs"] = n
    gs = BlockAssemblyGraspSim(gcfg, device_type="cuda", device_id=0, headless=True, seed=1, initial_piles=piles)
    gs.step(torch.zeros(n, 23).cuda())
    torch.cuda.synchronize()
    r = gs.sim.ROOT.view(n, 142, 13).cpu().numpy()
    ch = gs.sim.PILE_CHOICE.cpu().numpy()
    for e in range(n):                                # positions after one simulator step from the chosen harvested pile
        assert np.abs(r[e, 9:81, 0:3] - p[e % 8, ch[e], 0:72, 0:3]).max() < 0.02
