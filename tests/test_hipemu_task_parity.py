"""CPU (-m "not gpu"): the golden-vector tests of the task kernels (tests/test_gpu_*_parity.py) run a second time against the EMULATED
simulator - the product's sdx_capi / sdx_task / sdx_physics / sdx_camera SOURCES compiled by g++ for the SIMT emulator of
tests/hipemu and driven through the same C ABI.  The test bodies are the GPU tests' own functions; only the fixtures differ (an EmuSim
instead of an SdxSim, `.cuda()` as the identity).  This pins the kernels' logic to the reference's golden vectors without a GPU; the
`-m gpu` runs remain the check of the compiled gfx950 code."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import test_gpu_insert_parity as GI   # noqa: E402
import test_gpu_orient_parity as GO   # noqa: E402
import test_gpu_search_parity as GS   # noqa: E402
import test_gpu_task_parity as GT     # noqa: E402
from tests.hipemu.sim import EmuSim   # noqa: E402


@pytest.fixture(autouse=True)
def host_memory_is_the_device(monkeypatch):
    import seqdex_amd.sim as S
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: self.detach().clone())   # a device-to-host copy is a snapshot
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(S, "SdxSim", lambda n, device=None, **kw: EmuSim(n, **kw))


def _sim(**kw):
    s = EmuSim(16, seed=22, **kw)
    yield s
    s.close()


@pytest.fixture(scope="module")
def sim16():
    yield from _sim()


@pytest.fixture(scope="module")
def orient16():
    yield from _sim(task_kind=1)


@pytest.fixture(scope="module")
def insert16():
    yield from _sim(task_kind=2, max_episode_length=125.0)


@pytest.fixture(scope="module")
def search16():
    yield from _sim(task_kind=3, max_episode_length=75.0, act_moving_average=0.6, target_euler=[0.0, 3.14, 1.57])


# ---------------------------------------------------------------- BlockAssemblyGraspSim (F2, F3, F5, F8)
@pytest.mark.parametrize("phase", [0, 1, 2, 3])
def test_pre_physics_golden(sim16, golden_dir, phase):
    GT.test_pre_physics_golden(sim16, golden_dir, phase)


def test_observations_golden(sim16, golden_dir, scene):
    GT.test_observations_golden(sim16, golden_dir, scene)


def test_reward_golden(golden_dir):
    GT.test_reward_golden(golden_dir)


def test_reset_idx_golden(sim16, golden_dir):
    GT.test_reset_idx_golden(sim16, golden_dir)


# ---------------------------------------------------------------- BlockAssemblyOrient (O2, O3, O5)
@pytest.mark.parametrize("phase", [0, 1])
def test_orient_pre_physics_golden(orient16, golden_dir, scene, phase):
    GO.test_orient_pre_physics_golden(orient16, golden_dir, scene, phase)


def test_orient_observations_golden(orient16, golden_dir):
    GO.test_orient_observations_golden(orient16, golden_dir)


def test_orient_reward_golden(golden_dir):
    GO.test_orient_reward_golden(golden_dir)


# ---------------------------------------------------------------- BlockAssemblyInsertSim (I2, I3, I5)
def test_insert_pre_physics_golden(insert16, golden_dir):
    GI.test_insert_pre_physics_golden(insert16, golden_dir)


def test_insert_observations_golden(insert16, golden_dir):
    GI.test_insert_observations_golden(insert16, golden_dir)


def test_insert_reward_golden(golden_dir):
    GI.test_insert_reward_golden(golden_dir)


# ---------------------------------------------------------------- BlockAssemblySearch (S2, S3, S5, S7) and its camera
def test_search_pre_physics_golden(search16, golden_dir, scene):
    GS.test_search_pre_physics_golden(search16, golden_dir, scene)


def test_search_observation_and_state_layout(search16, golden_dir, scene):
    GS.test_search_observation_and_state_layout(search16, golden_dir, scene)


def test_search_reward_golden(golden_dir):
    GS.test_search_reward_golden(golden_dir)


def test_search_retri_tvalue_and_temporal_buffer(search16, golden_dir, scene):
    GS.test_search_retri_tvalue_and_temporal_buffer(search16, golden_dir, scene)      # the MLP itself is the driver's plain loop here


# ---------------------------------------------------------------- larger / composite cases of the GPU suites that the emulator can afford
def test_task_kernels_vs_oracle_1024(scene):
    GT.test_task_kernels_vs_oracle_1024(scene)


def test_terminal_state_harvesting(scene):
    GT.test_terminal_state_harvesting(scene)


def test_segmentation_camera_matches_numpy_ray_caster(scene):
    GS.test_segmentation_camera_matches_numpy_ray_caster(scene)


# ---------------------------------------------------------------- the full-size observation checks of the other three tasks (numpy oracle)
def test_orient_1024_observations_against_oracle(scene):
    import test_gpu_fullsize_tasks as GF
    GF.test_orient_1024_observations_against_oracle(scene)


def test_insert_2048_observations_against_oracle(scene):
    import test_gpu_fullsize_tasks as GF
    GF.test_insert_2048_observations_against_oracle(scene)


def test_search_128_observations_against_oracle(scene):
    import test_gpu_fullsize_tasks as GF
    GF.test_search_128_observations_against_oracle(scene)


def test_scripted_grasp_controller_kernel_against_its_numpy_statement(scene):
    """k_scripted_grasp (csrc/sdx_task.hip), the stand-in grasp policy of the chain benchmark: the kernel SOURCE on the emulator against the
    formulas it stands for - reach above the target brick, hold the wrist at the prepare orientation, descend, close the fingers from the
    step the hand arrived at (at the latest 58), then stop following the brick and raise the hand - over random hand / brick states and
    every phase of an episode."""
    import ctypes as C
    n = 64
    s = EmuSim(n, seed=3)
    try:
        rng = np.random.default_rng(0)
        rb = s.RB.view(n, -1, 13)
        root = s.ROOT.view(n, 142, 13)
        hbi = scene.hand_base_body
        seg = np.array([scene.seg_index(e) for e in range(n)])
        for e in range(n):                                        # hand base within a few centimetres of the pinch pose above the brick, or far away
            br = rng.uniform([0.1, 0.0, 0.55], [0.4, 0.4, 0.7]).astype(np.float32)
            off = np.array([-0.12, -0.025, 0.19], np.float32) if e % 3 == 0 else rng.uniform(-0.2, 0.3, 3).astype(np.float32)
            jit = rng.uniform(-0.01, 0.01, 3).astype(np.float32) if e % 6 == 0 else np.zeros(3, np.float32)
            root[e, seg[e], 0:3] = torch.from_numpy(br)
            rb[e, hbi, 0:3] = torch.from_numpy(br + off + jit)
            q = rng.standard_normal(4).astype(np.float32)
            q = q / np.linalg.norm(q) if e % 2 else np.array([0.7107, -0.7033, 0.0113, -0.0091], np.float32)
            rb[e, hbi, 3:7] = torch.from_numpy(q)
        prog = rng.integers(0, 80, n).astype(np.int32)
        prog[:4] = [0, 1, 58, 70]
        s.PROGRESS.copy_(torch.from_numpy(prog).to(s.PROGRESS.dtype))
        st0 = np.stack([np.where(rng.random(n) < 0.5, 1e9, rng.integers(10, 60, n)), np.where(rng.random(n) < 0.7, 1e9, rng.integers(30, 70, n))], 1)
        par = np.array([0.85, 12.0, 0.04, 0.19, 0.12, 0.025, 0.0, 0.0])
        close = torch.from_numpy(np.concatenate([st0.reshape(-1), par]).astype(np.float32))
        close0 = st0.astype(np.float32)
        act = torch.zeros(n, 23)
        s.lib.sdxk_scripted_grasp_actions.restype = C.c_int
        s.lib.sdxk_scripted_grasp_actions.argtypes = [C.c_void_p] * 4
        assert s.lib.sdxk_scripted_grasp_actions(s.h, C.c_void_p(close.data_ptr()), C.c_void_p(act.data_ptr()), None) == 0
        hb = rb[:, hbi].numpy().astype(np.float64)
        br = np.stack([root[e, seg[e]].numpy() for e in range(n)]).astype(np.float64)
        p = prog.astype(np.float64)
        cl = np.where(p < 2, 1e9, close0[:, 0].astype(np.float64))
        hold = np.where(p < 2, 1e9, close0[:, 1].astype(np.float64))
        r = hb[:, 0:3] - br[:, 0:3]
        horiz = np.hypot(r[:, 0] + par[4], r[:, 1] + par[5])
        above = np.where(horiz > 0.05, 0.25, par[3])
        want = np.zeros((n, 23))
        want[:, 0] = np.clip(2.5 * (br[:, 0] - par[4] - hb[:, 0]) / 0.64, -1, 1)
        want[:, 1] = np.clip(2.5 * (br[:, 1] - par[5] - hb[:, 1]) / 0.64, -1, 1)
        want[:, 2] = np.clip(2.5 * (br[:, 2] + above - hb[:, 2]) / 0.64, -1, 1)
        q0 = np.array([0.7107, -0.7033, 0.0113, -0.0091])
        q = hb[:, 3:7]
        rw = q0[3] * q[:, 3] + q[:, 0:3] @ q0[0:3]
        cr = np.cross(np.broadcast_to(q0[0:3], (n, 3)), q[:, 0:3])
        sg = np.sign(rw)
        want[:, 3:6] = np.clip(2.0 * (-q0[3] * q[:, 0:3] + q[:, 3:4] * q0[0:3] - cr) * sg[:, None] / 0.2, -1, 1)
        arrived = (horiz < 0.012) & (np.abs(r[:, 2] - par[3]) < 0.012)
        cl = np.where(arrived | (p >= 58), np.minimum(cl, p), cl)
        frac = np.clip(0.3 + (p - cl) / par[1] * (par[0] - 0.3), 0.3, par[0])
        hold = np.where(p >= cl + par[1], np.minimum(hold, p), hold)
        want[p >= hold, 0:3] = [0.0, 0.0, par[2]]
        want[:, 7:] = (2.0 * frac - 1.0)[:, None]
        want[:, [7, 11, 15]] = 0.0
        np.testing.assert_allclose(act.numpy(), want, rtol=2e-5, atol=2e-5)
        out = close.numpy()[:2 * n].reshape(n, 2)
        np.testing.assert_allclose(out[:, 0], cl.astype(np.float32), rtol=0, atol=0)
        np.testing.assert_allclose(out[:, 1], hold.astype(np.float32), rtol=0, atol=0)
        assert arrived.sum() >= 4 and (~arrived).sum() >= 20 and (cl < 1e8).sum() >= 10          # the cases the sample is meant to hold
        assert (p >= hold).sum() >= 8 and (p < hold).sum() >= 20
    finally:
        s.close()
