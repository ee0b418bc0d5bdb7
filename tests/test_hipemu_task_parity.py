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
