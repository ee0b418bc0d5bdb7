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
