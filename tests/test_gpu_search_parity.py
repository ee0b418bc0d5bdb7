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
