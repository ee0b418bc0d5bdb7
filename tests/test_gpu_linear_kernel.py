"""k_linear_mfma (the rollout's trunk layers and the T-value trainer's forward; seqdex_amd/csrc/sdxp_kernels.hip) against torch's float64
linear + ELU, for every tile shape the launcher can pick, ragged sizes and on-the-fly input normalisation (the reference normalises the
central-value input with its running mean/std before the trunk: rl_games central_value.py).  The launcher's default below
2048 rows is shape 3 (64 x 64 tiles, every chunk split over two groups of four waves), 2 beyond.  Shapes 1 and 6 (one k group) accumulate every
output in the same order, so they must agree bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [1, 2, 3, 4, 5, 6, 7]      # 7: k_linear_glds (operands staged by LDS-DMA)
CASES = [(1024, 1024, 396), (1024, 1024, 564), (1024, 512, 1024), (1024, 256, 512), (100, 2, 128), (65, 70, 36), (3, 1, 4), (130, 129, 68), (4096, 512, 1024)]


def _lib():
    from seqdex_amd import _abi
    lib = _abi.load_library()
    lib.sdxpk_linear.restype = None
    lib.sdxpk_linear.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p] * 3
    lib.sdxpk_linear2.restype = None
    lib.sdxpk_linear2.argtypes = ([C.c_void_p] * 4 + [C.c_int] * 2 + [C.c_void_p] * 2) * 2 + [C.c_int, C.c_int, C.c_void_p]
    lib.sdxpk_linear_force_shape.restype = None
    lib.sdxpk_linear_force_shape.argtypes = [C.c_int]
    return lib


def _reference(x, w, b, elu, mean=None, var=None):
    x = x.double()
    if mean is not None:
        x = ((x.float() - mean.float()) / torch.sqrt(var.float() + 1e-5)).clamp(-5.0, 5.0).double()
    y = x @ w.double().t() + b.double()
    return torch.nn.functional.elu(y) if elu else y


@pytest.mark.parametrize("norm", [False, True])
def test_every_tile_shape_matches_float64(norm):
    lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    try:
        for (m, n, k) in CASES:
            x = torch.randn(m, k, device="cuda", generator=g)
            w = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
            b = torch.randn(n, device="cuda", generator=g)
            mean = torch.randn(k, device="cuda", generator=g, dtype=torch.float64) if norm else None
            var = (torch.rand(k, device="cuda", generator=g, dtype=torch.float64) + 0.1) if norm else None
            ref = _reference(x, w, b, True, mean, var)
            outs = {}
            for shape in SHAPES:
                lib.sdxpk_linear_force_shape(shape)
                y = torch.full((m, n), float("nan"), device="cuda")
                lib.sdxpk_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k, 1,
                                 mean.data_ptr() if norm else None, var.data_ptr() if norm else None, None)
                torch.cuda.synchronize()
                np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=2e-5, err_msg="shape %d, %s" % (shape, (m, n, k)))
                outs[shape] = y
            assert torch.equal(outs[1], outs[6]) and torch.equal(outs[1], outs[2])        # same accumulation order
    finally:
        lib.sdxpk_linear_force_shape(0)


def test_two_products_in_one_launch():
    lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(6)
    m = 1024
    x0, x1 = torch.randn(m, 396, device="cuda", generator=g), torch.randn(m, 564, device="cuda", generator=g)
    w0, w1 = torch.randn(1024, 396, device="cuda", generator=g) / 20, torch.randn(512, 564, device="cuda", generator=g) / 24
    b0, b1 = torch.randn(1024, device="cuda", generator=g), torch.randn(512, device="cuda", generator=g)
    mean = torch.randn(564, device="cuda", generator=g, dtype=torch.float64)
    var = torch.rand(564, device="cuda", generator=g, dtype=torch.float64) + 0.1
    try:
        for shape in SHAPES:
            lib.sdxpk_linear_force_shape(shape)
            y0, y1 = torch.empty(m, 1024, device="cuda"), torch.empty(m, 512, device="cuda")
            lib.sdxpk_linear2(x0.data_ptr(), w0.data_ptr(), b0.data_ptr(), y0.data_ptr(), 1024, 396, None, None,
                              x1.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), 512, 564, mean.data_ptr(), var.data_ptr(), m, 1, None)
            torch.cuda.synchronize()
            np.testing.assert_allclose(y0.cpu().numpy(), _reference(x0, w0, b0, True).float().cpu().numpy(), rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(y1.cpu().numpy(), _reference(x1, w1, b1, True, mean, var).float().cpu().numpy(), rtol=2e-5, atol=2e-5)
    finally:
        lib.sdxpk_linear_force_shape(0)


def test_create_refuses_input_widths_the_rollout_kernels_cannot_take():
    """sdxp_create: the first trunk layer normalises through per-column tables of 1 024 entries and the dataset copy of k_act_heads moves a row
    as at most 256 16-byte pieces; widths outside [4, 1024] are refused with a message instead of being truncated"""
    from seqdex_amd.ppo import SdxPPO, make_config
    from seqdex_amd.sim import SdxError
    for kw in ({"state_dim": 1028}, {"obs_dim": 1028}):
        with pytest.raises(SdxError, match="above 1024"):
            SdxPPO(16, config=make_config(16, **kw))
