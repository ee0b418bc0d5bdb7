"""CPU: the SOURCE of the large-minibatch trunk products (seqdex_amd/csrc/sdx_gemm_nt.h: k_gemm_nt, k_stage) executed on the SIMT
emulator (tests/hipemu: MFMA as a wave collective with the ISA's lane layouts, global_load_lds as the lane-linear copy it is) against
numpy: source-side swizzle vs fragment reads, accumulator layout, the three epilogues incl. the transposed copies and the ones-MFMA
row sums, split reductions, ragged edges, both element types.  What the emulator cannot show - bank conflicts, timing - is measured on
the GPU (profiles/r4_bigmb_*); the GPU parity tests (tests/test_gpu_ppo_parity.py::test_large_minibatch_*) hold the assembled step to
the autograd oracle."""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="needs ROCm's clang++ for the host build")

EPI_FWD, EPI_NN, EPI_TN = 1, 3, 4


@pytest.fixture(scope="module")
def lib():
    from tests import hipemu
    l = C.CDLL(hipemu.build_gemm())
    vp, i, fp = C.c_void_p, C.c_int, C.c_void_p
    l.emu_gemm_nt.argtypes = [i, i, vp, i, vp, i, i, i, i, i, i, fp, i, C.c_longlong, vp, i, vp, i, fp, vp, i, vp, i, fp]
    l.emu_gemm_nt_narrow.argtypes = [i, vp, i, vp, i, i, i, i, fp, i, vp, i, fp]
    l.emu_stage.argtypes = [i, fp, i, i, i, i, vp, i, vp, i]
    l.emu_gemm_nt_wide.argtypes = [i, i, vp, i, vp, i, i, i, i, fp, i, vp, i, vp, i, fp, vp, i, vp, i]
    l.emu_gemm_tt.argtypes = [vp, i, vp, i, i, i, i, i, i, fp, i, C.c_longlong, vp, i]
    return l


def _elems(x, bf):
    """(array handed to the kernel, float64 values it represents)"""
    t = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
    if bf:
        b = t.to(torch.bfloat16)
        return b.view(torch.int16).numpy().copy(), b.to(torch.float64).numpy()
    return t.numpy().copy(), t.to(torch.float64).numpy()


def _from_elems(arr, bf):
    if bf:
        return torch.as_tensor(arr).view(torch.bfloat16).to(torch.float64).numpy()
    return arr.astype(np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 150, 192), (70, 36, 64)])
def test_forward_product_all_outputs(lib, bf, M, N, K):
    rng = np.random.default_rng(M + N + bf)
    KC = 64 if bf else 32
    assert K % KC == 0
    A, Av = _elems(rng.standard_normal((M, K)) * 0.5, bf)
    B, Bv = _elems(rng.standard_normal((N, K)) * 0.5, bf)
    bias = rng.standard_normal(N).astype(np.float32)
    Mp = (M + 63) // 64 * 64
    Cf = np.full((M, N), np.nan, np.float32)
    Cn = np.zeros((M, N), np.int16 if bf else np.float32)
    Ct = np.zeros((N, Mp), np.int16 if bf else np.float32)
    lib.emu_gemm_nt(bf, EPI_FWD, _p(A), K, _p(B), K, M, N, K, K, 1, _p(Cf), N, 0, _p(Cn), N, _p(Ct), Mp, _p(bias), None, 0, None, 0, None)
    want = elu(Av @ Bv.T + bias)
    np.testing.assert_allclose(Cf, want, rtol=1e-5, atol=2e-5)      # (bf16: the operands ARE bf16 values, products exact, fp32 accumulation)
    got_n, got_t = _from_elems(Cn, bf), _from_elems(Ct, bf)
    np.testing.assert_allclose(got_n, Cf.astype(np.float64), rtol=8e-3 if bf else 0, atol=0)          # the copies are the fp32 result, rounded
    np.testing.assert_allclose(got_t[:, :M], Cf.astype(np.float64).T, rtol=8e-3 if bf else 0, atol=0)
    assert not got_t[:, M:].any()                                                                     # padding columns are never written


@pytest.mark.parametrize("bf", [0, 1])
def test_narrow_tile_shape(lib, bf):
    rng = np.random.default_rng(5)
    M, N, K = 140, 100, 128
    A, Av = _elems(rng.standard_normal((M, K)), bf)
    B, Bv = _elems(rng.standard_normal((N, K)), bf)
    bias = np.zeros(N, np.float32)
    Cf = np.zeros((M, N), np.float32)
    Ct = np.zeros((N, 192), np.int16 if bf else np.float32)
    lib.emu_gemm_nt_narrow(bf, _p(A), K, _p(B), K, M, N, K, _p(Cf), N, _p(Ct), 192, _p(bias))
    np.testing.assert_allclose(Cf, elu(Av @ Bv.T), rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(_from_elems(Ct, bf)[:, :M], Cf.astype(np.float64).T, rtol=8e-3 if bf else 0)


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("epi", [EPI_FWD, EPI_NN])
@pytest.mark.parametrize("M", [270, 272])
def test_wide_tile_shape_all_copies(lib, bf, epi, M):
    """the 128 x 128 tile: two passes of the transposed epilogue (64 columns each), 19 workgroups through the XCD remap, ragged edges
    (M = 270) and whole 16-byte pieces along the rows (M = 272)"""
    rng = np.random.default_rng(17 + bf + epi)
    N, K = 300, 128
    A, Av = _elems(rng.standard_normal((M, K)) * 0.4, bf)
    B, Bv = _elems(rng.standard_normal((N, K)) * 0.4, bf)
    bias = rng.standard_normal(N).astype(np.float32)
    H, Hv = _elems(rng.standard_normal((M, N)), bf)
    Mp = 320
    Ht = np.zeros((N, Mp), H.dtype); Ht[:, :M] = H.T                         # the layer output's transposed copy, as the forward product leaves it
    Cf = np.full((M, N), np.nan, np.float32)
    Cn = np.zeros((M, N + 4), np.int16 if bf else np.float32)               # a row stride that is not a multiple of 8: scalar stores
    Ct = np.zeros((N, Mp), np.int16 if bf else np.float32)
    lib.emu_gemm_nt_wide(bf, epi, _p(A), K, _p(B), K, M, N, K, _p(Cf) if epi == EPI_FWD else None, N, _p(Cn), N + 4, _p(Ct), Mp,
                         _p(bias), _p(H), N, _p(Ht), Mp)
    want = elu(Av @ Bv.T + bias) if epi == EPI_FWD else (Av @ Bv.T) * np.where(Hv > 0, 1.0, Hv + 1.0)
    if epi == EPI_FWD:
        np.testing.assert_allclose(Cf, want, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(_from_elems(Cn, bf)[:, :N], want, rtol=8e-3 if bf else 1e-5, atol=1e-3 if bf else 2e-5)
    assert not _from_elems(Cn, bf)[:, N:].any()
    np.testing.assert_array_equal(_from_elems(Ct, bf)[:, :M], _from_elems(Cn, bf)[:, :N].T)
    assert not _from_elems(Ct, bf)[:, M:].any()


@pytest.mark.parametrize("bf", [0, 1])
def test_data_gradient_product(lib, bf):
    """dX = (dY W) * ELU'(H): A = dY [M][Nl], B = W^T [Kl][Nl]; outputs: element copy + transposed copy, no fp32 array"""
    rng = np.random.default_rng(7)
    M, Nl, Kl = 160, 128, 200
    A, Av = _elems(rng.standard_normal((M, Nl)) * 0.3, bf)
    B, Bv = _elems(rng.standard_normal((Kl, Nl)) * 0.3, bf)
    H, Hv = _elems(rng.standard_normal((M, Kl)), bf)         # the layer output in the element type of the run
    Mp = 192
    Ht = np.zeros((Kl, Mp), H.dtype); Ht[:, :M] = H.T
    Cn = np.zeros((M, Kl), np.int16 if bf else np.float32)
    Ct = np.zeros((Kl, Mp), np.int16 if bf else np.float32)
    lib.emu_gemm_nt(bf, EPI_NN, _p(A), Nl, _p(B), Nl, M, Kl, Nl, Nl, 1, None, 0, 0, _p(Cn), Kl, _p(Ct), Mp, None, _p(H), Kl, _p(Ht), Mp, None)
    want = (Av @ Bv.T) * np.where(Hv > 0, 1.0, Hv + 1.0)
    np.testing.assert_allclose(_from_elems(Cn, bf), want, rtol=8e-3 if bf else 1e-5, atol=1e-3 if bf else 2e-5)
    np.testing.assert_array_equal(_from_elems(Ct, bf)[:, :M], _from_elems(Cn, bf).T)
    assert not _from_elems(Ct, bf)[:, M:].any()


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("splits", [1, 3])
def test_weight_gradient_product_with_row_sums(lib, bf, splits):
    """G = dY^T X over the minibatch rows: A = dY^T [Nl][MBp], B = X^T [Kl][ld], split over the rows; the row sums of A (bias gradient)
    come from the ones-MFMA of the first column block"""
    rng = np.random.default_rng(11 + splits)
    KC = 64 if bf else 32
    Nl, Kl, MB = 128, 140, 3 * KC - 8                      # ragged minibatch: the last chunk is zero padded in A
    MBp = (MB + KC - 1) // KC * KC
    A0 = np.zeros((Nl, MBp)); A0[:, :MB] = rng.standard_normal((Nl, MB)) * 0.5
    ldb = MBp + 64
    B0 = rng.standard_normal((Kl, ldb))                     # what follows the window in a transposed dataset: finite, multiplied by A's zeros
    A, Av = _elems(A0, bf)
    B, Bv = _elems(B0, bf)
    kc = ((MBp + splits - 1) // splits + KC - 1) // KC * KC
    pz = Nl * Kl + Nl
    part = np.full((splits, pz), np.nan, np.float32)
    lib.emu_gemm_nt(bf, EPI_TN, _p(A), MBp, _p(B), ldb, Nl, Kl, MBp, kc, splits, _p(part), Kl, pz, None, 0, None, 0, None, None, 0, None, 0,
                    C.c_void_p(part.ctypes.data + 4 * Nl * Kl))
    assert np.isfinite(part).all()
    tot = part.astype(np.float64).sum(0)
    G, rs = tot[:Nl * Kl].reshape(Nl, Kl), tot[Nl * Kl:]
    np.testing.assert_allclose(G, Av @ Bv[:, :MBp].T, rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(rs, Av.sum(1), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("bf", [0, 1])
def test_stage_kernel(lib, bf):
    rng = np.random.default_rng(3)
    R, K, lds = 150, 100, 104
    KC = 64 if bf else 32
    Kp = (K + KC - 1) // KC * KC
    src = rng.standard_normal((R, lds)).astype(np.float32)
    ldt = 200
    dn = np.full((R, Kp), 77, np.int16 if bf else np.float32)
    dt = np.zeros((Kp, ldt), np.int16 if bf else np.float32)
    lib.emu_stage(bf, _p(src), lds, R, K, Kp, _p(dn), Kp, _p(dt), ldt)
    _, want = _elems(src[:, :K], bf)
    n, t = _from_elems(dn, bf), _from_elems(dt, bf)
    np.testing.assert_array_equal(n[:, :K], want)
    assert not n[:, K:].any()
    np.testing.assert_array_equal(t[:K, :R], want.T)
    assert not t[K:].any() and not t[:, R:].any()


@pytest.mark.parametrize("tile", [1, 2, 0])
@pytest.mark.parametrize("R,Nl,Kl,ldb,splits", [(200, 128, 100, 128, 2), (96, 256, 396, 416, 1), (328, 128, 130, 160, 3)])
def test_weight_gradient_from_row_major_operands(lib, tile, R, Nl, Kl, ldb, splits):
    """k_gemm_tt (fp32): G = dY^T X read from dY [rows][N_l] and X [rows][K_l] as the other products leave them - no transposed copies.
    Ragged rows (the last chunk comes partly from the page of zeros), columns past the operand's width clamped into it, row splits,
    both tiles, column sums of dY (the bias gradient) beside the MFMAs."""
    rng = np.random.default_rng(R + Kl + tile)
    A = (rng.standard_normal((R, Nl)) * 0.3).astype(np.float32)
    B = np.full((R, ldb), np.nan, np.float32)                       # padding columns hold garbage in the real buffers of layers > 0 ...
    B[:, :Kl] = rng.standard_normal((R, Kl)).astype(np.float32)
    B[:, Kl:] = 7.0                                                 # ... here a finite marker: products with it belong to outputs that are never stored
    kchunk = ((R + splits - 1) // splits + 31) // 32 * 32
    pz = Nl * Kl + Nl
    part = np.full((splits, pz), np.nan, np.float32)
    lib.emu_gemm_tt(_p(A), Nl, _p(B), ldb, Nl, Kl, R, kchunk, splits, _p(part), Kl, pz, C.c_void_p(part.ctypes.data + 4 * Nl * Kl), tile)
    assert np.isfinite(part).all()
    tot = part.sum(0)
    want = A.astype(np.float64).T @ B[:, :Kl].astype(np.float64)
    np.testing.assert_allclose(tot[:Nl * Kl].reshape(Nl, Kl), want, rtol=1e-5, atol=3e-5)
    np.testing.assert_allclose(tot[Nl * Kl:], A.astype(np.float64).sum(0), rtol=1e-5, atol=3e-5)
    for z in range(splits):                                         # every split covers exactly its own rows
        r0, r1 = z * kchunk, min(R, (z + 1) * kchunk)
        np.testing.assert_allclose(part[z, :Nl * Kl].reshape(Nl, Kl), A[r0:r1].astype(np.float64).T @ B[r0:r1, :Kl].astype(np.float64), rtol=1e-5, atol=3e-5)
