"""Where does k_gemm_nt differ from torch?  One product per (dtype, epilogue, tile shape); G and row-sum errors separately, the worst
element's position and the error summed per 32 x 32 output block (SDXP_NT_TILE must be set by the caller: the launcher reads it once)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from seqdex_amd import _abi  # noqa: E402
from time_gemm_nt import EPI_FWD, EPI_NN, EPI_TN, NtArgs  # noqa: E402

lib = _abi.load_library()
lib.sdxpk_gemm_nt_launch.restype = C.c_int
lib.sdxpk_gemm_nt_launch.argtypes = [C.c_int, C.c_int, C.POINTER(NtArgs), C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)


def blocks(err, bm=32, bn=32):
    M, N = err.shape
    e = torch.nn.functional.pad(err, (0, (-N) % bn, 0, (-M) % bm))
    return e.reshape(e.shape[0] // bm, bm, e.shape[1] // bn, bn).amax(dim=(1, 3))


for bf in (0, 1):
    dt = torch.bfloat16 if bf else torch.float32
    KC = 64 if bf else 32
    for (M, N, K, S) in ((256, 512, 8192, 16), (128, 64, 64, 1), (256, 512, 256, 1), (1024, 396, 8192, 16)):
        A = (rnd(M, K) * 0.2).to(dt).contiguous(); B = rnd(N, K).to(dt).contiguous()
        kc = ((K + S - 1) // S + KC - 1) // KC * KC
        pz = M * N + M
        part = torch.full((S, pz), float("nan"), device=dev)
        a = NtArgs(A.data_ptr(), K, B.data_ptr(), K, M, N, K, kc, part.data_ptr(), N, pz, None, 0, None, 0, None, None, 0, None, 0, part.data_ptr() + 4 * M * N)
        arr = (NtArgs * 3)(a, a, a)
        assert lib.sdxpk_gemm_nt_launch(bf, EPI_TN, arr, 1, S, None) == 0
        torch.cuda.synchronize()
        tot = part.double().sum(0)
        G, rs = tot[:M * N].reshape(M, N), tot[M * N:]
        ref, rref = A.double() @ B.double().t(), A.double().sum(1)
        eg, er = (G - ref).abs(), (rs - rref).abs()
        print("TN bf%d %dx%dx%d S%d: G max err %.3e (rel %.2e) at %s; rowsum max err %.3e (rel %.2e) at row %d; nan %d" % (
            bf, M, N, K, S, float(eg.max()), float(eg.max() / ref.abs().max()), divmod(int(eg.argmax()), N), float(er.max()), float(er.max() / rref.abs().max()),
            int(er.argmax()), int(torch.isnan(part).sum())), flush=True)
        if float(eg.max() / ref.abs().max()) > 1e-3 * (10 if bf else 1):
            print("  G error per 32x32 block (rows = output rows / 32):\n", (blocks(eg) / ref.abs().max()).cpu().numpy().round(3))
        if float(er.max() / rref.abs().max()) > 1e-4:
            print("  rowsum error by row:", (er / rref.abs().max()).cpu().numpy().round(3).tolist()[:256])
    for (M, N, K) in ((8192, 256, 512), (64, 128, 64), (300, 100, 96 if not bf else 128)):
        A = rnd(M, K).to(dt).contiguous(); B = (rnd(N, K) / K ** 0.5).to(dt).contiguous(); bias = rnd(N)
        Cf = torch.full((M, N), float("nan"), device=dev); Ct = torch.zeros(N, (M + 63) // 64 * 64, device=dev, dtype=dt)
        a = NtArgs(A.data_ptr(), K, B.data_ptr(), K, M, N, K, K, Cf.data_ptr(), N, 0, None, 0, Ct.data_ptr(), Ct.shape[1], bias.data_ptr(), None, 0, None, 0, None)
        arr = (NtArgs * 3)(a, a, a)
        assert lib.sdxpk_gemm_nt_launch(bf, EPI_FWD, arr, 1, 1, None) == 0
        torch.cuda.synchronize()
        ref = torch.nn.functional.elu(A.double() @ B.double().t() + bias.double())
        e = (Cf.double() - ref).abs()
        et = (Ct.double().t()[:M] - ref).abs()
        print("FWD bf%d %dx%dx%d: max err %.3e at %s, transposed copy %.3e, nan %d" % (bf, M, N, K, float(e.max()), divmod(int(e.argmax()), N), float(et.max()),
                                                                                  int(torch.isnan(Cf).sum())), flush=True)
        if float(e.max()) > 1e-3:
            print("  error per 32x32 block:\n", blocks(e).cpu().numpy().round(3)[:16])
