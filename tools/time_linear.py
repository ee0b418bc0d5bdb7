"""one trunk layer of the rollout at a time (sdxpk_linear2: actor + central value in one launch), operands left untouched between launches
(whatever the caches keep of them stays): us per launch.  Beside tools/time_act.py (where every layer reads what the previous launch
has just written) this separates the memory system from the kernel's own pipeline.   usage: python tools/time_linear.py [M] [shape ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd import _abi  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _abi.load_library()
lib.sdxpk_linear2.restype = None
lib.sdxpk_linear2.argtypes = ([C.c_void_p] * 4 + [C.c_int] * 2 + [C.c_void_p] * 2) * 2 + [C.c_int, C.c_int, C.c_void_p]
lib.sdxpk_linear_force_shape.argtypes = [C.c_int]
layers = [("L0 396/564 -> 1024", 1024, 396, 564), ("L1 1024 -> 512", 512, 1024, 1024), ("L2 512 -> 256", 256, 512, 512)]
for shape in ([int(a) for a in sys.argv[2:]] or [3, 1, 4]):
    lib.sdxpk_linear_force_shape(shape)
    tot = 0.0
    for name, n, k0, k1 in layers:
        x0, x1 = torch.randn(m, k0, device="cuda"), torch.randn(m, k1, device="cuda")
        w0, w1 = torch.randn(n, k0, device="cuda") / 30, torch.randn(n, k1, device="cuda") / 30
        b0, b1 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        y0, y1 = torch.empty(m, n, device="cuda"), torch.empty(m, n, device="cuda")
        call = lambda: lib.sdxpk_linear2(x0.data_ptr(), w0.data_ptr(), b0.data_ptr(), y0.data_ptr(), n, k0, None, None,
                                         x1.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), n, k1, None, None, m, 1, None)
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        tot += us
        fl = 2.0 * m * n * (k0 + k1)
        print("shape %d  %-20s %6.1f us per launch (back to back, launch gap included)  %5.1f TFLOP/s" % (shape, name, us, fl / us * 1e-6), flush=True)
    print("shape %d  three layers: %.1f us" % (shape, tot), flush=True)
lib.sdxpk_linear_force_shape(0)
