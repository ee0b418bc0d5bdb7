"""statistics of the teacher-forcing comparison of tests/test_gpu_physics_parity.py for whichever build of the library SDX_LIB_PATH names
(test infrastructure: uses the C oracle).  usage: python tests/helpers/parity_stats.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import physics_oracle as po  # noqa: E402
from seqdex_amd.sim import SdxSim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
state = np.load(os.path.join(ROOT, "tests", "golden", "P1_settled_state.npz"))
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
for warm in (0.0, 0.8):
    n = state["root"].shape[0]
    s = SdxSim(n, warm_start=warm)
    root, dof = state["root"].copy(), state["dof"].copy()
    ow = po.WarmState(n)
    for it in range(steps):
        s.ROOT.copy_(dev(root.reshape(-1, 13)))
        s.DOF.copy_(dev(dof.reshape(-1, 2)))
        s.TARGETS.copy_(dev(state["targets"]))
        s.simulate()
        torch.cuda.synchronize()
        g_root = s.ROOT.cpu().numpy().reshape(n, 142, 13)
        g_dof = s.DOF.cpu().numpy().reshape(n, 23, 2)
        g_nc = s.NCONTACTS.cpu().numpy()
        o_root, o_dof = root.copy(), dof.copy()
        _, _, _, o_nc = po.simulate(s._desc, o_root, o_dof, state["targets"], ow)
        dp = np.abs(g_root[:, 9:81, 0:7] - o_root[:, 9:81, 0:7]).max(-1)
        dv = np.abs(g_root[:, 9:81, 7:13] - o_root[:, 9:81, 7:13]).max(-1)
        print("warm %.1f step %d: contact counts equal in %d of %d envs (max diff %d)  brick pose: <1e-6 %.4f  <2e-5 %.4f  >=1e-4 %d  max %.2e   brick vel: <1e-4 %.4f <2e-3 %.4f max %.2e   joint pos max %.2e vel max %.2e"
              % (warm, it, int((g_nc == o_nc).sum()), n, int(np.abs(g_nc - o_nc).max()), float((dp < 1e-6).mean()), float((dp < 2e-5).mean()), int((dp >= 1e-4).sum()),
                 float(dp.max()), float((dv < 1e-4).mean()), float((dv < 2e-3).mean()), float(dv.max()), float(np.abs(g_dof[..., 0] - o_dof[..., 0]).max()),
                 float(np.abs(g_dof[..., 1] - o_dof[..., 1]).max())), flush=True)
        root, dof = o_root, o_dof
    s.close()
