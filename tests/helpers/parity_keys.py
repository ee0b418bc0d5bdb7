"""which contacts differ between the GPU kernel and the C oracle after one teacher-forced step of the golden piles (test infrastructure: uses the
oracle).  Both sides run the warm-started solver from an empty cache; their caches then hold the contact identities of the LAST solve of the
step (second substep).  Keys are decoded to (box a, box b, box pair, direction, sample) - the two sides number body pairs differently - and
compared as sets, env by env; bricks whose pose differs by more than 1e-4 m are listed with the differing contacts that involve them.
usage: python tests/helpers/parity_keys.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import physics_oracle as po  # noqa: E402
from seqdex_amd.sim import SdxSim  # noqa: E402

from tests.helpers.contact_keys import decode_kernel as decode_gpu, decode_oracle  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
state = np.load(os.path.join(ROOT, "tests", "golden", "P1_settled_state.npz"))
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
n = state["root"].shape[0]
s = SdxSim(n, warm_start=0.8)
ns = int(s._desc.n_static)
root, dof = state["root"].copy(), state["dof"].copy()
ow = po.WarmState(n)
for it in range(steps):
    s.ROOT.copy_(dev(root.reshape(-1, 13)))
    s.DOF.copy_(dev(dof.reshape(-1, 2)))
    s.TARGETS.copy_(dev(state["targets"]))
    s.simulate()
    torch.cuda.synchronize()
    g_root = s.ROOT.cpu().numpy().reshape(n, 142, 13)
    g_cnt = s.WARM_COUNT.cpu().numpy()
    g_key = s.WARM_KEYS.cpu().numpy().view(np.uint32)
    g_lam = s.WARM_LAMBDA.cpu().numpy()
    o_root, o_dof = root.copy(), dof.copy()
    po.simulate(s._desc, o_root, o_dof, state["targets"], ow)
    dp = np.abs(g_root[:, 9:81, 0:7] - o_root[:, 9:81, 0:7]).max(-1)
    dv = np.abs(g_root[:, 9:81, 7:13] - o_root[:, 9:81, 7:13]).max(-1)
    print("step %d: cstats %s" % (it, s.CONTACT_STATS.cpu().tolist()), flush=True)
    for e in range(n):
        G = {decode_gpu(g_key[e, c]): g_lam[e, :, c] for c in range(g_cnt[e])}
        O = {decode_oracle(ow.key[e, c], ns): ow.lam[e, :, c] for c in range(ow.count[e])}
        only_g, only_o = sorted(set(G) - set(O)), sorted(set(O) - set(G))
        bad = np.nonzero(dp[e] >= 1e-4)[0]
        print(" env %d: contacts gpu %d oracle %d, only on the gpu %d, only in the oracle %d; bricks off by >= 1e-4: %s (target brick %d)"
              % (e, len(G), len(O), len(only_g), len(only_o), [(int(b), float("%.2e" % dp[e, b]), float("%.2e" % dv[e, b])) for b in bad],
                 s.scene.seg_index(e) - 9), flush=True)
        for k in only_g[:12]:
            print("    gpu only   ", k, G[k])
        for k in only_o[:12]:
            print("    oracle only", k, O[k])
        # the largest impulse differences among the common contacts
        common = sorted(set(G) & set(O), key=lambda k: -abs(G[k][0] - O[k][0]))[:4]
        for k in common:
            print("    common, largest normal-impulse difference", k, G[k], O[k])
    root, dof = o_root, o_dof
s.close()
