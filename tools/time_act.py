"""policy inference of one env step (sdxp_act: trunks of the actor and the central value + heads) at N envs: us per call (HIP events),
for every tile shape of k_linear_mfma (0 = the launcher's own choice).   usage: python tools/time_act.py [N]"""
import ctypes as C
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd import _abi  # noqa: E402
from seqdex_amd.ppo import SdxPPO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/lego/ppo_continuous_grasp.yaml")))
ppo = SdxPPO(n, params=tr["params"], obs_dim=396, state_dim=564)
lib = _abi.load_library()
lib.sdxpk_linear_force_shape.argtypes = [C.c_int]
obs = torch.randn(n, 396, device="cuda").clamp(-5, 5)
st = torch.randn(n, 564, device="cuda").clamp(-5, 5)
names = {0: "automatic", 1: "64x64, chunks of 32", 2: "128x64, chunks of 32", 3: "64x64, 2 k groups, chunks of 32", 4: "64x64, 2 k groups, chunks of 64",
         5: "64x64, 4 k groups, chunks of 64", 6: "64x64, chunks of 64", 7: "64x64, LDS-DMA staging, 3 image pairs"}
for shape in ([int(a) for a in sys.argv[2:]] or (0, 1, 3, 4, 2, 5, 6, 7)):
    lib.sdxpk_linear_force_shape(shape)
    for _ in range(5):
        ppo.act(0, obs, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(64):
        ppo.act(i % 8, obs, st)
    e1.record()
    torch.cuda.synchronize()
    print("sdxp_act at N = %d, tile shape %d (%s): %.1f us per call" % (n, shape, names[shape], e0.elapsed_time(e1) / 64 * 1e3), flush=True)
lib.sdxpk_linear_force_shape(0)
