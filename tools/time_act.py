"""policy inference of one env step (sdxp_act: trunks of the actor and the central value + heads) at N envs: us per call (HIP events).
usage: [SDXP_LINEAR_TILE=1|2] python tools/time_act.py [N]"""
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seqdex_amd.ppo import SdxPPO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tr = yaml.safe_load(open(os.path.join(ROOT, "seqdex_amd/cfg/lego/ppo_continuous_grasp.yaml")))
ppo = SdxPPO(n, params=tr["params"], obs_dim=396, state_dim=564)
obs = torch.randn(n, 396, device="cuda").clamp(-5, 5)
st = torch.randn(n, 564, device="cuda").clamp(-5, 5)
for _ in range(5):
    ppo.act(0, obs, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(64):
    ppo.act(i % 8, obs, st)
e1.record()
torch.cuda.synchronize()
print("sdxp_act at N = %d, SDXP_LINEAR_TILE=%s: %.1f us per call" % (n, os.environ.get("SDXP_LINEAR_TILE", "auto"), e0.elapsed_time(e1) / 64 * 1e3))
