"""Phase clock of the persistent PPO update kernel (CU 0): run with SDXP_PERSIST_STAMPS=1."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from seqdex_amd.ppo import SdxPPO, make_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ag = SdxPPO(n, config=make_config(n))
g = torch.Generator().manual_seed(0)
for t in range(8):
    ag.act(t, torch.randn(n, 396, generator=g).cuda(), torch.randn(n, 564, generator=g).cuda(), None, None)
    ag.store_rewards(t, torch.rand(n, generator=g).cuda(), None)
ag.finish_rollout(torch.randn(n, 564, generator=g).cuda(), None)
torch.cuda.synchronize(); t0 = time.time()
ag.update(); torch.cuda.synchronize(); dt = time.time() - t0
d = ag.t["DEBUG"].cpu().numpy()[:32].astype(np.float64)
steps = n * 8 // 4 * 5
names = ["A:dY0 gather+gram+ctl", "A:adam L0", "A:stage+fwd0", "sh1:gram in+adam L1", "wait x1", "B:fwd1", "sh2:gram x1+adam L2/heads", "wait x2",
         "C:fwd2", "sh3:gram x2", "wait hw+x3", "D:head fwd", "D:losses,dmu", "D:bwd heads", "D:B1", "sh4:stats,grams,hterm", "wait dY1", "E:B0", "sh5:gram dY1"]
print("update %.3f s, %.1f us/step; us per step by phase (CU 0, s_memtime at 100 MHz?):" % (dt, dt / steps * 1e6))
scale = dt / d.sum()
for nm, v in zip(names, d): print("  %-26s %6.2f us" % (nm, v * scale / steps * 1e6))
