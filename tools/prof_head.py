import sys, torch, numpy as np
sys.path.insert(0,'.')
from seqdex_amd.ppo import SdxPPO, make_config
n=64
ag=SdxPPO(n, config=make_config(n))
g=torch.Generator().manual_seed(0)
for t in range(8):
    ag.act(t, torch.randn(n,396,generator=g).cuda(), torch.randn(n,564,generator=g).cuda(), None, None)
    ag.store_rewards(t, torch.rand(n,generator=g).cuda(), None)
ag.finish_rollout(torch.randn(n,564,generator=g).cuda(), None)
ag.update(); torch.cuda.synchronize()
d=ag.t["DEBUG"].cpu().numpy()
print('k_head phase cycles:', np.diff(d[:6]))
print('k_ctrl phase cycles:', np.diff(d[8:12]))
