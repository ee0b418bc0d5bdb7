import time, torch, numpy as np, sys
sys.path.insert(0,'.')
from seqdex_amd.sim import SdxSim
n=1024
s=SdxSim(n)
a=(torch.rand(n,23)*2-1).cuda()
for _ in range(20): s.step(a)
torch.cuda.synchronize()
def tm(f,k=50):
    torch.cuda.synchronize(); t=time.time()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.time()-t)/k*1e6
print('step us', tm(lambda: s.step(a)))
print('pre us', tm(lambda: s.pre_physics(a)))
print('sim us', tm(lambda: s.simulate()))
print('post us', tm(lambda: s.post_physics()))
print('nc', s.NCONTACTS.float().mean().item(), s.NCONTACTS.max().item())
print('progress', s.PROGRESS[:4].tolist(), 'rew', s.REW[:4].tolist())

import numpy as np
d=s.DEBUG.cpu().numpy()
print('k_physics env0 substep0 phase cycles [fk, mass, pd+twists, collide, solve, integrate]:', np.diff(d[:7]), 'np/nc env0', int(s.NCONTACTS[0]))
print('solve: [load rows, w prep | it0: zero, pass1, pass2, bodyupd+qd, twists]', np.diff(d[16:24]))
