"""Is the full-epoch difference between the two update implementations a bug or the sensitivity of the training dynamics?
Compares (a) persistent vs graph and (b) graph vs graph with ONE central-value weight perturbed by 1e-6 before the epoch."""
import sys, os, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fullsize_properties import _filled_agent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def report(tag, a, c, p0a, p0c):
    for k, p0 in (("AC_PARAMS", p0a), ("CV_PARAMS", p0c)):
        ma, mc = a.t[k] - p0, c.t[k] - p0
        print("%-28s %s move L2 %.4f / %.4f  diff L2 %.3e  max diff %.3e  cos %.4f" % (
            tag, k, float(ma.norm()), float(mc.norm()), float((ma - mc).norm()), float((ma - mc).abs().max()),
            float((ma * mc).sum() / (ma.norm() * mc.norm()))))


a = _filled_agent(n, 9); c = _filled_agent(n, 9, impl="graph"); d = _filled_agent(n, 9, impl="graph")
p0a, p0c = a.t["AC_PARAMS"].clone(), a.t["CV_PARAMS"].clone()
d.t["CV_PARAMS"][12345] += 1e-6
for ag in (a, c, d):
    ag.update()
torch.cuda.synchronize()
report("persistent vs graph", a, c, p0a, p0c)
report("graph vs graph(+1e-6 on 1 w)", c, d, p0a, p0c)
for f in ("sum_cv_loss", "cv_gnorm"):
    print(f, getattr(a.ctrl(), f), getattr(c.ctrl(), f), getattr(d.ctrl(), f))
