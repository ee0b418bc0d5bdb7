#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle/): golden vectors of the transition-value training step, produced with the REFERENCE's network class
(policy_sequencing/terminal_value_function.py::GraspInsertTValue, imported from /root/reference in this container) under the loss and
optimiser its trainer uses (transition_value_trainer.py:187,189,225-231: BCEWithLogitsLoss, Adam lr 1e-3).

  python oracle/gen_golden_tvalue_train.py      # needs /root/reference; writes tests/golden/TV1_train.npz (data only)
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (mocks isaacgym / cv2 / h5py and puts the reference on sys.path)


def main():
    G.import_reference()
    import importlib
    from unittest import mock
    sys.modules.setdefault("utils.cnn_module", mock.MagicMock())
    tvf = importlib.import_module("policy_sequencing.terminal_value_function")
    torch.manual_seed(11)
    net = tvf.GraspInsertTValue(input_dim=4, output_dim=2)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(net.parameters(), lr=0.001)
    crit = torch.nn.BCEWithLogitsLoss()
    g = torch.Generator().manual_seed(5)
    B = 1024
    succ = G.rand_quat(g, 700)
    fail = G.rand_quat(g, 900)
    fail[:, 3] = -fail[:, 3].abs() * 0.3 - 0.2            # separable-ish classes
    fail = fail / fail.norm(dim=-1, keepdim=True)
    target = torch.zeros(B, 2)
    target[:B // 2, 1] = 1
    target[B // 2:, 0] = 1
    out = {"succ": succ.numpy(), "fail": fail.numpy()}
    for k, v in sd0.items():
        out["w0_" + k.replace(".", "_")] = v.numpy()
    losses = []
    for it in range(4):
        si = torch.randint(0, 700, (B // 2,), generator=g)
        fi = torch.randint(0, 900, (B // 2,), generator=g)
        noise = torch.rand(B, 4, generator=g) * 2 - 1
        x = torch.cat([succ[si], fail[fi]]) + noise * 0.05
        x = x / torch.norm(x, dim=-1, keepdim=True)
        pred = net(x)
        loss = crit(pred, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        out["x%d" % it] = x.numpy()
        out["si%d" % it], out["fi%d" % it], out["noise%d" % it] = si.numpy(), fi.numpy(), noise.numpy()
        if it == 0:
            out["pred0"] = pred.detach().numpy()
            for n_, p in net.named_parameters():
                out["g0_" + n_.replace(".", "_")] = p.grad.detach().numpy().copy()
        if it in (0, 3):                                   # parameters after the first and after the fourth Adam step
            for k, v in net.state_dict().items():
                out["w%d_" % (it + 1) + k.replace(".", "_")] = v.detach().numpy().copy()
    out["losses"] = np.array(losses, np.float32)
    path = os.path.join(G.OUT, "TV1_train.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), losses)


if __name__ == "__main__":
    main()
