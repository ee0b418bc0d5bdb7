"""TEST INFRASTRUCTURE ONLY (oracle/): plain-PyTorch restatement of one iteration of the reference's transition-value trainer
(policy_sequencing/transition_value_trainer.py:209-231 = TT) for the network of policy_sequencing/terminal_value_function.py:30-46:
GraspInsertTValue 4-256-128-64-2 with ELU after every layer, BCEWithLogitsLoss against one-hot [failure, success] labels, Adam(1e-3).
Pinned by tests/golden/TV1_train.npz, which oracle/gen_golden_tvalue_train.py produced by running the reference's own module
class with torch.optim.Adam on fixed batches (the sampling itself is random in the reference and is not part of the pin)."""
import torch
from torch import nn


class GraspInsertTValue(nn.Module):                     # terminal_value_function.py:30-46
    def __init__(self, input_dim=4, output_dim=2):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(input_dim, 256), nn.Linear(256, 128)
        self.linear3, self.output_layer = nn.Linear(128, 64), nn.Linear(64, output_dim)
        self.activate_func = nn.ELU()

    def forward(self, x):
        x = self.activate_func(self.linear1(x))
        x = self.activate_func(self.linear2(x))
        x = self.activate_func(self.linear3(x))
        return self.activate_func(self.output_layer(x))


def labels(batch):
    t = torch.zeros(batch, 2)                           # TT:203-205
    t[:batch // 2, 1] = 1
    t[batch // 2:, 0] = 1
    return t


def noisy_batch(success_rows, failure_rows, noise):
    """TT:216-222: rows + U(-1,1) * 0.05 (given as `noise` [B, 4] in [-1, 1)), renormalised; successes first"""
    x = torch.cat([success_rows, failure_rows]) + noise * 0.05
    return x / torch.norm(x, dim=-1, keepdim=True)


def train_steps(state_dict, batches, lr=0.001):
    """runs len(batches) iterations (TT:225-231) from `state_dict`; returns (final state_dict, losses, outputs of the last forward)"""
    net = GraspInsertTValue()
    net.load_state_dict({k: torch.as_tensor(v).float() for k, v in state_dict.items()})
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    crit = nn.BCEWithLogitsLoss()
    losses, out = [], None
    for x in batches:
        x = torch.as_tensor(x).float()
        out = net(x)
        loss = crit(out, labels(x.shape[0]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return {k: v.detach().clone() for k, v in net.state_dict().items()}, losses, out.detach()
