"""TEST INFRASTRUCTURE: plain torch.nn modules with the STRUCTURE rl_games 1.5.2's network builder gives the networks of
cfg/lego/ppo_continuous_grasp.yaml (YG:8-29, 74-95), written from the recalled upstream layout of SURVEY.md App. C - rl_games is not
installed anywhere we run and the reference ships no checkpoint, so this is the closest stand-in for "a state_dict produced by
rl_games" that can exist here (README.md:87-96, scripts/evaluation.py:111-114 load such files).  Nothing here imports seqdex_amd:
the names come out of nn.Module's own registration, not out of seqdex_amd/rlgames_checkpoint.py.

  A2CBuilder.Network.__init__ registers, in this order: actor_cnn, critic_cnn (empty Sequentials), actor_mlp = Sequential(Linear,
  ELU, Linear, ELU, Linear, ELU), critic_mlp (the same when `separate`, else an empty Sequential), value = Linear, value_act, and
  for a continuous action space mu = Linear, mu_act, sigma_act, sigma = Parameter(zeros) (fixed_sigma).
  ModelA2CContinuousLogStd.Network / ModelCentralValue.Network hold it as `a2c_network` (+ `running_mean_std` when normalize_input).
  CentralValueTrain holds that model as `model`.
"""
import math

import torch
import torch.nn as nn


def _mlp(in_dim, units):
    layers, d = [], in_dim
    for u in units:
        layers += [nn.Linear(d, u), nn.ELU()]
        d = u
    return nn.Sequential(*layers)


class RunningMeanStd(nn.Module):
    """rl_games.algos_torch.running_mean_std.RunningMeanStd: float64 buffers, eps 1e-5, clamp +-5 in forward"""

    def __init__(self, dim):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(dim, dtype=torch.float64))
        self.register_buffer("running_var", torch.ones(dim, dtype=torch.float64))
        self.register_buffer("count", torch.ones((), dtype=torch.float64))

    def forward(self, x):
        y = (x - self.running_mean.float()) / torch.sqrt(self.running_var.float() + 1e-5)
        return torch.clamp(y, -5.0, 5.0)


class A2CNetwork(nn.Module):
    def __init__(self, in_dim, units, actions_num=0, separate=False):
        super().__init__()
        self.separate = separate
        self.actor_cnn = nn.Sequential()
        self.critic_cnn = nn.Sequential()
        self.actor_mlp = _mlp(in_dim, units)
        self.critic_mlp = _mlp(in_dim, units) if separate else nn.Sequential()
        self.value = nn.Linear(units[-1], 1)
        self.value_act = nn.Identity()
        if actions_num:
            self.mu = nn.Linear(units[-1], actions_num)
            self.mu_act = nn.Identity()
            self.sigma_act = nn.Identity()
            self.sigma = nn.Parameter(torch.zeros(actions_num, dtype=torch.float32), requires_grad=True)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def forward(self, x):
        a = self.actor_mlp(x)
        c = self.critic_mlp(x) if self.separate else a
        value = self.value_act(self.value(c))
        if hasattr(self, "mu"):
            mu = self.mu_act(self.mu(a))
            return mu, mu * 0.0 + self.sigma_act(self.sigma), value
        return value


class Model(nn.Module):
    """ModelA2CContinuousLogStd.Network / ModelCentralValue.Network"""

    def __init__(self, net, in_dim, normalize_input):
        super().__init__()
        self.a2c_network = net
        if normalize_input:
            self.running_mean_std = RunningMeanStd(in_dim)

    def forward(self, x):
        if hasattr(self, "running_mean_std"):
            x = self.running_mean_std(x)
        return self.a2c_network(x)


class CentralValueTrain(nn.Module):
    def __init__(self, state_dim, units):
        super().__init__()
        self.model = Model(A2CNetwork(state_dim, units), state_dim, normalize_input=True)

    def forward(self, states):
        return self.model(states)


def build(obs_dim=396, state_dim=564, act_dim=23, units=(1024, 512, 256), seed=0):
    torch.manual_seed(seed)
    model = Model(A2CNetwork(obs_dim, units, act_dim, separate=True), obs_dim, normalize_input=False)
    cvt = CentralValueTrain(state_dim, units)
    return model, cvt


def neglogp(x, mean, std, logstd):
    return 0.5 * (((x - mean) / std) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * x.shape[-1] + logstd.sum(-1)
