"""closed-form checks that pin oracle/ppo_oracle.py (PARITY UNPINNED vs rl_games, see its header)."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle.ppo_oracle import DEFAULT_CFG, PPOOracle, RunningMeanStd  # noqa: E402


def small_cfg(**kw):
    c = dict(DEFAULT_CFG)
    c.update(obs_dim=12, state_dim=8, act_dim=3, units=[16, 8, 8], minibatch=4, mini_epochs=2)
    c.update(kw)
    return c


def test_gae_hand_computed():
    o = PPOOracle(small_cfg())
    r = torch.tensor([[1.0], [2.0], [3.0]]); v = torch.tensor([[0.5], [0.4], [0.3]])
    d = torch.tensor([[0.0], [0.0], [1.0]])                # done flag stored BEFORE step t (PS:347)
    adv, ret = o.gae(r, v, d, torch.tensor([0.2]), torch.tensor([0.0]))
    g, lam = 0.99, 0.95
    a2 = 3.0 + g * 0.2 - 0.3
    a1 = 2.0 + g * 0.3 * 0.0 - 0.4                          # next step (t=2) starts a new episode: no bootstrap
    a0 = 1.0 + g * 0.4 - 0.5 + g * lam * a1
    np.testing.assert_allclose(adv.squeeze(1).numpy(), [a0, a1, a2], rtol=1e-6)
    np.testing.assert_allclose(ret.numpy(), (adv + v).numpy())


def test_neglogp_matches_torch_normal():
    mu, ls = torch.randn(5, 3), torch.randn(3) * 0.1
    x = torch.randn(5, 3)
    want = -torch.distributions.Normal(mu, ls.exp()).log_prob(x).sum(-1)
    got = PPOOracle.neglogp(x, mu, ls.exp().expand_as(mu), ls)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def test_running_mean_std_single_merge_closed_form():
    """one merge of a 4-sample batch into the prior (count 1, mean 0, var 1).  rl_games feeds the UNBIASED batch
    variance into the parallel-variance formula (m_b = var_unbiased * batch_count), reproduced here."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, generator=g) * 3 + 1
    r = RunningMeanStd(6)
    r.update(x)
    xb = x.double().mean(0)
    ss = ((x.double() - xb) ** 2).sum(0)
    np.testing.assert_allclose(r.mean.numpy(), (4 * xb / 5).numpy(), rtol=1e-12)
    np.testing.assert_allclose(r.var.numpy(), ((1.0 + (4.0 / 3.0) * ss + xb ** 2 * 4.0 / 5.0) / 5.0).numpy(), rtol=1e-12)
    assert float(r.count) == 5.0
    y = r(x)
    assert float(y.abs().max()) <= 5.0


def test_update_first_step_is_plain_clipped_adam():
    """first minibatch: ratio == 1 -> a_loss grad = -A * dlogp; Adam's first step moves every touched weight by ~lr."""
    torch.manual_seed(0)
    c = small_cfg(adaptive_lr=False, mini_epochs=1)
    o = PPOOracle(c)
    n = 4
    obs, st = torch.randn(n, 12), torch.randn(n, 8)
    r = o.act(obs, st, torch.randn(n, 3))
    ds = dict(obs=obs, states=st, actions=r["actions"], mus=r["mus"].clone(), sigmas=r["sigmas"].clone(),
              neglogp=r["neglogp"], values=r["values"], returns=r["values"] + torch.tensor([1.0, -1.0, 0.5, -0.5]))
    before = o.ac_flat().clone()
    stats = o.update(ds)
    delta = (o.ac_flat() - before).abs()
    assert stats["kl"][0] < 1e-4                                  # KL of a policy with itself (only the 1e-5 epsilons)
    assert delta.max() <= c["lr"] * 1.0001 and delta.max() > 0.5 * c["lr"]
    assert abs(stats["a"][0]) < 1e-6 + abs(float((-(ds["advantages"])).mean()))   # ratio == 1: a_loss = -A
    assert stats["gnorm"][0] > 0


def test_adaptive_lr_rule():
    c = small_cfg()
    o = PPOOracle(c)
    o.lr = 3e-4
    # the legacy schedule (PS:306-312): kl > 2*thr -> lr/1.5 ; kl < thr/2 -> lr*1.5 ; clamp [1e-6, 1e-2]
    for kl, want in [(0.05, 3e-4 / 1.5), (0.001, 3e-4 * 1.5), (0.02, 3e-4)]:
        lr = 3e-4
        if kl > 2 * c["kl_threshold"]:
            lr = max(lr / 1.5, 1e-6)
        if kl < 0.5 * c["kl_threshold"]:
            lr = min(lr * 1.5, 1e-2)
        assert math.isclose(lr, want)
