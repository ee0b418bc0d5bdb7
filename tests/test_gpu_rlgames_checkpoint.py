"""-m gpu: SURVEY.md section 8(f) rank 4 - a checkpoint written the way rl_games writes one (torch.save of {'model': model.state_dict(),
'assymetric_vf_nets': central_value_net.state_dict(), 'optimizer': optimizer.state_dict(), ...}) by a plain torch.nn module with rl_games'
builder structure (tests/helpers/rlgames_like.py) is restored through A2CAgent.restore, and the HIP path then computes THAT module's
outputs: the deterministic action (= mu, YG:69 `player.deterministic: True`), the sampled action and its neglogp, and the central
value of the running-mean/std-normalised state (README.md:87-96 and scripts/evaluation.py:111-114 load released checkpoints this way).
The converse direction - a file written by A2CAgent.save loads into the module with strict=True and reproduces the HIP outputs - is
checked too.  PARITY UNPINNED against rl_games itself (absent): the module's structure is the recalled one of SURVEY.md App. C."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from helpers import rlgames_like as RL  # noqa: E402


def _agent(n, seed):
    from seqdex_amd.a2c_agent import A2CAgent
    from seqdex_amd.ppo import SdxPPO, make_config
    ppo = SdxPPO(n, config=make_config(n), seed=seed)
    ag = A2CAgent.__new__(A2CAgent)
    ag.ppo, ag.epoch_num, ag.frame, ag.last_mean_rewards = ppo, 0, 0, -100500
    return ag


def _module_outputs(model, cvt, obs, states, eps):
    with torch.no_grad():
        mu, sigma, _ = model(obs)
        sigma = torch.exp(sigma)                          # continuous_a2c_logstd: the network's `sigma` output is log(std)
        a = mu + sigma * eps
        nlp = RL.neglogp(a, mu, sigma, torch.log(sigma))
        v = cvt(states).squeeze(-1)
    return mu, a, nlp, v


def test_restore_of_an_rlgames_written_checkpoint_reproduces_the_module(tmp_path):
    n = 64
    model, cvt = RL.build(seed=5)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                 # a "trained" file: non-trivial sigma, biases and running statistics
        model.a2c_network.sigma.copy_(0.3 * torch.randn(23, generator=g))
        for m in list(model.modules()) + list(cvt.modules()):
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
        cvt.model.running_mean_std.running_mean.copy_(torch.randn(564, generator=g).double())
        cvt.model.running_mean_std.running_var.copy_((0.5 + torch.rand(564, generator=g)).double())
        cvt.model.running_mean_std.count.fill_(12345.0)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, eps=1e-8)
    x = torch.randn(16, 396, generator=g)
    for _ in range(2):
        mu, sigma, v = model(x)
        (mu.pow(2).mean() + v.pow(2).mean() + sigma.sum() * 0.01).backward()
        opt.step(); opt.zero_grad()
    path = str(tmp_path / "last_AllegroHandLegoTestPAISim_ep_19000_rew_1530.9819.pth")
    torch.save({"model": model.state_dict(), "assymetric_vf_nets": cvt.state_dict(), "optimizer": opt.state_dict(), "epoch": 19000,
                "frame": 19000 * 8 * 1024, "last_mean_rewards": 1530.9819, "env_state": None}, path)
    ag = _agent(n, seed=77)
    try:
        ag.restore(path)
        assert ag.epoch_num == 19000 and abs(ag.last_mean_rewards - 1530.9819) < 1e-9
        obs = torch.randn(n, 396, generator=g).clamp(-5, 5)
        st = (torch.randn(n, 564, generator=g) * 2).clamp(-5, 5)
        eps = torch.randn(n, 23, generator=g)
        dones = torch.zeros(n, dtype=torch.int64)
        mu, a, nlp, v = _module_outputs(model, cvt, obs, st, eps)
        a_det = ag.ppo.act(0, obs.cuda(), st.cuda(), dones.cuda(), torch.zeros(n, 23).cuda()).cpu().clone()       # the player's mean action
        np.testing.assert_allclose(a_det.numpy(), mu.numpy(), rtol=2e-5, atol=2e-5)
        a_smp = ag.ppo.act(1, obs.cuda(), st.cuda(), dones.cuda(), eps.cuda()).cpu().clone()
        torch.cuda.synchronize()
        np.testing.assert_allclose(a_smp.numpy(), a.numpy(), rtol=2e-5, atol=2e-5)
        T = ag.ppo.t
        np.testing.assert_allclose(T["MB_NEGLOGP"].cpu().numpy()[:, 1], nlp.numpy(), rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(T["MB_VALUES"].cpu().numpy()[:, 1], v.numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(ag.ppo.get_values(st.cuda()).cpu().numpy(), v.numpy(), rtol=2e-5, atol=2e-5)
        # the optimiser state came along: Adam moments in the flat layout, step counter, RunningMeanStd.count
        c = ag.ppo.ctrl()
        assert c.ac_t == 2 and c.rms_count == 12345.0
        sd = opt.state_dict()
        names = [k for k, _ in model.named_parameters()]
        i_mu = names.index("a2c_network.mu.weight")
        off_mu = 1024 * 396 + 1024 + 512 * 1024 + 512 + 256 * 512 + 256
        got = T["AC_ADAM_M"].cpu()[off_mu:off_mu + 23 * 256].reshape(23, 256)
        assert torch.equal(got, sd["state"][i_mu]["exp_avg"]) and float(got.abs().max()) > 0
    finally:
        ag.ppo.close()


def test_a_saved_checkpoint_loads_into_the_rlgames_like_module(tmp_path):
    n = 32
    ag = _agent(n, seed=9)
    try:
        g = torch.Generator().manual_seed(2)
        with torch.no_grad():
            ag.ppo.t["CV_RMS_MEAN"].copy_(torch.randn(564, generator=g).double())
            ag.ppo.t["CV_RMS_VAR"].copy_((0.5 + torch.rand(564, generator=g)).double())
        ag.save(str(tmp_path / "ck"))
        ck = torch.load(str(tmp_path / "ck.pth"), map_location="cpu", weights_only=False)
        model, cvt = RL.build(seed=1)
        model.load_state_dict(ck["model"], strict=True)                     # rl_games: self.model.load_state_dict(weights['model'])
        cvt.load_state_dict(ck["assymetric_vf_nets"], strict=True)          #           self.central_value_net.load_state_dict(...)
        torch.optim.Adam(model.parameters(), lr=1.0).load_state_dict(ck["optimizer"])
        obs = torch.randn(n, 396, generator=g).clamp(-5, 5)
        st = (torch.randn(n, 564, generator=g) * 2).clamp(-5, 5)
        mu, _, _, v = _module_outputs(model, cvt, obs, st, torch.zeros(n, 23))
        a = ag.ppo.act(0, obs.cuda(), st.cuda(), torch.zeros(n, dtype=torch.int64).cuda(), torch.zeros(n, 23).cuda()).cpu()
        np.testing.assert_allclose(a.numpy(), mu.numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(ag.ppo.get_values(st.cuda()).cpu().numpy(), v.numpy(), rtol=2e-5, atol=2e-5)
    finally:
        ag.ppo.close()
