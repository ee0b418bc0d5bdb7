"""TEST INFRASTRUCTURE ONLY (oracle/): plain-PyTorch (autograd, CPU, fp32) restatement of the PPO arithmetic that
rl_games==1.5.2's A2CAgent performs for train_rlgames.py --task=BlockAssemblyGraspSim.

PARITY UNPINNED: rl_games is a third-party dependency (requirements.txt:6) that is absent from /root/reference and
not installed here; the reference ships no tests or golden vectors for it.  This restatement follows the published
algorithm as mirrored in-tree by policy_sequencing/policy_seq_runner.py (PS) and utils/rl_games_custom.py (RC):
  network       YG:8-29 (cfg/lego/ppo_continuous_grasp.yaml), biases zeroed, fixed_sigma logstd Parameter
  action/neglogp RC:1697-1723, RC:2114-2126
  GAE           PS:329-336
  dataset       PS:338-339 (env-major flatten), RC:1639-1651 (advantage normalisation, unbiased std)
  losses        RC:1813-1822, RC:2129-2132 ; bound loss soft bound 1.1 ; policy_kl
  step          RC:1859-1877 (clip_grad_norm_ 1.0, Adam eps 1e-8) ; legacy adaptive LR PS:306-312
  central value separate net + Adam(1e-3), running mean/std input normalisation updated in the first mini-epoch
It is checked against closed forms in tests/test_ppo_oracle.py and used as the checker of the HIP kernels in
tests/test_gpu_ppo_parity.py.  Nothing in seqdex_amd/ imports it.
"""
import math

import torch
import torch.nn as nn


class MLP(nn.Module):
    def __init__(self, in_dim, units, out_dim):
        super().__init__()
        dims = [in_dim] + list(units)
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(units))])
        self.head = nn.Linear(dims[-1], out_dim)
        for m in list(self.layers) + [self.head]:
            nn.init.zeros_(m.bias)

    def forward(self, x):
        for l in self.layers:
            x = nn.functional.elu(l(x))
        return self.head(x)


class RunningMeanStd:
    """rl_games RunningMeanStd (float64 statistics, eps 1e-5, clamp +-5)."""

    def __init__(self, dim):
        self.mean = torch.zeros(dim, dtype=torch.float64)
        self.var = torch.ones(dim, dtype=torch.float64)
        self.count = torch.ones((), dtype=torch.float64)

    def update(self, x):
        bm = x.double().mean(0)
        bv = x.double().var(0)  # unbiased, torch default
        bc = x.shape[0]
        delta = bm - self.mean
        tot = self.count + bc
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean = self.mean + delta * bc / tot
        self.var = m2 / tot
        self.count = tot

    def __call__(self, x):
        y = (x - self.mean.float()) / torch.sqrt(self.var.float() + 1e-5)
        return torch.clamp(y, -5.0, 5.0)


class PPOOracle:
    def __init__(self, cfg, seed=0):
        torch.manual_seed(seed)
        self.cfg = dict(cfg)
        c = self.cfg
        self.actor = MLP(c["obs_dim"], c["units"], c["act_dim"])
        self.critic = MLP(c["obs_dim"], c["units"], 1)
        self.logstd = nn.Parameter(torch.zeros(c["act_dim"]))
        self.cv = MLP(c["state_dim"], c["units"], 1)
        self.rms = RunningMeanStd(c["state_dim"])
        self.ac_params = list(self.actor.parameters()) + [self.logstd] + list(self.critic.parameters())
        self.opt = torch.optim.Adam(self.ac_params, lr=c["lr"], eps=1e-8)
        self.cv_opt = torch.optim.Adam(self.cv.parameters(), lr=c["cv_lr"], eps=1e-8)
        self.lr = c["lr"]

    # ---- flat parameter vectors in the layout of SDXP_T_AC_PARAMS / SDXP_T_CV_PARAMS
    def ac_flat(self):
        parts = []
        for l in self.actor.layers:
            parts += [l.weight.reshape(-1), l.bias]
        parts += [self.actor.head.weight.reshape(-1), self.actor.head.bias, self.logstd]
        for l in self.critic.layers:
            parts += [l.weight.reshape(-1), l.bias]
        parts += [self.critic.head.weight.reshape(-1), self.critic.head.bias]
        return torch.cat([p.detach().reshape(-1) for p in parts])

    def cv_flat(self):
        parts = []
        for l in self.cv.layers:
            parts += [l.weight.reshape(-1), l.bias]
        parts += [self.cv.head.weight.reshape(-1), self.cv.head.bias]
        return torch.cat([p.detach().reshape(-1) for p in parts])

    def load_flat(self, ac, cv):
        def fill(params, flat):
            o = 0
            for p in params:
                n = p.numel()
                p.data.copy_(flat[o:o + n].reshape(p.shape))
                o += n
            assert o == flat.numel()
        a = []
        for l in self.actor.layers:
            a += [l.weight, l.bias]
        a += [self.actor.head.weight, self.actor.head.bias, self.logstd]
        for l in self.critic.layers:
            a += [l.weight, l.bias]
        a += [self.critic.head.weight, self.critic.head.bias]
        fill(a, torch.as_tensor(ac))
        c = []
        for l in self.cv.layers:
            c += [l.weight, l.bias]
        c += [self.cv.head.weight, self.cv.head.bias]
        fill(c, torch.as_tensor(cv))

    # ---- R3: get_action_values
    @torch.no_grad()
    def act(self, obs, states, eps):
        mu = self.actor(obs)
        sigma = torch.exp(self.logstd).expand_as(mu)
        a = mu + sigma * eps
        nlp = self.neglogp(a, mu, sigma, self.logstd)
        v = self.cv(self.rms(states)).squeeze(-1)
        return dict(actions=a, mus=mu, sigmas=sigma, neglogp=nlp, values=v)

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * x.shape[-1] + logstd.sum(-1)

    @staticmethod
    def ac_loss(a_loss, c_loss, critic_coef, entropy, entropy_coef, b_loss, bounds_loss_coef):
        """composition of the losses (RC:2129-2132): a + 0.5 c critic_coef - entropy entropy_coef + b bounds_loss_coef"""
        return a_loss + 0.5 * c_loss * critic_coef - entropy * entropy_coef + b_loss * bounds_loss_coef

    @staticmethod
    def normalize_advantages(returns, values):
        """RC:1639-1651: (A - mean) / (std_unbiased + 1e-8) over the whole batch, A = returns - values"""
        adv = returns - values
        return (adv - adv.mean()) / (adv.std() + 1e-8)

    @torch.no_grad()
    def values(self, states):
        return self.cv(self.rms(states)).squeeze(-1)

    # ---- R5: discount_values.  All arrays [H, N]
    def gae(self, rewards, values, dones, last_values, last_dones):
        c = self.cfg
        H = rewards.shape[0]
        adv = torch.zeros_like(rewards)
        lastgae = torch.zeros_like(last_values)
        for t in reversed(range(H)):
            if t == H - 1:
                nonterminal, nextv = 1.0 - last_dones, last_values
            else:
                nonterminal, nextv = 1.0 - dones[t + 1], values[t + 1]
            delta = rewards[t] + c["gamma"] * nextv * nonterminal - values[t]
            lastgae = delta + c["gamma"] * c["tau"] * nonterminal * lastgae
            adv[t] = lastgae
        return adv, adv + values

    # ---- R6-R8 + central value: one train_epoch's update phase on an env-major dataset dict
    def update(self, ds, max_steps=None):
        """ds: dict of env-major flattened tensors (rows r = env*H + t): obs, states, actions, mus, sigmas, neglogp,
        values, returns.  Mutates ds['mus'/'sigmas'] like dataset.update_mu_sigma.  Returns statistics.
        max_steps: stop each of the two loops after that many minibatches (the library's SDXP_MAX_STEPS debug limit)."""
        c = self.cfg
        mbs = c["minibatch"]
        nmb = ds["obs"].shape[0] // mbs
        adv = ds["returns"] - ds["values"]
        if c.get("normalize_advantage", True):
            adv = self.normalize_advantages(ds["returns"], ds["values"])
        ds["advantages"] = adv
        stats = dict(a=[], c=[], b=[], kl=[], cv=[], lr=[], gnorm=[], cv_gnorm=[])
        # central value first (RC:1323-1324)
        todo = [(ep, i) for ep in range(c["mini_epochs"]) for i in range(nmb)]
        if max_steps is not None:
            todo = todo[:max_steps]
        for ep, i in todo:
            if True:
                sl = slice(i * mbs, (i + 1) * mbs)
                st = ds["states"][sl]
                if ep == 0 and c.get("cv_normalize_input", True):
                    self.rms.update(st)
                v = self.cv(self.rms(st) if c.get("cv_normalize_input", True) else st).squeeze(-1)
                loss = self._critic_loss(ds["values"][sl], v, ds["returns"][sl]).mean()
                self.cv_opt.zero_grad()
                loss.backward()
                gn = nn.utils.clip_grad_norm_(self.cv.parameters(), c["grad_norm"])
                self.cv_opt.step()
                stats["cv"].append(float(loss))
                stats["cv_gnorm"].append(float(gn))
        for ep, i in todo:
            if True:
                sl = slice(i * mbs, (i + 1) * mbs)
                obs = ds["obs"][sl]
                mu = self.actor(obs)
                sigma = torch.exp(self.logstd).expand_as(mu)
                nlp = self.neglogp(ds["actions"][sl], mu, sigma, self.logstd)
                v = self.critic(obs).squeeze(-1)
                ratio = torch.exp(ds["neglogp"][sl] - nlp)
                A = adv[sl]
                a_loss = torch.max(-A * ratio, -A * torch.clamp(ratio, 1 - c["e_clip"], 1 + c["e_clip"]))
                c_loss = self._critic_loss(ds["values"][sl], v, ds["returns"][sl])
                b_loss = (torch.clamp_min(mu - 1.1, 0.0) ** 2 + torch.clamp_max(mu + 1.1, 0.0) ** 2).sum(-1)
                loss = self.ac_loss(a_loss.mean(), c_loss.mean(), c["critic_coef"], torch.zeros(()), 0.0, b_loss.mean(), c["bounds_loss_coef"])
                for p in self.ac_params:
                    p.grad = None
                loss.backward()
                gn = nn.utils.clip_grad_norm_(self.ac_params, c["grad_norm"])
                for g in self.opt.param_groups:
                    g["lr"] = self.lr
                self.opt.step()
                with torch.no_grad():
                    omu, osg = ds["mus"][sl], ds["sigmas"][sl]
                    mu_d, sg_d = mu.detach(), sigma.detach()
                    kl = (torch.log(osg / sg_d + 1e-5) + (sg_d ** 2 + (omu - mu_d) ** 2) / (2.0 * (osg ** 2 + 1e-5)) - 0.5).sum(-1).mean()
                    ds["mus"][sl] = mu_d
                    ds["sigmas"][sl] = sg_d
                stats["a"].append(float(a_loss.mean())); stats["c"].append(float(c_loss.mean()))
                stats["b"].append(float(b_loss.mean())); stats["kl"].append(float(kl)); stats["lr"].append(self.lr)
                stats["gnorm"].append(float(gn))
                if c.get("adaptive_lr", True):   # legacy schedule: after every minibatch
                    if float(kl) > 2.0 * c["kl_threshold"]:
                        self.lr = max(self.lr / 1.5, 1e-6)
                    if float(kl) < 0.5 * c["kl_threshold"]:
                        self.lr = min(self.lr * 1.5, 1e-2)
        return stats

    def _critic_loss(self, old_v, v, ret):
        c = self.cfg
        if c.get("clip_value", True):
            vc = old_v + torch.clamp(v - old_v, -c["e_clip"], c["e_clip"])
            return torch.max((v - ret) ** 2, (vc - ret) ** 2)
        return (ret - v) ** 2


DEFAULT_CFG = dict(obs_dim=396, state_dim=564, act_dim=23, units=[1024, 512, 256], horizon=8, minibatch=4, mini_epochs=5,
                   gamma=0.99, tau=0.95, lr=3e-4, cv_lr=1e-3, e_clip=0.1, grad_norm=1.0, critic_coef=1.0,
                   bounds_loss_coef=1e-3, kl_threshold=0.02, clip_value=True, normalize_advantage=True,
                   cv_normalize_input=True, adaptive_lr=True)
