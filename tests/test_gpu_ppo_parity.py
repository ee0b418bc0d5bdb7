"""-m gpu: the HIP PPO kernels (through the sdxp_* C ABI) against oracle/ppo_oracle.py, a plain-PyTorch autograd
restatement of the rl_games arithmetic (PARITY UNPINNED vs rl_games itself - see the oracle header).
SURVEY.md §8(a) rows R1-R9."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle.ppo_oracle import DEFAULT_CFG, PPOOracle  # noqa: E402


def make_pair(n, seed=3, **over):
    from seqdex_amd import _abi
    from seqdex_amd.ppo import SdxPPO, make_config
    cfg = make_config(n)
    for k, v in over.items():
        setattr(cfg, k, v)
    agent = SdxPPO(n, config=cfg, seed=seed)
    oc = dict(DEFAULT_CFG)
    oc.update(minibatch=cfg.minibatch, mini_epochs=cfg.mini_epochs, lr=cfg.lr, cv_lr=cfg.cv_lr,
              adaptive_lr=bool(cfg.adaptive_lr), critic_coef=cfg.critic_coef)
    orc = PPOOracle(oc, seed=0)
    orc.load_flat(agent.t["AC_PARAMS"].cpu(), agent.t["CV_PARAMS"].cpu())
    return agent, orc


def test_param_layout_and_init():
    agent, orc = make_pair(16)
    try:
        assert agent.param_count(0) == 2131503 and agent.param_count(1) == 1234945
        np.testing.assert_array_equal(orc.ac_flat().numpy(), agent.t["AC_PARAMS"].cpu().numpy())
        p = agent.t["AC_PARAMS"].cpu().numpy()
        w1 = p[:1024 * 396]
        assert abs(np.abs(w1).max() - 1 / np.sqrt(396)) < 2e-3                              # U(-1/sqrt(fan_in), +)
        assert (p[1024 * 396:1024 * 396 + 1024] == 0).all()                                 # biases zeroed
    finally:
        agent.close()


def rollout(agent, orc, n, g, steps=8):
    """drive both sides with the same synthetic observations and the same noise; returns the oracle-side dataset"""
    H = steps
    obs_l, st_l, eps_l, rew_l, done_l = [], [], [], [], []
    buf = dict(actions=[], mus=[], sigmas=[], neglogp=[], values=[])
    for t in range(H):
        obs = torch.randn(n, 396, generator=g).clamp(-5, 5)
        st = torch.randn(n, 564, generator=g).clamp(-5, 5) * 2
        eps = torch.randn(n, 23, generator=g)
        dones = (torch.rand(n, generator=g) < 0.15).long()
        rew = torch.rand(n, generator=g)
        a = agent.act(t, obs.cuda(), st.cuda(), dones.cuda(), eps.cuda())
        agent.store_rewards(t, rew.cuda(), dones.cuda())
        r = orc.act(obs, st, eps)
        np.testing.assert_allclose(a.cpu().numpy(), r["actions"].numpy(), rtol=2e-4, atol=2e-4)
        for k in buf:
            buf[k].append(r[k])
        obs_l.append(obs); st_l.append(st); eps_l.append(eps); rew_l.append(rew); done_l.append(dones.float())
    last_st = torch.randn(n, 564, generator=g)
    last_done = (torch.rand(n, generator=g) < 0.15).long()
    agent.finish_rollout(last_st.cuda(), last_done.cuda())
    torch.cuda.synchronize()
    values = torch.stack(buf["values"])
    adv, ret = orc.gae(torch.stack(rew_l), values, torch.stack(done_l), orc.values(last_st), last_done.float())
    flat = lambda x: torch.stack(x).transpose(0, 1).reshape(n * H, *x[0].shape[1:]).contiguous()   # swap_and_flatten01
    ds = dict(obs=flat(obs_l), states=flat(st_l), actions=flat(buf["actions"]), mus=flat(buf["mus"]).clone(),
              sigmas=flat(buf["sigmas"]).clone(), neglogp=flat(buf["neglogp"]), values=flat(buf["values"]),
              returns=ret.transpose(0, 1).reshape(-1).contiguous())
    return ds


def test_rollout_gae_and_dataset():
    n = 64
    agent, orc = make_pair(n)
    try:
        g = torch.Generator().manual_seed(1)
        ds = rollout(agent, orc, n, g)
        T = agent.t
        np.testing.assert_allclose(T["MB_NEGLOGP"].cpu().numpy().reshape(-1), ds["neglogp"].numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(T["MB_VALUES"].cpu().numpy().reshape(-1), ds["values"].numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(T["MB_MUS"].cpu().numpy().reshape(-1, 23), ds["mus"].numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_array_equal(T["MB_OBS"].cpu().numpy().reshape(-1, 396), ds["obs"].numpy())
        np.testing.assert_allclose(T["RETURNS"].cpu().numpy(), ds["returns"].numpy(), rtol=2e-4, atol=2e-4)
        adv = ds["returns"] - ds["values"]
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        np.testing.assert_allclose(T["ADVANTAGES"].cpu().numpy(), adv.numpy(), rtol=2e-3, atol=2e-3)
    finally:
        agent.close()


@pytest.mark.parametrize("adaptive", [0, 1])
def test_update_matches_autograd_adam(adaptive):
    """full update phase (5 mini-epochs x 32 minibatches of 4, three networks): parameters, Adam moments, running
    mean/std, LR schedule and loss statistics against torch.autograd + torch.optim.Adam + clip_grad_norm_."""
    n = 16
    agent, orc = make_pair(n, adaptive_lr=adaptive)
    try:
        g = torch.Generator().manual_seed(2)
        ds = rollout(agent, orc, n, g)
        agent.update()
        torch.cuda.synchronize()
        st = orc.update(ds)
        c = agent.ctrl()
        nsteps = 5 * (n * 8 // 4)
        assert c.n_mb == nsteps and c.ac_t == nsteps and c.ac_pending == 0
        # statistics (means over all minibatches)
        np.testing.assert_allclose(c.sum_a_loss / nsteps, np.mean(st["a"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_c_loss / nsteps, np.mean(st["c"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_cv_loss / nsteps, np.mean(st["cv"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_kl / nsteps, np.mean(st["kl"]), rtol=5e-3, atol=1e-5)
        np.testing.assert_allclose(c.ac_lr, orc.lr, rtol=1e-6)
        np.testing.assert_allclose(c.ac_gnorm, st["gnorm"][-1], rtol=2e-3)
        np.testing.assert_allclose(c.cv_gnorm, st["cv_gnorm"][-1], rtol=2e-3)
        # parameters after 160 optimiser steps per network
        ac = agent.t["AC_PARAMS"].cpu().numpy(); cv = agent.t["CV_PARAMS"].cpu().numpy()
        oa = orc.ac_flat().numpy(); ocv = orc.cv_flat().numpy()
        assert np.abs(ac - oa).max() < 2e-4, np.abs(ac - oa).max()
        assert np.abs(cv - ocv).max() < 5e-4, np.abs(cv - ocv).max()
        np.testing.assert_allclose(agent.t["CV_RMS_MEAN"].cpu().numpy(), orc.rms.mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(agent.t["CV_RMS_VAR"].cpu().numpy(), orc.rms.var.numpy(), rtol=1e-5, atol=1e-7)
        assert abs(c.rms_count - float(orc.rms.count)) < 1e-9
        # update_mu_sigma wrote the new mus back into the dataset
        np.testing.assert_allclose(agent.t["MB_MUS"].cpu().numpy().reshape(-1, 23), ds["mus"].numpy(), rtol=1e-3, atol=1e-3)
    finally:
        agent.close()


@pytest.mark.parametrize("mbsize,critic_coef", [(64, 4.0), (96, 1.0), (48, 1.0)])
def test_large_minibatch_update_matches_autograd_adam(mbsize, critic_coef):
    """minibatch_size > 8 (the insert policy's schedule, cfg/lego/ppo_continuous_insert.yaml: 4096, critic_coef 4) takes the GEMM-shaped
    step of sdxp_bigmb.hip (fp32 MFMA forward / data-gradient / weight-gradient GEMMs, explicit flat gradients, clip + Adam): the
    whole update phase against torch.autograd + Adam.  96 does not divide the 64-wide tiles: edge handling; 48 leaves the fused head kernel
    (k_big_heads: 32 rows per block) a half-empty second block."""
    n = 48
    agent, orc = make_pair(n, minibatch=mbsize, cv_minibatch=mbsize, critic_coef=critic_coef)
    try:
        assert agent.update_impl() == "gemm"
        g = torch.Generator().manual_seed(7)
        ds = rollout(agent, orc, n, g)
        agent.update()
        torch.cuda.synchronize()
        st = orc.update(ds)
        c = agent.ctrl()
        nsteps = 5 * (n * 8 // mbsize)
        assert c.n_mb == nsteps and c.ac_t == nsteps and c.cv_t == nsteps and c.mini_epoch == 5 and c.mb_index == 0
        np.testing.assert_allclose(c.sum_a_loss / nsteps, np.mean(st["a"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_c_loss / nsteps, np.mean(st["c"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_cv_loss / nsteps, np.mean(st["cv"]), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(c.sum_kl / nsteps, np.mean(st["kl"]), rtol=5e-3, atol=1e-5)
        np.testing.assert_allclose(c.ac_lr, orc.lr, rtol=1e-6)
        np.testing.assert_allclose(c.ac_gnorm, st["gnorm"][-1], rtol=2e-3)
        np.testing.assert_allclose(c.cv_gnorm, st["cv_gnorm"][-1], rtol=2e-3)
        ac = agent.t["AC_PARAMS"].cpu().numpy(); cv = agent.t["CV_PARAMS"].cpu().numpy()
        oa = orc.ac_flat().numpy(); ocv = orc.cv_flat().numpy()
        assert np.abs(ac - oa).max() < 2e-4, np.abs(ac - oa).max()
        assert np.abs(cv - ocv).max() < 5e-4, np.abs(cv - ocv).max()
        np.testing.assert_allclose(agent.t["CV_RMS_MEAN"].cpu().numpy(), orc.rms.mean.numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(agent.t["CV_RMS_VAR"].cpu().numpy(), orc.rms.var.numpy(), rtol=1e-5, atol=1e-7)
        assert abs(c.rms_count - float(orc.rms.count)) < 1e-9
        np.testing.assert_allclose(agent.t["MB_MUS"].cpu().numpy().reshape(-1, 23), ds["mus"].numpy(), rtol=1e-3, atol=1e-3)
    finally:
        agent.close()


@pytest.mark.parametrize("mbsize", [64, 96, 384])
def test_large_minibatch_bf16_gradients_against_fp32_path(mbsize):
    """one minibatch, same parameters and data: flat gradients of the bf16-MFMA step against the fp32-MFMA step of the same library
    (which the autograd oracle holds to 2e-4).  bf16 keeps 8 mantissa bits: the gradient of every network agrees to a few 1e-3 in
    norm, layer by layer."""
    n = 48
    a, orc = make_pair(n, minibatch=mbsize, cv_minibatch=mbsize, adaptive_lr=0, mixed_precision=1)
    b, _ = make_pair(n, minibatch=mbsize, cv_minibatch=mbsize, adaptive_lr=0)
    try:
        for ag in (a, b):
            rollout(ag, orc, n, torch.Generator().manual_seed(7))
            ag.backward(0, -1)
            ag.backward(0, 0)
        torch.cuda.synchronize()
        ca, cb = a.ctrl(), b.ctrl()
        for k in range(1, 7):
            np.testing.assert_allclose(ca.acc[k], cb.acc[k], rtol=2e-2, atol=1e-3 * mbsize)       # minibatch sums of the losses / KL
        for name in ("AC_GRADS", "CV_GRADS"):
            ga, gb = a.t[name].cpu().numpy().astype(np.float64), b.t[name].cpu().numpy().astype(np.float64)
            err = np.linalg.norm(ga - gb) / np.linalg.norm(gb)
            cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
            print("bf16 gradient %s mb %d: relative error %.4f cosine %.6f" % (name, mbsize, err, cos))
            assert err < 2e-2 and cos > 0.9995, (name, err, cos)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("mbsize", [64, 96])
def test_large_minibatch_bf16_policy_against_fp32_autograd(mbsize):
    """BASELINE.json configs[4] "bf16 policy" / SURVEY 8(d) config 5: `mixed_precision: True` runs the trunk GEMMs of the large-minibatch
    step on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 on their way into LDS, fp32 accumulation, fp32 master weights, Adam state
    and heads).  Against the fp32 torch.autograd + Adam oracle the epoch's statistics must agree to bf16 accuracy (8 mantissa bits:
    2e-2 relative on losses and gradient norms) and the parameter UPDATE must point the same way: Adam normalises every element's
    step to ~lr, so single elements whose gradient is noise-level may differ by a whole step while the update as a vector may not."""
    n = 48
    agent, orc = make_pair(n, minibatch=mbsize, cv_minibatch=mbsize, adaptive_lr=0, mixed_precision=1)
    try:
        assert agent.update_impl() == "gemm"
        p0a, p0c = agent.t["AC_PARAMS"].cpu().numpy().copy(), agent.t["CV_PARAMS"].cpu().numpy().copy()
        g = torch.Generator().manual_seed(7)
        ds = rollout(agent, orc, n, g)
        agent.update()
        torch.cuda.synchronize()
        st = orc.update(ds)
        c = agent.ctrl()
        nsteps = 5 * (n * 8 // mbsize)
        assert c.n_mb == nsteps and c.ac_t == nsteps
        np.testing.assert_allclose(c.sum_a_loss / nsteps, np.mean(st["a"]), rtol=3e-2, atol=2e-3)
        np.testing.assert_allclose(c.sum_c_loss / nsteps, np.mean(st["c"]), rtol=3e-2, atol=2e-3)
        np.testing.assert_allclose(c.sum_cv_loss / nsteps, np.mean(st["cv"]), rtol=3e-2, atol=2e-3)
        # (the gradient norm of the LAST step is not compared: after 30 steps the two trajectories have drifted apart and that single
        # number moves by tens of per cent with them; the update as a whole is what is held below)
        for name, got, want, p0 in (("ac", agent.t["AC_PARAMS"].cpu().numpy(), orc.ac_flat().numpy(), p0a),
                                    ("cv", agent.t["CV_PARAMS"].cpu().numpy(), orc.cv_flat().numpy(), p0c)):
            dg, dw = (got - p0).astype(np.float64), (want - p0).astype(np.float64)
            cos = float(dg @ dw / (np.linalg.norm(dg) * np.linalg.norm(dw)))
            rel = float(np.abs(dg - dw).mean() / np.abs(dw).mean())
            print("bf16 %s: update cosine %.5f, mean |diff| / mean |update| %.4f, max |diff| %.2e" % (name, cos, rel, np.abs(dg - dw).max()))
            # the gradient itself agrees to 4e-3 (test above); what is compared here is 20-30 Adam steps later.  The central value
            # steps with lr 1e-3 on a loss whose gradient is small after the first steps: elements whose gradient is noise-level
            # take different +-lr steps, measured: cosine 0.90 / 0.9997 for minibatch 64 / 96
            assert (cos > 0.99 and rel < 0.12) if name == "ac" else (cos > 0.85 and rel < 0.5), (name, cos, rel)
            assert np.abs(dg - dw).max() <= 2.0 * nsteps * 1e-3           # nobody moved further than Adam's per-step bound allows
        # and it IS the bf16 path: the fp32 path holds 2e-4 on the same data, this one must not (rounding is visible)
        assert np.abs(agent.t["AC_PARAMS"].cpu().numpy() - orc.ac_flat().numpy()).max() > 2e-5
    finally:
        agent.close()


def test_large_minibatch_explicit_path_equals_update():
    """sdxp_backward(0, mb) + sdxp_apply_flat-equivalent calls (the multi-rank order of calls at world size 1) == sdxp_update."""
    n = 32
    a1, orc = make_pair(n, seed=5, minibatch=64, cv_minibatch=64)
    a2, _ = make_pair(n, seed=5, minibatch=64, cv_minibatch=64)
    try:
        for ag in (a1, a2):
            g = torch.Generator().manual_seed(4)
            rollout(ag, orc, n, g)
        a1.update()
        a2.backward(0, -1)
        for me in range(5):
            for mb in range(n * 8 // 64):
                a2.backward(0, mb)
                a2.apply(0, float("-inf"))
                a2.apply(1, 0.0)
        torch.cuda.synchronize()
        np.testing.assert_allclose(a2.t["AC_PARAMS"].cpu().numpy(), a1.t["AC_PARAMS"].cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(a2.t["CV_PARAMS"].cpu().numpy(), a1.t["CV_PARAMS"].cpu().numpy(), rtol=0, atol=2e-6)
        assert a1.ctrl().ac_t == a2.ctrl().ac_t == 5 * (n * 8 // 64)
    finally:
        a1.close(); a2.close()


def test_explicit_gradient_path_equals_fused_path():
    """world_size 1: sdxp_backward / sdxp_apply (materialised flat gradients, the multi-rank path minus the all-reduce)
    must land on the same parameters as the fused rank-MB lazy-Adam path of sdxp_update."""
    n = 16
    a1, orc = make_pair(n, seed=5)
    a2, _ = make_pair(n, seed=5)
    try:
        np.testing.assert_array_equal(a1.t["AC_PARAMS"].cpu().numpy(), a2.t["AC_PARAMS"].cpu().numpy())
        for ag in (a1, a2):
            g = torch.Generator().manual_seed(4)
            rollout(ag, orc, n, g)
        a1.update()
        a2.backward(0, -1)
        for ep in range(5):
            for mb in range(n * 8 // 4):
                a2.backward(0, mb)
                a2.apply(0)
                a2.apply(1)
        torch.cuda.synchronize()
        c1, c2 = a1.ctrl(), a2.ctrl()
        assert c1.ac_t == c2.ac_t == 160 and c2.ac_pending == 0
        np.testing.assert_allclose(c1.ac_lr, c2.ac_lr, rtol=1e-6)
        d_ac = np.abs(a1.t["AC_PARAMS"].cpu().numpy() - a2.t["AC_PARAMS"].cpu().numpy()).max()
        d_cv = np.abs(a1.t["CV_PARAMS"].cpu().numpy() - a2.t["CV_PARAMS"].cpu().numpy()).max()
        # the two paths sum the same terms in different orders (and the persistent kernel uses v_rcp/v_sqrt in Adam): both are
        # held to the autograd oracle at 2e-4 / 5e-4 above, and to each other at half of that
        assert d_ac < 1e-4 and d_cv < 1e-4, (d_ac, d_cv)
        # the flat gradient buffers hold the last minibatch's gradients (a fully clipped minibatch has an all-zero
        # gradient, so only finiteness is asserted here; the values are covered by the parameter comparison above)
        g_ac, g_cv = a2.t["AC_GRADS"].cpu().numpy(), a2.t["CV_GRADS"].cpu().numpy()
        assert np.isfinite(g_ac).all() and np.isfinite(g_cv).all()
        np.testing.assert_allclose(c1.cv_gnorm, c2.cv_gnorm, rtol=2e-3)
    finally:
        a1.close(); a2.close()


def test_update_with_narrow_padded_observation_186():
    """BlockAssemblyOrient's 186-wide observation: the library pads the network input to 188 (two dead inputs, sdxp_config.obs_cols);
    the oracle runs a 188-input network on explicitly zero-padded rows.  Covers sdxp_act's padding and the persistent update kernel
    with an input narrower than its compiled maximum."""
    from seqdex_amd.ppo import SdxPPO, make_config
    n = 16
    cfg = make_config(n, obs_dim=186)
    assert cfg.obs_dim == 188 and cfg.obs_cols == 186
    agent = SdxPPO(n, config=cfg, seed=3)
    oc = dict(DEFAULT_CFG)
    oc.update(obs_dim=188, minibatch=cfg.minibatch, mini_epochs=cfg.mini_epochs, lr=cfg.lr, cv_lr=cfg.cv_lr, adaptive_lr=True)
    orc = PPOOracle(oc, seed=0)
    orc.load_flat(agent.t["AC_PARAMS"].cpu(), agent.t["CV_PARAMS"].cpu())
    try:
        g = torch.Generator().manual_seed(2)
        H = 8
        obs_l, st_l, rew_l, done_l = [], [], [], []
        buf = dict(actions=[], mus=[], sigmas=[], neglogp=[], values=[])
        for t in range(H):
            obs = torch.randn(n, 186, generator=g).clamp(-5, 5)
            obs188 = torch.cat([obs, torch.zeros(n, 2)], dim=1)
            st = torch.randn(n, 564, generator=g).clamp(-5, 5) * 2
            eps = torch.randn(n, 23, generator=g)
            dones = (torch.rand(n, generator=g) < 0.15).long()
            rew = torch.rand(n, generator=g)
            a = agent.act(t, obs.cuda(), st.cuda(), dones.cuda(), eps.cuda())
            agent.store_rewards(t, rew.cuda(), dones.cuda())
            r = orc.act(obs188, st, eps)
            np.testing.assert_allclose(a.cpu().numpy(), r["actions"].numpy(), rtol=2e-4, atol=2e-4)
            for k in buf:
                buf[k].append(r[k])
            obs_l.append(obs188); st_l.append(st); rew_l.append(rew); done_l.append(dones.float())
        last_st = torch.randn(n, 564, generator=g)
        last_done = (torch.rand(n, generator=g) < 0.15).long()
        agent.finish_rollout(last_st.cuda(), last_done.cuda())
        torch.cuda.synchronize()
        adv, ret = orc.gae(torch.stack(rew_l), torch.stack(buf["values"]), torch.stack(done_l), orc.values(last_st), last_done.float())
        flat = lambda x: torch.stack(x).transpose(0, 1).reshape(n * H, *x[0].shape[1:]).contiguous()
        ds = dict(obs=flat(obs_l), states=flat(st_l), actions=flat(buf["actions"]), mus=flat(buf["mus"]).clone(),
                  sigmas=flat(buf["sigmas"]).clone(), neglogp=flat(buf["neglogp"]), values=flat(buf["values"]),
                  returns=ret.transpose(0, 1).reshape(-1).contiguous())
        impl = agent.update_checked()
        orc.update(ds)
        ac, cv = agent.t["AC_PARAMS"].cpu().numpy(), agent.t["CV_PARAMS"].cpu().numpy()
        assert np.abs(ac - orc.ac_flat().numpy()).max() < 2e-4, (impl, np.abs(ac - orc.ac_flat().numpy()).max())
        assert np.abs(cv - orc.cv_flat().numpy()).max() < 5e-4
        np.testing.assert_allclose(agent.ctrl().ac_lr, orc.lr, rtol=1e-6)
    finally:
        agent.close()
