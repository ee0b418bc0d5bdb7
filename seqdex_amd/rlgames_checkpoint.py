"""rl_games 1.5.2 checkpoint layout <-> the library's flat parameter buffers.

rl_games' `A2CBase.get_full_state_weights()` stores `model` = state_dict of `ModelA2CContinuousLogStd.Network` (network builder
`actor_critic`, `separate: True`, `fixed_sigma: True`) and, for the central value, `assymetric_vf_nets` = state_dict of
`CentralValueTrain` (its own `actor_critic` network with `central_value: True` + the input RunningMeanStd).  The flat buffers of
libseqdex_hip.so use the torch layout W[out][in] in the order of `SdxpOff` / `SdxpCOff` (csrc/sdxp_types.h), so the conversion is
slicing and naming only.  rl_games is not installed here and the reference ships no checkpoint: the key names below are the
upstream ones as recalled (PARITY UNPINNED); `flat_from_rlgames` therefore matches keys by suffix and checks every shape.

Structure the names come from (rl_games `A2CBuilder.Network.__init__`): `actor_cnn`, `critic_cnn` (empty),
`actor_mlp = Sequential(Linear, ELU, Linear, ELU, Linear, ELU)` (Linear at indices 0 / 2 / 4), `critic_mlp` the same when
`separate: True` (YG:10) and EMPTY otherwise, then `value = Linear`, and for a continuous space `mu = Linear`, `sigma = Parameter`.
The central-value network (YG:86-95) has `central_value: True`, no `space` block and no `separate` key: its trunk is therefore
`actor_mlp` and it has no `mu` / `sigma`; `CentralValueTrain.model` wraps it (`model.a2c_network.*`) and the input RunningMeanStd
sits beside it (`model.running_mean_std.*` in 1.5.x, `running_mean_std.*` before).  Rounds 2-3 wrote the central-value trunk as
`critic_mlp`; such files are still read.  tests/test_gpu_rlgames_checkpoint.py builds a plain torch.nn module of exactly this
structure, saves ITS state_dict / optimizer.state_dict() the way rl_games does and checks that a restored handle computes the
module's outputs.
"""
import numpy as np
import torch

AC_KEYS = [("a2c_network.actor_mlp.%d", 3), ("a2c_network.mu", 1), ("a2c_network.sigma", 0),
           ("a2c_network.critic_mlp.%d", 3), ("a2c_network.value", 1)]


def _layers(in_dim, units):
    dims, d = [], in_dim
    for u in units:
        dims.append((u, d))
        d = u
    return dims


def rlgames_from_flat(ac_flat, cv_flat, obs_dim, state_dim, act_dim=23, units=(1024, 512, 256), rms_mean=None, rms_var=None,
                      rms_count=None, obs_cols=None, state_cols=None):
    """flat actor-critic / central-value parameter vectors -> (model_state_dict, assymetric_vf_nets_state_dict).  obs_cols /
    state_cols: the real input widths when the library pads its network inputs (see flat_from_rlgames): first-layer weights and
    the running statistics are cut to them, so rl_games can load the file for that task."""
    ac, cv = torch.as_tensor(ac_flat).float().cpu(), torch.as_tensor(cv_flat).float().cpu()
    model, o = {}, 0
    obs_cols, state_cols = obs_cols or obs_dim, state_cols or state_dim

    def take(buf, off, shape):
        n = int(np.prod(shape))
        return buf[off:off + n].reshape(shape).clone(), off + n

    for i, (out, inn) in enumerate(_layers(obs_dim, units)):          # Sequential(Linear, ELU, ...): Linear at indices 0, 2, 4
        model["a2c_network.actor_mlp.%d.weight" % (2 * i)], o = take(ac, o, (out, inn))
        model["a2c_network.actor_mlp.%d.bias" % (2 * i)], o = take(ac, o, (out,))
    model["a2c_network.mu.weight"], o = take(ac, o, (act_dim, units[-1]))
    model["a2c_network.mu.bias"], o = take(ac, o, (act_dim,))
    model["a2c_network.sigma"], o = take(ac, o, (act_dim,))
    for i, (out, inn) in enumerate(_layers(obs_dim, units)):
        model["a2c_network.critic_mlp.%d.weight" % (2 * i)], o = take(ac, o, (out, inn))
        model["a2c_network.critic_mlp.%d.bias" % (2 * i)], o = take(ac, o, (out,))
    model["a2c_network.value.weight"], o = take(ac, o, (1, units[-1]))
    model["a2c_network.value.bias"], o = take(ac, o, (1,))
    assert o == ac.numel(), (o, ac.numel())
    for k in ("a2c_network.actor_mlp.0.weight", "a2c_network.critic_mlp.0.weight"):
        model[k] = model[k][:, :obs_cols].clone()
    vf, o = {}, 0
    for i, (out, inn) in enumerate(_layers(state_dim, units)):
        vf["model.a2c_network.actor_mlp.%d.weight" % (2 * i)], o = take(cv, o, (out, inn))
        vf["model.a2c_network.actor_mlp.%d.bias" % (2 * i)], o = take(cv, o, (out,))
    vf["model.a2c_network.value.weight"], o = take(cv, o, (1, units[-1]))
    vf["model.a2c_network.value.bias"], o = take(cv, o, (1,))
    assert o == cv.numel(), (o, cv.numel())
    vf["model.a2c_network.actor_mlp.0.weight"] = vf["model.a2c_network.actor_mlp.0.weight"][:, :state_cols].clone()
    if rms_mean is not None:
        vf["model.running_mean_std.running_mean"] = torch.as_tensor(rms_mean).double().cpu()[:state_cols].clone()
        vf["model.running_mean_std.running_var"] = torch.as_tensor(rms_var).double().cpu()[:state_cols].clone()
        vf["model.running_mean_std.count"] = torch.tensor(float(rms_count if rms_count is not None else 0.0), dtype=torch.float64)
    return model, vf


def _find(sd, suffix, shape):
    hits = [k for k in sd if k.endswith(suffix)]
    if len(hits) != 1:
        raise KeyError("rl_games checkpoint: expected exactly one key ending in %r, found %r" % (suffix, hits))
    v = torch.as_tensor(sd[hits[0]]).float().cpu()
    if tuple(v.shape) != tuple(shape):
        raise ValueError("rl_games checkpoint: %s has shape %s, this network needs %s" % (hits[0], tuple(v.shape), tuple(shape)))
    return v.reshape(-1)


def flat_from_rlgames(model, vf, obs_dim, state_dim, act_dim=23, units=(1024, 512, 256), obs_cols=None, state_cols=None):
    """(model state_dict, assymetric_vf_nets state_dict) -> (ac_flat, cv_flat, rms or None).  Keys are matched by suffix (DataParallel
    / wrapper prefixes are ignored).  obs_cols / state_cols: width of the checkpoint's first layers when the library pads its network
    inputs (Orient 186 -> 188 observation columns, InsertSim 188 -> 564 state columns): the missing input columns get zero weights."""
    def first(sd, name, out, inn, cols):
        cols = cols or inn
        w = _find(sd, name + ".weight", (out, cols)).reshape(out, cols)
        if cols != inn:
            w = torch.cat([w, torch.zeros(out, inn - cols)], dim=1)
        return w.reshape(-1)

    ac = _ac_flat_only(model, obs_dim, act_dim, units, obs_cols)
    parts = []
    # the central-value trunk is `actor_mlp` in an rl_games file (no `separate` key, see the header); `critic_mlp` in files of rounds 2-3
    cv_trunk = "actor_mlp" if any(k.endswith("a2c_network.actor_mlp.0.weight") for k in vf) else "critic_mlp"
    for i, (out, inn) in enumerate(_layers(state_dim, units)):
        name = "a2c_network.%s.%d" % (cv_trunk, 2 * i)
        parts.append(first(vf, name, out, inn, state_cols) if i == 0 else _find(vf, name + ".weight", (out, inn)))
        parts.append(_find(vf, name + ".bias", (out,)))
    parts.append(_find(vf, "a2c_network.value.weight", (1, units[-1])))
    parts.append(_find(vf, "a2c_network.value.bias", (1,)))
    cv = torch.cat(parts)
    rms = None
    mean_keys = [k for k in vf if k.endswith("running_mean_std.running_mean")]
    if mean_keys:
        pre = mean_keys[0][:-len("running_mean")]
        cols = state_cols or state_dim
        mean, var = torch.zeros(state_dim, dtype=torch.float64), torch.ones(state_dim, dtype=torch.float64)
        rm = torch.as_tensor(vf[pre + "running_mean"]).double().reshape(-1)
        rv = torch.as_tensor(vf[pre + "running_var"]).double().reshape(-1)
        if rm.numel() < cols or rv.numel() != rm.numel():
            raise ValueError("checkpoint running_mean_std has %d / %d entries, the central value network reads %d state columns"
                             % (rm.numel(), rv.numel(), cols))
        mean[:cols] = rm[:cols]
        var[:cols] = rv[:cols]
        cnt = float(torch.as_tensor(vf[pre + "count"])) if (pre + "count") in vf else 0.0
        rms = (mean, var, cnt)
    return ac, cv, rms


# ---------------------------------------------------------------------------------------------------------------------------------
# torch.optim.Adam state of the actor-critic (rl_games: weights['optimizer'] = self.optimizer.state_dict(), params = list(model.parameters()))
def ac_parameter_order(units=(1024, 512, 256)):
    """names of the actor-critic parameters in `model.parameters()` order: nn.Module yields a module's OWN parameters before those of
    its children, so the `sigma` Parameter of the network comes first, then the children in registration order (header).  `units`: the
    trunk's layer widths (one Linear per entry); separate critic and fixed sigma, as the shipped YAMLs build it."""
    names = ["a2c_network.sigma"]
    for trunk in ("actor_mlp", "critic_mlp"):
        for i in range(len(units)):
            names += ["a2c_network.%s.%d.weight" % (trunk, 2 * i), "a2c_network.%s.%d.bias" % (trunk, 2 * i)]
    return names + ["a2c_network.value.weight", "a2c_network.value.bias", "a2c_network.mu.weight", "a2c_network.mu.bias"]


def torch_adam_from_flat(m_flat, v_flat, step, lr, obs_dim, act_dim=23, units=(1024, 512, 256), obs_cols=None):
    """flat Adam moments of the actor-critic -> torch.optim.Adam.state_dict() over `model.parameters()` (what rl_games stores and loads)"""
    mn, _ = rlgames_from_flat(m_flat, _dummy_cv(units), obs_dim, 4, act_dim, units, obs_cols=obs_cols)
    vn, _ = rlgames_from_flat(v_flat, _dummy_cv(units), obs_dim, 4, act_dim, units, obs_cols=obs_cols)
    order = ac_parameter_order(units)
    state = {i: {"step": torch.tensor(float(step)), "exp_avg": mn[k], "exp_avg_sq": vn[k]} for i, k in enumerate(order)}
    group = {"lr": float(lr), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(order)))}
    return {"state": state, "param_groups": [group]}


def _dummy_cv(units):
    n, d = 0, 4
    for u in units:
        n += u * d + u
        d = u
    return torch.zeros(n + d + 1)


def flat_from_torch_adam(opt_sd, obs_dim, act_dim=23, units=(1024, 512, 256), obs_cols=None):
    """torch.optim.Adam.state_dict() of the actor-critic (parameters in `ac_parameter_order(units)`) -> (m_flat, v_flat, step); None when the
    state is EMPTY (a checkpoint saved before the first optimiser step).  A non-empty state that does not have one entry per parameter of
    this layout (another network shape, a shared critic, a learned sigma) cannot be mapped: ValueError - the caller decides whether to go
    on with fresh moments, it is not done silently."""
    order = ac_parameter_order(units)
    st = opt_sd.get("state", {})
    if len(st) == 0:
        return None
    ids = opt_sd["param_groups"][0]["params"] if opt_sd.get("param_groups") else sorted(st)
    if len(st) != len(order) or len(ids) != len(order):
        raise ValueError("optimizer state with %d entries (%d parameter ids) does not fit the actor-critic layout of units %s (%d parameters: "
                         "separate critic, fixed sigma)" % (len(st), len(ids), tuple(units), len(order)))
    m = {k: st[i]["exp_avg"] for k, i in zip(order, ids)}
    v = {k: st[i]["exp_avg_sq"] for k, i in zip(order, ids)}
    shapes = _ac_shapes(obs_dim, act_dim, units, obs_cols)
    for k in order:
        if tuple(m[k].shape) != shapes[k] or tuple(v[k].shape) != shapes[k]:
            raise ValueError("optimizer state of %s has shape %s, the layout needs %s" % (k, tuple(m[k].shape), shapes[k]))
    mf = _ac_flat_only(m, obs_dim, act_dim, units, obs_cols)
    vf = _ac_flat_only(v, obs_dim, act_dim, units, obs_cols)
    step = int(round(float(torch.as_tensor(st[ids[0]]["step"]))))
    return mf, vf, step


def _ac_shapes(obs_dim, act_dim, units, obs_cols):
    """parameter name -> shape of the actor-critic in the rl_games layout (first layers `obs_cols` wide)"""
    out = {"a2c_network.sigma": (act_dim,), "a2c_network.mu.weight": (act_dim, units[-1]), "a2c_network.mu.bias": (act_dim,),
           "a2c_network.value.weight": (1, units[-1]), "a2c_network.value.bias": (1,)}
    for trunk in ("actor_mlp", "critic_mlp"):
        for i, (o, inn) in enumerate(_layers(obs_dim, units)):
            out["a2c_network.%s.%d.weight" % (trunk, 2 * i)] = (o, (obs_cols or inn) if i == 0 else inn)
            out["a2c_network.%s.%d.bias" % (trunk, 2 * i)] = (o,)
    return out


def _ac_flat_only(model, obs_dim, act_dim, units, obs_cols):
    """the actor-critic half of flat_from_rlgames"""
    def first(name, out, inn, cols):
        cols = cols or inn
        w = _find(model, name + ".weight", (out, cols)).reshape(out, cols)
        if cols != inn:
            w = torch.cat([w, torch.zeros(out, inn - cols)], dim=1)
        return w.reshape(-1)
    parts = []
    for trunk, head, hout in (("actor_mlp", "mu", act_dim), ("critic_mlp", "value", 1)):
        for i, (out, inn) in enumerate(_layers(obs_dim, units)):
            name = "a2c_network.%s.%d" % (trunk, 2 * i)
            parts.append(first(name, out, inn, obs_cols) if i == 0 else _find(model, name + ".weight", (out, inn)))
            parts.append(_find(model, name + ".bias", (out,)))
        parts.append(_find(model, "a2c_network.%s.weight" % head, (hout, units[-1])))
        parts.append(_find(model, "a2c_network.%s.bias" % head, (hout,)))
        if head == "mu":
            parts.append(_find(model, "a2c_network.sigma", (act_dim,)))
    return torch.cat(parts)
