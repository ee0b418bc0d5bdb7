"""CPU: the names and the parameter order seqdex_amd/rlgames_checkpoint.py writes and reads against a plain torch.nn module with the
structure of rl_games' network builder (tests/helpers/rlgames_like.py - the names there come out of nn.Module registration)."""
import torch

from helpers import rlgames_like as RL


def test_state_dict_names_and_shapes_equal_the_rlgames_like_module():
    from seqdex_amd.rlgames_checkpoint import ac_parameter_order, flat_from_rlgames, rlgames_from_flat
    model, cvt = RL.build()
    sd, vsd = model.state_dict(), cvt.state_dict()
    g = torch.Generator().manual_seed(0)
    ac, cv = torch.randn(2131503, generator=g), torch.randn(1234945, generator=g)
    ours_m, ours_v = rlgames_from_flat(ac, cv, 396, 564, rms_mean=torch.zeros(564), rms_var=torch.ones(564), rms_count=1.0)
    assert set(ours_m) == set(sd), (set(ours_m) ^ set(sd))                      # exactly rl_games' keys: model.load_state_dict(strict) works
    assert set(ours_v) == set(vsd), (set(ours_v) ^ set(vsd))
    for k in sd:
        assert tuple(ours_m[k].shape) == tuple(sd[k].shape), k
    for k in vsd:
        assert tuple(ours_v[k].shape) == tuple(vsd[k].shape) and ours_v[k].dtype == vsd[k].dtype, k
    model.load_state_dict(ours_m, strict=True)
    cvt.load_state_dict(ours_v, strict=True)
    # parameter order of torch.optim.Adam(model.parameters()) = what the optimizer state is indexed by
    assert [n for n, _ in model.named_parameters()] == ac_parameter_order()
    # and a file written by the module reads back into the flat layout: first block = actor layer 0 in torch's W[out][in] order
    ac2, cv2, rms = flat_from_rlgames(model.state_dict(), cvt.state_dict(), 396, 564)
    assert torch.equal(ac2, ac) and torch.equal(cv2, cv) and rms[2] == 1.0


def test_torch_adam_state_maps_into_the_flat_moments():
    from seqdex_amd.rlgames_checkpoint import flat_from_rlgames, flat_from_torch_adam, torch_adam_from_flat
    model, _ = RL.build(seed=3)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, eps=1e-8)
    x = torch.randn(8, 396)
    for _ in range(3):
        mu, sigma, v = model(x)
        (mu.pow(2).mean() + v.pow(2).mean() + (sigma * 0.1).sum()).backward()
        opt.step(); opt.zero_grad()
    sd = opt.state_dict()
    m, v, step = flat_from_torch_adam(sd, 396)
    assert step == 3
    names = [n for n, _ in model.named_parameters()]
    exp_m = {n: sd["state"][i]["exp_avg"] for i, n in enumerate(names)}
    want, _, _ = flat_from_rlgames(exp_m, RL.build()[1].state_dict(), 396, 564)    # the same slicing as for the parameters themselves
    assert torch.equal(m, want) and float(m.abs().max()) > 0
    back = torch_adam_from_flat(m, v, step, 3e-4, 396)
    opt2 = torch.optim.Adam(model.parameters(), lr=1.0)
    opt2.load_state_dict(back)                                                   # rl_games: self.optimizer.load_state_dict(weights['optimizer'])
    for i in range(len(names)):
        assert torch.equal(opt2.state_dict()["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    assert opt2.state_dict()["param_groups"][0]["lr"] == 3e-4
