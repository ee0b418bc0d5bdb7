"""CPU tests of the host-side mirrors: flag/config semantics (utils/config.py), scene constants, VecTask surface."""
import os

import numpy as np
import yaml
import pytest


def test_get_args_and_load_cfg_overrides():
    from seqdex_amd.config import get_args, load_cfg
    a = get_args(["--task=BlockAssemblyGraspSim", "--num_envs=1024", "--seed", "7", "--max_iterations", "3", "--headless",
                  "--minibatch_size", "64", "--rl_device", "cpu"])
    assert a.train and not a.play and a.checkpoint == "" and a.device == "cuda"
    cfg, cfg_train, logdir = load_cfg(a)
    assert cfg["env"]["numEnvs"] == 1024 and cfg["seed"] == 7
    assert cfg_train["params"]["config"]["max_epochs"] == 3
    assert cfg_train["params"]["config"]["num_actors"] == 1024
    assert cfg_train["params"]["config"]["minibatch_size"] == 4      # --minibatch_size is parsed but never applied (TR)
    assert logdir == "logs/BlockAssemblyGraspSim"
    b = get_args(["--test", "--checkpoint", "x.pth"])
    assert b.play and not b.train and b.checkpoint == "x.pth"
    with pytest.raises(SystemExit):
        get_args(["--task=NotATask"])
    with pytest.raises(SystemExit):
        get_args(["--pipeline", "cpu"])


def test_yaml_values_match_reference_hyperparameters():
    import yaml, os
    from seqdex_amd import config
    tr = yaml.safe_load(open(os.path.join(config.HERE, config.TRAIN_CFG["BlockAssemblyGraspSim"])))["params"]
    c = tr["config"]
    assert (c["horizon_length"], c["minibatch_size"], c["mini_epochs"]) == (8, 4, 5)            # YG:49-51
    assert (c["gamma"], c["tau"], float(c["learning_rate"]), c["e_clip"]) == (0.99, 0.95, 3e-4, 0.1)
    assert c["central_value_config"]["normalize_input"] is True and float(c["central_value_config"]["learning_rate"]) == 1e-3
    assert tr["network"]["mlp"]["units"] == [1024, 512, 256] and tr["network"]["space"]["continuous"]["fixed_sigma"] is True
    env = yaml.safe_load(open(os.path.join(config.HERE, config.TASK_CFG["BlockAssemblyGraspSim"])))
    assert env["env"]["episodeLength"] == 150 and env["sim"]["substeps"] == 2
    assert env["sim"]["physx"]["num_position_iterations"] == 16 and env["sim"]["physx"]["contact_offset"] == 0.002


def test_scene_constants(scene):
    assert scene.link_names[7] == "panda_link7" and len(scene.link_names) == 24
    assert [scene.link_names[i] for i in scene.fingertip_bodies] == ["link_3.0", "link_7.0", "link_11.0", "link_15.0"]
    np.testing.assert_allclose(scene.lower[3], -3.0718); np.testing.assert_allclose(scene.upper[3], -0.0698)
    assert [scene.seg_index(i) - 9 for i in range(8)] == [0, 1, 2, 0, 0, 5, 6, 0]                # GS:962-965
    d = scene.to_desc()
    assert d.n_rbox == 40 and d.n_static == 8 and d.substeps == 2 and d.solver_iters == 16     # 31 shapes + 8 fingertip slabs + 1 palm slab
    # collision compounds (round 5): every brick type two slabs of its hull inside the bounding box, the base plate body + 16 studs of 2 boxes
    for t in range(8):
        assert d.brick_nsub[t] == 2 and 6 <= d.hollow_nsub[t] <= 8
        c, h = np.array(d.brick_center[t]), np.array(d.brick_half[t])
        for k in range(2):
            sc_, sh_ = np.array(d.brick_sub_center[t][k]), np.array(d.brick_sub_half[t][k])
            assert (sc_ - sh_ >= c - h - 1e-6).all() and (sc_ + sh_ <= c + h + 1e-6).all()
        assert abs(d.brick_sub_center[t][1][2] + d.brick_sub_half[t][1][2] - 0.0387) < 2e-4        # the upper slab ends at the stud tops
        assert abs(d.brick_sub_center[t][0][2] - d.brick_sub_half[t][0][2] + 0.01875) < 1e-4       # the lower one starts at the body's bottom
    assert d.seg_hollow == 0 and d.static_sub_n[7] == 33 and d.n_static_sub == 7 + 33 and [d.static_sub_n[s] for s in range(7)] == [1] * 7
    studs = np.array([list(d.static_sub_center[d.static_sub_first[7] + 1 + 2 * i]) for i in range(16)])
    assert sorted(set(np.round(studs[:, 0] - 0.25, 4))) == [-0.045, -0.015, 0.015, 0.045]        # the 4 x 4 stud grid, 30 mm pitch
    assert abs(d.kp[0] - 400) < 1e-6 and abs(d.kp[7] - 50) < 1e-6 and abs(d.effort[7] - 5) < 1e-6   # GS:580-590
    masses = [b["mass"] for b in scene.brick_types]
    assert 0.02 < min(masses) and max(masses) < 0.12                                              # 567 kg/m3 x hull volume


def test_vec_task_surface_without_gpu():
    from seqdex_amd.vec_task_rlgames import Box, RLgamesVecTaskPython, VecTask
    class T:  # minimal task stand-in
        num_envs, num_obs, num_states, num_actions, device = 4, 396, 564, 23, "cpu"
    env = VecTask(T(), "cpu")
    info = env.get_env_info()
    assert info["agents"] == 1 and info["action_space"].shape == (23,) and info["observation_space"].shape == (396,)
    assert info["state_space"].shape == (564,) and env.num_envs == 4 and env.num_acts == 23 and env.num_obs == 396
    assert env.get_number_of_agents == 1 and env.has_action_masks() is False
    assert float(info["action_space"].low[0]) == -1.0 and np.isinf(info["observation_space"].high[0])
    with pytest.raises(ValueError):
        VecTask(T(), "cpu", clip_observations=3.0)
    assert issubclass(RLgamesVecTaskPython, VecTask)


def test_insert_sim_scene_desc_places_three_plate_variants(scene):
    """BlockAssemblyInsertSim (task_kind 2): plate actor at (0.25, -0.2, 0.618) (IS:1438-1440), static box 7 = stud-less plate body
    whose z extent depends on env % 3 (4x4x{1,2,4}, IS:971-977); a seated brick's origin is exactly at the insertion site."""
    d = scene.to_desc(task_kind=2)
    assert d.abi_version == __import__("seqdex_amd._abi", fromlist=["x"]).SDX_ABI_VERSION and d.static_var_slot == 7 and d.n_static == 8
    np.testing.assert_allclose(list(d.base_plate_pos), [0.25, -0.2, 0.618], atol=1e-7)
    np.testing.assert_allclose(list(d.static_center[7])[:2], [0.25, -0.2], atol=1e-7)
    assert abs(d.static_half[7][0] - (0.06 + scene.INSERT_PLATE_MARGIN)) < 1e-6
    assert d.seg_hollow == 1 and list(d.static_var_row) == [7, 8, 9] and d.n_static_sub == 7 + 3 * 33
    for k in range(3):
        row = d.static_var_row[k]
        assert d.static_sub_n[row] == 33
        b0 = d.static_sub_first[row]                                           # box 0 = the plate's body, 1.. = its studs (shaft, tip)
        top = d.static_sub_center[b0][2] + d.static_sub_half[b0][2]
        assert abs(top + 0.01875 - (0.618 + 0.0375 * (1 + k))) < 6e-4          # body top + half a brick body = site height
        assert abs(d.static_sub_center[b0][2] - d.static_sub_half[b0][2] - (0.618 - 0.01875)) < 1e-6
        assert abs(d.static_sub_center[b0 + 2][2] + d.static_sub_half[b0 + 2][2] - (top + 0.0387 - 0.01875)) < 1e-4   # stud tips 19.95 mm above
        assert abs(d.static_center[row][2] + d.static_half[row][2] - (top + 0.0387 - 0.01875)) < 1e-4                 # ... = the bounding box's top
    g = scene.to_desc()                                                       # GraspSim keeps its single plate
    assert g.static_var_slot == -1 and abs(g.static_center[7][1] + 0.19) < 1e-6


def test_launcher_maps_the_four_tasks():
    from seqdex_amd import config
    assert set(config.TASK_CFG) == {"BlockAssemblyGraspSim", "BlockAssemblyOrient", "BlockAssemblyInsertSim", "BlockAssemblySearch"}
    for t, rel in config.TASK_CFG.items():
        cfg = yaml.safe_load(open(os.path.join(os.path.dirname(config.__file__), rel)))
        assert cfg["env"]["episodeLength"] == {"BlockAssemblyGraspSim": 150, "BlockAssemblyOrient": 75, "BlockAssemblyInsertSim": 125,
                                                "BlockAssemblySearch": 75}[t]


def test_rlgames_checkpoint_layout_round_trip():
    """flat parameter buffers <-> rl_games 1.5.2 state_dict names: sizes add up to the parameter counts of
    SURVEY.md section 2b, the round trip is exact, wrapper prefixes are ignored, wrong shapes are refused, and a checkpoint of the
    reference's 186-wide Orient observation loads into the library's 188-wide (zero-padded) first layer."""
    import torch
    from seqdex_amd.rlgames_checkpoint import flat_from_rlgames, rlgames_from_flat
    g = torch.Generator().manual_seed(0)
    ac, cv = torch.randn(2131503, generator=g), torch.randn(1234945, generator=g)
    model, vf = rlgames_from_flat(ac, cv, 396, 564, rms_mean=torch.arange(564.0), rms_var=torch.ones(564) * 2, rms_count=77.0)
    assert tuple(model["a2c_network.actor_mlp.0.weight"].shape) == (1024, 396) and tuple(model["a2c_network.sigma"].shape) == (23,)
    assert tuple(model["a2c_network.critic_mlp.4.weight"].shape) == (256, 512) and tuple(model["a2c_network.value.weight"].shape) == (1, 256)
    assert tuple(vf["model.a2c_network.actor_mlp.0.weight"].shape) == (1024, 564)    # no `separate` key in YG:86-95
    assert torch.equal(model["a2c_network.actor_mlp.0.weight"].reshape(-1), ac[:1024 * 396])          # torch layout W[out][in], first block
    wrapped = {"module." + k: v for k, v in model.items()}
    ac2, cv2, rms = flat_from_rlgames(wrapped, vf, 396, 564)
    assert torch.equal(ac2, ac) and torch.equal(cv2, cv)
    assert rms[2] == 77.0 and torch.equal(rms[0], torch.arange(564.0).double())
    bad = dict(model)
    bad["a2c_network.mu.weight"] = torch.zeros(22, 256)
    with pytest.raises(ValueError):
        flat_from_rlgames(bad, vf, 396, 564)
    m186, v188 = rlgames_from_flat(torch.randn(2131503 - 2 * 1024 * 210, generator=g), cv, 186, 564)
    acp, _, _ = flat_from_rlgames(m186, v188, 188, 564, obs_cols=186)
    w0 = acp[:1024 * 188].reshape(1024, 188)
    assert torch.equal(w0[:, :186], m186["a2c_network.actor_mlp.0.weight"]) and not w0[:, 186:].any()
    # saving from the library's padded widths cuts the first layers and the running statistics to the real ones (what rl_games
    # builds for that task: Orient obs 186, InsertSim states 188) and loading such a file zero-fills the padded columns again
    acp188 = torch.randn(2131503 - 2 * 1024 * 208, generator=g)
    mo, vo = rlgames_from_flat(acp188, cv, 188, 564, rms_mean=torch.arange(564.0), rms_var=torch.ones(564), rms_count=5.0,
                               obs_cols=186, state_cols=188)
    assert tuple(mo["a2c_network.actor_mlp.0.weight"].shape) == (1024, 186) and tuple(mo["a2c_network.critic_mlp.0.weight"].shape) == (1024, 186)
    assert tuple(vo["model.a2c_network.actor_mlp.0.weight"].shape) == (1024, 188) and vo["model.running_mean_std.running_mean"].numel() == 188
    ac3, cv3, rms3 = flat_from_rlgames(mo, vo, 188, 564, obs_cols=186, state_cols=188)
    w3 = ac3[:1024 * 188].reshape(1024, 188)
    assert torch.equal(w3[:, :186], acp188[:1024 * 188].reshape(1024, 188)[:, :186]) and not w3[:, 186:].any()
    c3 = cv3[:1024 * 564].reshape(1024, 564)
    assert torch.equal(c3[:, :188], cv[:1024 * 564].reshape(1024, 564)[:, :188]) and not c3[:, 188:].any()
    assert rms3[2] == 5.0 and torch.equal(rms3[0][:188], torch.arange(188.0).double()) and not rms3[0][188:].any()


def test_pile_pickle_round_trip(tmp_path):
    """the reference hands pile states over as a pickled list of 8 tensors [slots, 132, 13] filled from index 0 (GS:412-413, SE:1349-1350);
    written and read back, with the reference's zero-padded tail trimmed to the filled slots"""
    import pickle
    import numpy as np
    import torch
    from seqdex_amd.piles import load_pile_pickle, save_pile_pickle
    rng = np.random.default_rng(0)
    piles = rng.normal(size=(8, 5, 132, 13)).astype(np.float32)
    piles[..., 3:7] /= np.linalg.norm(piles[..., 3:7], axis=-1, keepdims=True)
    p = str(tmp_path / "piles.pkl")
    save_pile_pickle(p, piles)
    lst = pickle.load(open(p, "rb"))
    assert isinstance(lst, list) and len(lst) == 8 and all(torch.is_tensor(x) and tuple(x.shape) == (5, 132, 13) for x in lst)
    np.testing.assert_array_equal(load_pile_pickle(p), piles)
    # the reference's buffers are larger than what was filled: slots past the fill index are zero; groups may be filled unevenly
    big = [torch.zeros(40, 132, 13) for _ in range(8)]
    for t in range(8):
        big[t][:3 + t % 2] = torch.as_tensor(piles[t, :3 + t % 2])
    pickle.dump(big, open(p, "wb"))
    got = load_pile_pickle(p)
    assert got.shape == (8, 3, 132, 13)
    np.testing.assert_array_equal(got, piles[:, :3])
    save_pile_pickle(p, piles, counts=[2] * 8)
    assert load_pile_pickle(p).shape == (8, 2, 132, 13)


def test_scene_desc_round3_defaults(scene):
    """the solver / task constants round 3 added to sdx_scene_desc carry the reference's values (or this engine's documented defaults):
    warm start 0.8 ramped over 16 solves, T-value gates 0.99 (OR:1203) / 0.8 (GS:1406), link angular damping
    0.01 (GS:546), and a spawn lattice whose lowest brick layer starts 2 mm above the floor slab instead of inside it (GS:737-742)."""
    d = scene.to_desc()
    assert abs(d.warm_start - 0.8) < 1e-7 and d.warm_age == 16.0
    assert abs(d.orient_tvalue_gate - 0.99) < 1e-7 and abs(d.grasp_tvalue_gate - 0.8) < 1e-7
    assert abs(d.robot_angular_damping - 0.01) < 1e-9
    floor_top = scene.statics[6]["center"][2] + scene.statics[6]["half"][2]
    lows = []
    for i, fs in enumerate(scene.raw["free_spawn"]):
        t = scene.brick_types[fs["type"]]
        lows.append(d.free_spawn_pos[i][2] + t["center"][2] - t["half"][2])
        assert abs(d.free_spawn_pos[i][0] - fs["pos"][0]) < 1e-6 and abs(d.free_spawn_pos[i][1] - fs["pos"][1]) < 1e-6   # only z moves
    assert abs(min(lows) - (floor_top + 0.002)) < 1e-5
    assert 0.06 < scene.spawn_lift < 0.07
    # overrides reach the descriptor (what BlockAssemblyOrient(tvalue_gate=...) / BlockAssemblyGraspSim(harvest_tvalue_gate=...) use)
    d2 = scene.to_desc(orient_tvalue_gate=0.5, grasp_tvalue_gate=0.28, warm_start=0.0)
    assert abs(d2.orient_tvalue_gate - 0.5) < 1e-7 and abs(d2.grasp_tvalue_gate - 0.28) < 1e-7 and d2.warm_start == 0.0


def test_chain_fills_missing_pile_groups_from_settled_piles(monkeypatch):
    """evaluation.py::fill_missing_pile_groups on CPU tensors (the settled piles come from a stand-in generator here; the GPU suite runs the
    real one): complete groups keep their harvested states, K = the smallest complete fill, more than two empty groups are not papered over"""
    import numpy as np
    import torch
    import seqdex_amd.piles as piles_mod
    from seqdex_amd.scripts.evaluation import fill_missing_pile_groups
    calls = []

    def fake_generate(per_type=8, steps=150, device="cpu", seed=22):
        calls.append((per_type, seed))
        return np.full((8, per_type, 132, 13), 7.0, np.float32)

    monkeypatch.setattr(piles_mod, "generate_piles", fake_generate)
    harvest = torch.arange(8 * 12 * 132 * 13, dtype=torch.float32).reshape(8, 12, 132, 13)
    piles, lacking = fill_missing_pile_groups(harvest, torch.tensor([12, 30, 9, 12, 12, 12, 0, 3]), 8, seed=5)
    assert lacking == [6, 7] and tuple(piles.shape) == (8, 9, 132, 13) and calls == [(9, 5)]
    for t in range(8):
        assert torch.equal(piles[t], torch.full((9, 132, 13), 7.0) if t in lacking else harvest[t, :9])
    assert fill_missing_pile_groups(harvest, torch.full((8,), 12), 8, seed=5) == (None, [])
    assert fill_missing_pile_groups(harvest, torch.tensor([0, 0, 0, 12, 12, 12, 12, 12]), 8, seed=5) == (None, [])
    assert calls == [(9, 5)]


def test_ring_rows_returns_the_appends_in_serial_step_env_order():
    """SdxSim.ring_rows: ring slots are claimed with atomics, i.e. in hardware order; the (step << 24 | env) keys
    written with every append give back the order a serial loop over steps and envs produces - what makes the chain identical run to run"""
    import torch
    from seqdex_amd.sim import SdxSim
    key = lambda step, env: (step << 24) | env
    keys = torch.tensor([key(3, 7), key(1, 900), key(3, 2), key(1, 4), key(2, 0), 0, 0, 0], dtype=torch.int64)
    rows = torch.arange(8, dtype=torch.float32).view(8, 1).repeat(1, 3)                     # row i carries the value i
    got = SdxSim.ring_rows(None, rows, keys, 5)
    assert got[:, 0].tolist() == [3.0, 1.0, 4.0, 2.0, 0.0]                                  # (1,4) (1,900) (2,0) (3,2) (3,7)
    assert SdxSim.ring_rows(None, rows, keys, 0).shape == (0, 3)
    import types
    me = types.SimpleNamespace(ring_wrapped=False)
    assert SdxSim.ring_rows(me, rows, keys, 5).shape == (5, 3) and me.ring_wrapped is False
    full = SdxSim.ring_rows(me, rows, torch.arange(8, 0, -1), 100)                         # more appends than slots: every slot is filled
    assert full[:, 0].tolist() == [7.0, 6.0, 5.0, 4.0, 3.0, 2.0, 1.0, 0.0]
    assert me.ring_wrapped is True                                                          # ... and the caller can tell that the serial order is gone
    g = SdxSim.ring_rows(None, rows, keys, 5)
    g[0, 0] = -1.0
    assert rows[3, 0] == 3.0                                                                # a copy, not a view of the ring
