"""-m gpu: the transition-value trainer on the HIP path (csrc/sdx_tvtrain.hip through the sdxtv_* C ABI) against the golden vectors
captured with the reference's GraspInsertTValue class under BCEWithLogitsLoss + Adam (tests/golden/TV1_train.npz), the torch oracle,
and - end to end - on datasets that the InsertSim reset kernels logged."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import tvalue_train_oracle as TO  # noqa: E402

NAMES = ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias",
         "output_layer.weight", "output_layer.bias"]


def _sd(g, pre):
    return {n: torch.as_tensor(g[pre + n.replace(".", "_")]) for n in NAMES}


def test_training_steps_match_reference_golden(golden_dir):
    from seqdex_amd.tvalue_trainer import TValue_Trainer
    g = np.load(os.path.join(golden_dir, "TV1_train.npz"))
    tr = TValue_Trainer((g["succ"], g["fail"]), seed=3)
    try:
        tr.init_TValue_function(state_dict=_sd(g, "w0_"))
        for it in range(4):
            tr.t["BATCH"].copy_(torch.as_tensor(g["x%d" % it]).cuda())
            tr.step()
            torch.cuda.synchronize()
            np.testing.assert_allclose(float(tr.t["LOSS"][0]), g["losses"][it], rtol=2e-5)
            if it == 0:
                np.testing.assert_allclose(tr.t["OUTPUT"].cpu().numpy(), g["pred0"], rtol=2e-5, atol=2e-6)
                flat_g = torch.cat([torch.as_tensor(g["g0_" + n.replace(".", "_")]).reshape(-1) for n in NAMES]).numpy()
                np.testing.assert_allclose(tr.t["GRADS"].cpu().numpy(), flat_g, rtol=2e-4, atol=2e-8)
                sd = tr.state_dict()
                for n in NAMES:
                    # Adam's first step is lr * g / (|g| + eps): elements whose gradient is ~eps are sensitive to its last bits
                    np.testing.assert_allclose(sd[n].numpy(), g["w1_" + n.replace(".", "_")], rtol=1e-5, atol=2e-5)
        sd = tr.state_dict()
        for n in NAMES:        # four Adam steps: the first ones move every weight by ~lr regardless of the gradient's size, so compare tightly
            np.testing.assert_allclose(sd[n].numpy(), g["w4_" + n.replace(".", "_")], rtol=1e-4, atol=2e-5)
    finally:
        tr.close()


def test_sampler_statistics_and_oracle_agreement(golden_dir):
    """the device sampler draws B/2 success and B/2 failure rows, adds noise in [-0.05, 0.05) per component and renormalises; a
    batch it produced, fed to the torch oracle, gives the same loss / parameters as the device step"""
    from seqdex_amd.tvalue_trainer import TValue_Trainer
    g = np.load(os.path.join(golden_dir, "TV1_train.npz"))
    tr = TValue_Trainer((g["succ"], g["fail"]), seed=9)
    try:
        tr.init_TValue_function(state_dict=_sd(g, "w0_"))
        tr.sample()
        torch.cuda.synchronize()
        x = tr.t["BATCH"].cpu().numpy().copy()
        np.testing.assert_allclose(np.linalg.norm(x, axis=1), 1.0, atol=1e-6)
        succ, fail = g["succ"][:-100], g["fail"]                              # the trainer holds the last 100 successes out
        for rows, data in ((x[:512], succ), (x[512:], fail)):
            d = np.abs(rows[:, None, :] - data[None, :, :]).max(-1).min(-1)   # distance to the nearest dataset row
            assert d.max() < 0.09 and d.mean() > 0.005                        # noisy (+-0.05, renormalised) copies of dataset rows
        tr.sample()
        torch.cuda.synchronize()
        assert np.abs(tr.t["BATCH"].cpu().numpy() - x).max() > 0.1            # a new draw every call
        xb = tr.t["BATCH"].cpu().numpy().copy()
        tr.step()
        torch.cuda.synchronize()
        sd_o, losses, _ = TO.train_steps(_sd(g, "w0_"), [xb])
        np.testing.assert_allclose(float(tr.t["LOSS"][0]), losses[0], rtol=2e-5)
        sd = tr.state_dict()
        for n in NAMES:
            np.testing.assert_allclose(sd[n].numpy(), sd_o[n].numpy(), rtol=1e-5, atol=2e-5)
    finally:
        tr.close()


def test_train_rollout_separates_the_classes(golden_dir):
    from seqdex_amd.tvalue_trainer import TValue_Trainer
    g = np.load(os.path.join(golden_dir, "TV1_train.npz"))
    tr = TValue_Trainer((g["succ"], g["fail"]), seed=1)
    try:
        tr.init_TValue_function(rollout=600)
        loss = tr.train_rollout(validate_every=300)
        assert len(tr.losses) == 2 and loss < 0.45, tr.losses              # starts at ln 2 = 0.693
        # the fixture's successes are uniform quaternions, its failures have w < -0.2: about a third of the successes are
        # indistinguishable from failures, the rest must be recognised - and so must the failures
        assert 0.55 < tr.valid_t_value_success_rate <= 1.0
        p = torch.sigmoid(tr.predict(torch.as_tensor(g["fail"][:256])))
        assert float((p[:, 0] > p[:, 1]).float().mean()) > 0.9
    finally:
        tr.close()


def test_insert_task_logs_tvalue_datasets_and_trainer_consumes_them(scene):
    """BlockAssemblyInsertSim's reset kernel logs the camera-frame target quaternion of every finished episode into the success or
    failure ring (IS:1392-1410); TValue_Trainer.from_task trains on them and the fitted weights go back into the task."""
    import yaml
    from seqdex_amd.tasks.block_assembly_insert_sim import BlockAssemblyInsertSim
    from seqdex_amd.tvalue_trainer import TValue_Trainer, flat_from_state_dict
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root_dir, "seqdex_amd/cfg/allegro_hand_block_assembly_insert_sim.yaml")))
    n = 64
    cfg["env"]["numEnvs"] = n
    task = BlockAssemblyInsertSim(cfg, device_type="cuda", device_id=0, headless=True, seed=3, piles_per_type=2, synthetic_states_per_type=4)
    g = torch.Generator().manual_seed(0)
    resets = 0
    for t in range(130):
        task.step(((torch.rand(n, 23, generator=g) * 2 - 1) * 0.2).cuda())
        resets += int(task.reset_buf.sum())
    task.step(torch.zeros(n, 23).cuda())                       # the resets flagged by the last step happen (and log) in this one
    torch.cuda.synchronize()
    cnt = task.sim.TV_COUNT.cpu().numpy()
    assert cnt.sum() == resets and cnt[1] > 100, (cnt, resets)  # random actions insert nothing: failures
    q = task.sim.TV_FAILURE[:int(cnt[1])].cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-4)        # unit quaternions (camera-frame target rotation)
    # give the success ring some rows so that the trainer has both classes (none occur with random actions)
    succ = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(160, 1) + 0.02 * torch.randn(160, 4, generator=g)
    task.sim.TV_SUCCESS[:160] = (succ / succ.norm(dim=1, keepdim=True)).cuda()
    task.sim.TV_COUNT[0] = 160
    tr = TValue_Trainer.from_task(task, seed=2)
    try:
        assert tr.num_success_data == 60 and tr.num_failure_data == int(min(cnt[1], 1048576))
        tr.init_TValue_function(rollout=200)
        tr.train_rollout(validate_every=200)
        assert np.isfinite(tr.losses[-1]) and tr.losses[-1] < 0.69
        task.sim.set_tvalue_weights(flat_from_state_dict(tr.state_dict()).numpy())
        task.step(torch.zeros(n, 23).cuda())
        torch.cuda.synchronize()
        assert np.isfinite(task.tvalue.cpu().numpy()).all()
    finally:
        tr.close()


def test_bi_optimization_outer_loop_one_round(tmp_path, monkeypatch):
    """one round of the chain's outer loop at toy size: Orient -> GraspSim -> InsertSim forward, then InsertSim again for the
    transition-value data, the trainer, and GraspSim fine-tuned with the new value; checkpoints in rl_games' layout are written and
    re-loaded along the way."""
    from seqdex_amd.scripts import bi_optimization as bo
    monkeypatch.chdir(tmp_path)
    paths, tv = bo.block_assembly(rounds=1, num_envs=64, epochs=2, tvalue_rollout=20, insert_minibatch=256)
    for k in ("search", "orient", "grasp", "insert"):
        ck = torch.load(paths[k], map_location="cpu", weights_only=False)
        assert "a2c_network.mu.weight" in ck["model"] and ck["epoch"] == 2
    from seqdex_amd.scripts import evaluation as ev
    res = ev.block_assembly(paths["orient"], paths["grasp"], paths["insert"], num_envs=64, games=64, insert_minibatch=256,
                            search_path=paths["search"])
    assert set(res) == {"BlockAssemblySearch", "BlockAssemblyOrient", "BlockAssemblyGraspSim", "BlockAssemblyInsertSim"}
    assert all(np.isfinite(v["reward"]) and v["length"] > 0 for v in res.values()), res
    assert 0.0 <= res["BlockAssemblyInsertSim"]["insert_success_rate"] <= 1.0
    assert tv is None or set(tv) == {"linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "linear3.weight", "linear3.bias",
                                      "output_layer.weight", "output_layer.bias"}
