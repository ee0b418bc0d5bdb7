"""host logic of the chained evaluation's harvest-gate ladders (seqdex_amd/scripts/evaluation.py::block_assembly_chain, stage 1 / 2) on CPU:
`main_rlgames` is replaced by a stand-in whose harvest counts depend on the gate it is given, so the control flow - replay at the next rung,
what is reported, what is handed on, when the chain refuses - is exercised without a GPU."""
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")


class _Sim:
    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.closed = False

    def close(self):
        self.closed = True


def _stub_stages(monkeypatch, orient_accepts_below, grasp_accepts_below, grasp_counts=(5, 0, 9, 7, 0, 0, 3, 6)):
    """Orient harvests 12 piles per group when its gate is < orient_accepts_below, else none; GraspSim harvests `grasp_counts` when its gate is
    < grasp_accepts_below, else nothing.  Every call is recorded."""
    from seqdex_amd.scripts import evaluation as ev
    calls = []

    def fake_main(task, num_envs, **kw):
        tk = kw.get("task_kwargs") or {}
        st = {"task": task, "num_envs": num_envs, "env_steps": 80 * num_envs, "steps_per_env": 80, "wall_s": 0.5, "env_steps_per_s": 160.0 * num_envs,
              "policy": "stub", "success_buf_mean": 0.0}
        if task == "BlockAssemblyOrient":
            gate = tk["tvalue_gate"]
            calls.append(("orient", gate))
            n = 12 if gate < orient_accepts_below else 0
            piles = torch.zeros(8, 16, 132, 13)
            piles[..., 6] = 1.0
            sim = _Sim(PILE_HARVEST=piles, PILE_HARVEST_COUNT=torch.full((8,), n, dtype=torch.int32), PILE_HARVEST_KEYS=torch.arange(8 * 16).view(8, 16))
            obj = types.SimpleNamespace(sim=sim, pile_terminal_states=lambda: (piles[:, :n].clone() if n else None))
            return obj, st
        if task == "BlockAssemblyGraspSim":
            gate = tk["harvest_tvalue_gate"]
            calls.append(("grasp", gate))
            cnt = np.array(grasp_counts if gate < grasp_accepts_below else (0,) * 8, dtype=np.int32)
            sim = _Sim(HARVEST_COUNT=torch.from_numpy(cnt))
            states = ([torch.zeros(int(c), 1, 13) for c in cnt], [torch.zeros(int(c), 23, 2) for c in cnt])
            return types.SimpleNamespace(sim=sim, grasp_terminal_states=lambda: states), st
        assert task == "BlockAssemblyInsertSim"
        calls.append(("insert", tk["grasp_states"] is not None))
        return types.SimpleNamespace(sim=_Sim(), grasp_states_source="given", synthetic_groups=[]), st

    monkeypatch.setattr(ev, "main_rlgames", fake_main)
    return ev, calls


def test_a_single_gate_behaves_as_before(monkeypatch):
    ev, calls = _stub_stages(monkeypatch, orient_accepts_below=0.6, grasp_accepts_below=0.3)
    res, hand = ev.block_assembly_chain(64, None, synthetic_fallback=True, orient_tvalue_gate=0.5, grasp_tvalue_gate=0.28)
    assert calls == [("orient", 0.5), ("grasp", 0.28), ("insert", True)]
    assert res["orient"]["tvalue_gate"] == 0.5 and "tvalue_gates_tried" not in res["orient"]
    assert res["grasp"]["tvalue_gate"] == 0.28 and res["grasp"]["grasp_states_harvested_per_type"] == [5, 0, 9, 7, 0, 0, 3, 6]
    assert tuple(hand["piles"].shape) == (8, 12, 132, 13)
    assert res["chain_env_steps"] == 3 * 80 * 64


def test_the_ladder_replays_a_stage_until_a_rung_harvests(monkeypatch):
    ev, calls = _stub_stages(monkeypatch, orient_accepts_below=0.29, grasp_accepts_below=0.1)
    res, hand = ev.block_assembly_chain(64, None, synthetic_fallback=True, orient_tvalue_gate=ev.CHAIN_ORIENT_GATES, grasp_tvalue_gate=ev.CHAIN_GRASP_GATES)
    assert calls == [("orient", 0.5), ("orient", 0.4), ("orient", 0.3), ("orient", 0.28), ("grasp", 0.28), ("grasp", 0.0), ("insert", True)]
    o = res["orient"]
    assert o["tvalue_gate"] == 0.28 and [t["tvalue_gate"] for t in o["tvalue_gates_tried"]] == [0.5, 0.4, 0.3, 0.28]
    assert [sum(t["piles_harvested_per_type"]) for t in o["tvalue_gates_tried"]] == [0, 0, 0, 96]
    assert abs(o["wall_s_of_the_rungs_not_handed_on"] - 1.5) < 1e-9            # three empty rungs of 0.5 s, not part of the chain's rate
    g = res["grasp"]
    assert g["tvalue_gate"] == 0.0 and len(g["tvalue_gates_tried"]) == 2 and abs(g["wall_s_of_the_rungs_not_handed_on"] - 0.5) < 1e-9
    assert res["chain_env_steps"] == 3 * 80 * 64                                # only the runs that were handed on are counted
    assert len(hand["grasp_obj"]) == 8 and hand["grasp_obj"][2].shape[0] == 9


def test_a_grasp_rung_with_fewer_than_three_groups_is_not_handed_on(monkeypatch):
    ev, calls = _stub_stages(monkeypatch, orient_accepts_below=1.0, grasp_accepts_below=1.0, grasp_counts=(4, 0, 0, 0, 2, 0, 0, 0))
    res, hand = ev.block_assembly_chain(64, None, synthetic_fallback=True, orient_tvalue_gate=0.5, grasp_tvalue_gate=(0.28, 0.0))
    assert calls[:3] == [("orient", 0.5), ("grasp", 0.28), ("grasp", 0.0)]      # two groups only: the next rung is played; the last rung is kept whatever it gives
    assert res["grasp"]["tvalue_gate"] == 0.0 and res["grasp"]["grasp_states_harvested_per_type"] == [4, 0, 0, 0, 2, 0, 0, 0]


def test_without_fallback_the_chain_refuses_as_the_reference_does(monkeypatch):
    ev, calls = _stub_stages(monkeypatch, orient_accepts_below=0.1, grasp_accepts_below=0.1)
    with pytest.raises(RuntimeError, match="harvested no pile state"):
        ev.block_assembly_chain(64, None, synthetic_fallback=False, orient_tvalue_gate=0.99, grasp_tvalue_gate=0.8)
    assert calls == [("orient", 0.99)]
    ev, calls = _stub_stages(monkeypatch, orient_accepts_below=1.0, grasp_accepts_below=0.1)
    with pytest.raises(RuntimeError, match="harvested no grasp terminal state"):
        ev.block_assembly_chain(64, None, synthetic_fallback=False, orient_tvalue_gate=0.99, grasp_tvalue_gate=0.8)
    assert calls == [("orient", 0.99), ("grasp", 0.8)]


def test_checkpoint_driven_evaluation_plays_the_stages_in_order_and_hands_their_states_on(monkeypatch):
    """block_assembly(): scripts/evaluation.py:111-119 - every stage restored from its checkpoint through the launcher's command line, the
    pile / grasp states of a stage are what the next one is constructed with; Search only when a checkpoint is given, at <= 128 envs"""
    from seqdex_amd.scripts import evaluation as ev
    calls = []
    dug, piles = torch.zeros(8, 9, 132, 13), torch.ones(8, 11, 132, 13)
    states = ([torch.zeros(2, 1, 13)] * 8, [torch.zeros(2, 23, 2)] * 8)

    def fake_play(task, num_envs, play=True, use_t_value=False, policy_path="", games=0, task_kwargs=None, minibatch_size=0):
        calls.append((task, num_envs, use_t_value, policy_path, games, dict(task_kwargs or {}), minibatch_size))
        sim = _Sim(HARVEST_COUNT=torch.full((8,), 2, dtype=torch.int32))
        obj = types.SimpleNamespace(sim=sim, extras={"success_buf": torch.tensor([1.0, 0.0])}, grasp_states_source="given",
                                    pile_terminal_states=lambda: dug if task == "BlockAssemblySearch" else piles, grasp_terminal_states=lambda: states)
        return 1.5, 75.0, obj

    monkeypatch.setattr(ev, "play_checkpoint", fake_play)
    out = ev.block_assembly("o.pth", "g.pth", "i.pth", num_envs=512, games=64, insert_minibatch=1024, search_path="s.pth")
    assert [c[0] for c in calls] == ["BlockAssemblySearch", "BlockAssemblyOrient", "BlockAssemblyGraspSim", "BlockAssemblyInsertSim"]
    assert [c[1] for c in calls] == [128, 512, 512, 512] and [c[3] for c in calls] == ["s.pth", "o.pth", "g.pth", "i.pth"]
    assert all(c[2] for c in calls) and all(c[4] == 64 for c in calls)                     # use_t_value on, --games for every stage
    assert calls[1][5]["initial_piles"] is dug and calls[2][5]["initial_piles"] is piles and calls[3][5]["grasp_states"] is states
    assert calls[3][6] == 1024 and calls[0][6] == 0
    assert out["BlockAssemblySearch"]["piles_handed_on"] == 9 and out["BlockAssemblyOrient"]["piles_handed_on"] == 11
    assert out["BlockAssemblyGraspSim"]["grasp_states_handed_on"] == 16 and out["BlockAssemblyInsertSim"]["insert_success_rate"] == 0.5
    calls.clear()
    ev.block_assembly("o.pth", "g.pth", "i.pth", num_envs=64)                              # no Search checkpoint: three stages, Orient settles its own piles
    assert [c[0] for c in calls] == ["BlockAssemblyOrient", "BlockAssemblyGraspSim", "BlockAssemblyInsertSim"] and calls[0][5]["initial_piles"] is None
