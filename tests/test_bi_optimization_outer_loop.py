"""host logic of the bi-optimisation outer loop (seqdex_amd/scripts/bi_optimization.py::block_assembly, after the reference's
scripts/bi_optimization.py:110-124) on CPU: the training run and the transition-value fit are replaced by stand-ins that record what
they were given, so the ORDER of the seven runs of a round, what each run is handed (checkpoints, piles, grasp states, transition
value, gates, epochs, env counts) and what the report says are checked without a GPU."""
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")


class _Sim:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def close(self):
        self.closed = True


def _patch(monkeypatch, grasp_counts=(3, 1, 2, 5, 4, 1, 2, 6), fits=(False, True, False)):
    """fits: which of the three transition_value_trainer calls of a round return a new state_dict (insert, grasp, orient leg)"""
    from seqdex_amd.scripts import bi_optimization as bo
    runs, fit_calls = [], []
    dug = torch.ones(8, 12, 132, 13)
    piles = torch.full((8, 20, 132, 13), 2.0)
    states = ([torch.zeros(int(c), 1, 13) for c in grasp_counts], [torch.zeros(int(c), 23, 2) for c in grasp_counts])

    def fake_main(task, num_envs, use_t_value=False, policy_path="", max_iterations=0, task_kwargs=None, tvalue_state=None, keep=False,
                  minibatch_size=0, mixed_precision=False, report=None, leg=""):
        runs.append(dict(task=task, num_envs=num_envs, use_t_value=use_t_value, policy_path=policy_path, epochs=max_iterations,
                         task_kwargs=dict(task_kwargs or {}), tvalue=tvalue_state, leg=leg, mixed_precision=mixed_precision,
                         minibatch_size=minibatch_size))
        if report is not None:
            report.append({"leg": leg, "task": task, "num_envs": num_envs, "epochs": max_iterations})
        sim = _Sim(TV_COUNT=torch.tensor([150, 9000]), HARVEST_COUNT=torch.tensor(np.array(grasp_counts, dtype=np.int32)))
        obj = types.SimpleNamespace(sim=sim, pile_terminal_states=lambda: dug if task == "BlockAssemblySearch" else piles,
                                    grasp_terminal_states=lambda: states, grasp_states_source="given")
        return "logs/%s/nn/%s.pth" % (task, task), (obj if keep else None)

    def fake_fit(task_obj, rollout, state_dict=None, seed=0):
        k = len(fit_calls) % 3
        fit_calls.append((rollout, state_dict, seed))
        return {"w": torch.full((2,), float(len(fit_calls)))} if fits[k] else state_dict

    monkeypatch.setattr(bo, "main_rlgames", fake_main)
    monkeypatch.setattr(bo, "transition_value_trainer", fake_fit)
    return bo, runs, fit_calls, dug, piles, states


def test_one_round_runs_the_reference_order_with_the_reference_hand_offs(monkeypatch):
    bo, runs, fits, dug, piles, states = _patch(monkeypatch)
    report = []
    paths, tv = bo.block_assembly(rounds=1, num_envs=4096, tvalue_rollout=300, mixed_precision=True, report=report,
                                  stage_epochs={"search": 20, "orient": 10, "grasp": 20, "insert": 48, "insert_backward": 32},
                                  gates={"orient": 0.0, "grasp": 0.0}, gates_after_fit={"orient": 0.5, "grasp": 0.28})
    # bi_optimization.py:115-124: forward Search, Orient, GraspSim, InsertSim; backward InsertSim, GraspSim, Orient
    assert [(r["leg"], r["task"]) for r in runs] == [
        ("forward", "BlockAssemblySearch"), ("forward", "BlockAssemblyOrient"), ("forward", "BlockAssemblyGraspSim"), ("forward", "BlockAssemblyInsertSim"),
        ("backward", "BlockAssemblyInsertSim"), ("backward", "BlockAssemblyGraspSim"), ("backward", "BlockAssemblyOrient")]
    assert [r["num_envs"] for r in runs] == [128, 4096, 4096, 4096, 4096, 4096, 128]           # :111 and :124 run at 128 envs
    assert [r["epochs"] for r in runs] == [20, 10, 20, 48, 32, 20, 10]                            # "<task>_backward" falls back to the task's entry
    assert all(r["mixed_precision"] for r in runs)
    assert [r["use_t_value"] for r in runs] == [False] * 4 + [True] * 3
    # hand-offs: the very tensors the previous stage returned
    assert runs[1]["task_kwargs"]["initial_piles"] is dug and runs[2]["task_kwargs"]["initial_piles"] is piles
    assert runs[3]["task_kwargs"]["grasp_states"] is states and runs[4]["task_kwargs"]["grasp_states"] is states
    assert runs[5]["task_kwargs"]["initial_piles"] is piles and runs[6]["task_kwargs"]["initial_piles"] is dug
    # checkpoints: the backward legs start from the forward checkpoints of their task
    assert [r["policy_path"] for r in runs[:4]] == ["", "", "", ""]
    assert runs[4]["policy_path"] == paths["insert"] and runs[5]["policy_path"].endswith("BlockAssemblyGraspSim.pth") and runs[6]["policy_path"].endswith("BlockAssemblyOrient.pth")
    # gates: open while no transition value exists; the insert leg's fit was skipped here, so the backward grasp leg still has none;
    # the grasp leg's fit succeeded, so the backward Orient leg carries it and uses the after-fit gate
    assert runs[1]["task_kwargs"]["tvalue_gate"] == 0.0 and runs[2]["task_kwargs"]["harvest_tvalue_gate"] == 0.0
    assert runs[5]["task_kwargs"]["harvest_tvalue_gate"] == 0.0 and runs[5]["tvalue"] is None
    assert runs[6]["task_kwargs"]["tvalue_gate"] == 0.5 and runs[6]["tvalue"] is not None and float(runs[6]["tvalue"]["w"][0]) == 2.0
    assert [f[2] for f in fits] == [0, 100, 200] and all(f[0] == 300 for f in fits)               # one fit after every backward leg (:121,:122,:124)
    assert tv is runs[6]["tvalue"]                                                                # the Orient leg's fit was skipped: the value is unchanged
    hand = [r for r in report if "handoff" in r]
    assert [h["empty"] for h in hand] == [False, False, False, True, False, True]
    assert hand[2]["harvested_per_type"] == [3, 1, 2, 5, 4, 1, 2, 6] and hand[2]["harvested_by"] == "the trained grasp policy"
    assert hand[3]["outcomes_success_failure"] == [150, 9000] and hand[4]["finite"]


def test_second_round_resumes_every_policy_and_keeps_the_transition_value(monkeypatch):
    bo, runs, fits, dug, piles, states = _patch(monkeypatch, fits=(True, True, True))
    paths, tv = bo.block_assembly(rounds=2, num_envs=256, epochs=3, tvalue_rollout=50, gates_after_fit={"orient": 0.5, "grasp": 0.28})
    assert len(runs) == 14 and len(fits) == 6
    second = runs[7:]
    assert [r["policy_path"] != "" for r in second] == [True] * 7                                  # round 2 resumes all four checkpoints
    assert all(r["epochs"] == 3 for r in runs)
    assert second[1]["tvalue"] is not None and second[1]["task_kwargs"]["tvalue_gate"] == 0.5      # the value of round 1 gates round 2's forward pass
    assert "tvalue_gate" not in runs[1]["task_kwargs"]                                             # no `gates` given: the task's own threshold (0.99, OR:1203)
    assert [f[2] for f in fits] == [0, 100, 200, 1, 101, 201]
    assert float(tv["w"][0]) == 6.0


def test_a_grasp_stage_without_states_hands_none_on(monkeypatch):
    bo, runs, fits, *_ = _patch(monkeypatch, grasp_counts=(0,) * 8)
    report = []
    bo.block_assembly(rounds=1, num_envs=64, epochs=2, report=report)
    assert runs[3]["task_kwargs"]["grasp_states"] is None and runs[3]["task_kwargs"]["synthetic_fallback"] is False
    h = [r for r in report if "handoff" in r][2]
    assert h["empty"] and h["source"] == "InsertSim synthesises its start states"


def test_grasp_minibatch_override_reaches_both_grasp_legs_and_nothing_else(monkeypatch):
    """round 5: the GraspSim legs may train on larger minibatches than the shipped 4 (which do not learn on this engine);
    every other stage keeps its own schedule, and the per-stage epoch table takes "grasp_backward" for the backward leg"""
    bo, runs, fits, *_ = _patch(monkeypatch)
    bo.block_assembly(rounds=1, num_envs=4096, grasp_minibatch=bo.CONFIG5_GRASP_MINIBATCH, stage_epochs=bo.CONFIG5_LEARNED_EPOCHS)
    assert [r["minibatch_size"] for r in runs] == [0, 0, 2048, 0, 0, 2048, 0]
    assert [r["epochs"] for r in runs] == [20, 10, 400, 48, 32, 100, 60]      # (backward Orient leg: 60 since round 6, CONFIG5_LEARNED_EPOCHS)
    runs.clear()
    bo.block_assembly(rounds=1, num_envs=4096, stage_epochs=bo.CONFIG5_EPOCHS)
    assert [r["minibatch_size"] for r in runs] == [0] * 7 and [r["epochs"] for r in runs] == [20, 10, 20, 48, 32, 20, 10]
