"""-m gpu, VERDICT r4 item 2(a): a hand of this engine can lift a brick out of the pile.  1 024 BlockAssemblyGraspSim envs under the scripted
reach - descend - pinch - hold controller (seqdex_amd/scripts/evaluation.py::scripted_grasp_controller, csrc/sdx_task.hip::k_scripted_grasp):
in at least 30 % of the envs the target brick is held >= 5 cm above where it lay with finger_dist < 0.5 (GS:1164-1165, 1725) - 54 % measured
(profiles/r5_scripted_lift_scan.txt; 0 % through round 4, when the pinch closed over the studs and the hand kept following the brick it held)."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_scripted_controller_lifts_the_target_brick():
    from seqdex_amd.scripts.evaluation import scripted_lift_statistics
    st = scripted_lift_statistics(1024)
    assert st["held_5cm_frac"] >= 0.30, st
    assert st["held_max_m"] > 0.15, st                                         # carried up by the task's own lift phase (GS:1600-1609)
    assert st["contact_stats"][1] == 0 and st["contact_stats"][3] == 0, st     # no contact lost, no pair list overflowed
    assert sum(1 for c in st["per_type_held_5cm"] if c >= 10) >= 5, st         # not one lucky brick type
