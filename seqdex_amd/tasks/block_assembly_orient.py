"""Host-side mirror of the reference task class `BlockAssemblyOrient`
(tasks/block_assembly/allegro_hand_block_assembly_orient.py:94 = OR; BASELINE.json configs[2]).

Same robot, bin and brick pile as BlockAssemblyGraspSim; what the Orient task changes and this build reproduces:
  * targets: fingers from the action (moving average 0.2), arm by IK that tracks the target brick (OR:1720-1778);
  * observation: 62 numbers per frame, 186-wide buffer whose columns 62.. are never written (compute_real_observations OR:1308-1326);
    asymmetric states as in GraspSim; the T-value is gated at 0.99 (OR:1203);
  * reward exp(-5 (1 - (z_align + 1)/2) - 5 max(d - 0.4, 0)), reset on time-out only (OR:1843-1907), episodeLength 75;
  * finger drives kp 20 / effort 0.7 (OR:596-597), target brick 50 x heavier (OR:977);
  * reset with the two scripted 50-step pre-grasp phases (OR:1427-1461, 1655-1695) - inside sdx_step, on the device.
  * terminal-state harvesting (OR:1463-1488): finished episodes that leave the target brick reachable hand their whole brick pile on
    (`pile_terminal_states()` -> `BlockAssemblyGraspSim(initial_piles=...)`), ring of 512 per brick-type group (SDX_PILE_SLOTS=10000 in the environment: the reference's 10 000).
Not reproduced: the 36-brick floor of this scene (the GraspSim slab is used), the density 2000 of the fixed
bricks (they are static here anyway).
"""
import torch

from .. import _abi
from .block_assembly_grasp_sim import BlockAssemblyGraspSim


class BlockAssemblyOrient(BlockAssemblyGraspSim):
    TASK_KIND = 1
    ONE_FRAME_NUM_OBS = 62                                                     # OR:191-192

    def __init__(self, *args, tvalue_gate=0.99, **kw):
        """tvalue_gate: the threshold the task binarises its transition value at (OR:1203: 0.99)"""
        self.tvalue_gate = float(tvalue_gate)
        super().__init__(*args, **kw)

    def _scene_overrides(self, scene):
        kp = [float(scene.raw["robot"]["dof"][j]["kp"]) for j in range(23)]
        effort = [float(scene.raw["robot"]["dof"][j]["effort"]) for j in range(23)]
        for j in range(7, 23):                                                 # OR:595-597
            kp[j], effort[j] = 20.0, 0.7
        return {"kp": kp, "effort": effort, "seg_mass_scale": 50.0, "target_euler": [0.0, 3.1415, 1.571],
                "orient_tvalue_gate": getattr(self, "tvalue_gate", 0.99)}

    def pile_terminal_states(self):
        """[8, K, 132, 13] pile states harvested so far (K = the smallest fill over the 8 brick-type groups; None while one is empty):
        the reference's saved_searching_ternimal_states list (OR:1483-1510), the format BlockAssemblyGraspSim loads (GS:412-413)."""
        import numpy as np
        s = self.sim
        cnt = np.minimum(s.PILE_HARVEST_COUNT.cpu().numpy(), s.PILE_HARVEST.shape[1])
        k = int(cnt.min())
        if k == 0:
            return None
        # serial (step, env) order of the appends (SdxSim.ring_rows), the first k of every group: GraspSim's resets index these by position
        return torch.stack([s.ring_rows(s.PILE_HARVEST[t], s.PILE_HARVEST_KEYS[t], cnt[t])[:k] for t in range(8)])

    def save_pile_terminal_states(self, path):
        """the harvested piles as the reference's pickle (list[8] of [K_t, 132, 13]; OR:1505-1510, SE:1349-1350): what
        `BlockAssemblyGraspSim(initial_piles=path)` / the reference's GS:412-413 load"""
        import numpy as np
        from ..piles import save_pile_pickle
        s = self.sim
        cnt = np.minimum(s.PILE_HARVEST_COUNT.cpu().numpy(), s.PILE_HARVEST.shape[1])
        save_pile_pickle(path, [s.ring_rows(s.PILE_HARVEST[t], s.PILE_HARVEST_KEYS[t], cnt[t]) for t in range(8)], counts=cnt)
