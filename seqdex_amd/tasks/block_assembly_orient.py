"""Host-side mirror of the reference task class `BlockAssemblyOrient`
(tasks/block_assembly/allegro_hand_block_assembly_orient.py:94 = OR; BASELINE.json configs[2]).

Same robot, bin and brick pile as BlockAssemblyGraspSim; what the Orient task changes and this build reproduces:
  * targets: fingers from the action (moving average 0.2), arm by IK that tracks the target brick (OR:1720-1778);
  * observation: 62 numbers per frame, 186-wide buffer whose columns 62.. are never written (compute_real_observations OR:1308-1326);
    asymmetric states as in GraspSim; the T-value is gated at 0.99 (OR:1203);
  * reward exp(-5 (1 - (z_align + 1)/2) - 5 max(d - 0.4, 0)), reset on time-out only (OR:1843-1907), episodeLength 75;
  * finger drives kp 20 / effort 0.7 (OR:596-597), target brick 50 x heavier (OR:977);
  * reset with the two scripted 50-step pre-grasp phases (OR:1427-1461, 1655-1695) - inside sdx_step, on the device.
Not reproduced: the terminal-state harvesting of this task (OR:1463-1515, 8 x 11 024 x 108 x 13 floats), the
36-brick floor of this scene (the GraspSim slab is used), the density 2000 of the fixed bricks (they are static here anyway).
"""
from .. import _abi
from .block_assembly_grasp_sim import BlockAssemblyGraspSim


class BlockAssemblyOrient(BlockAssemblyGraspSim):
    TASK_KIND = 1
    ONE_FRAME_NUM_OBS = 62                                                     # OR:191-192

    def _scene_overrides(self, scene):
        kp = [float(scene.raw["robot"]["dof"][j]["kp"]) for j in range(23)]
        effort = [float(scene.raw["robot"]["dof"][j]["effort"]) for j in range(23)]
        for j in range(7, 23):                                                 # OR:595-597
            kp[j], effort[j] = 20.0, 0.7
        return {"kp": kp, "effort": effort, "seg_mass_scale": 50.0, "target_euler": [0.0, 3.1415, 1.571]}
