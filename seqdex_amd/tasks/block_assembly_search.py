"""Host-side mirror of the reference task class `BlockAssemblySearch`
(tasks/block_assembly/allegro_hand_block_assembly_search.py:54 = SE; first policy of the chain).
The hand digs through the brick pile until the target brick becomes visible to a fixed camera.  Reproduced:
  * targets: fingers = moving average (0.6) of the scaled action, arm by the IK that keeps the hand base 0.24 above / 0.18 behind the
    target brick with a fixed wrist orientation (SE:1557-1596);
  * observation: the same 62 numbers as Orient in a 186-wide buffer (SE:1220-1230); its own asymmetric state (SE:1168-1218: pixel
    statistics of the segmentation image at columns 120..122, twists re-ordered, the hand-position history columns are zero as in the
    reference, which only ever averages a zeroed buffer);
  * reward min(-0.2 d, -0.06) - arm contacts - 0.005 |a|^2 + lift term, time-out reset, episodeLength 75 (SE:1660-1711);
  * finger drives kp 20 / damping 1 / effort 0.7 (SE:495-497);
  * segmentation camera (SE:755-757,873-878): 128 x 128, at (0.35, 0.19, 1.0) looking at (0.2, 0.19, 0), rendered after the 60 settling
    steps of a reset and - with the hand parked - at the last step of an episode; ray-cast against the scene's BOXES (csrc/sdx_camera.hip);
  * reset (SE:1274-1538): success = more target pixels than the brick type's threshold; successes hand their whole pile on
    (`pile_terminal_states()` -> `BlockAssemblyOrient(initial_piles=...)`); bricks back on the spawn lattice with +-0.02 noise, target
    dropped from 0.9 m, 60 settling steps, render, hand to the prepare pose.
Also built: RetriGraspTValue(650, 2) on the ten-frame buffer of 65-number frames (SE:395-410,1133-1166; computed and exposed as in the
reference, where it does not enter the reward either).  Not reproduced: teleoperation
perturbations, cv2 debug windows, the hand states saved next to the piles (SE:1325).  The pixel counts come from box geometry, not
from the studded meshes: thresholds tuned on Isaac Gym's renderer are only approximately meaningful (parity unpinned).
"""
from .block_assembly_orient import BlockAssemblyOrient


class BlockAssemblySearch(BlockAssemblyOrient):
    TASK_KIND = 3
    ONE_FRAME_NUM_OBS = 62                                                     # SE:155

    def _scene_overrides(self, scene):
        o = super()._scene_overrides(scene)                                    # finger gains 20 / 0.7 as Orient (SE:495-497)
        kd = [float(scene.raw["robot"]["dof"][j]["kd"]) for j in range(23)]
        for j in range(7, 23):
            kd[j] = 1.0
        o.update(kd=kd, seg_mass_scale=1.0, target_euler=[0.0, 3.14, 1.57])    # SE:854 (x1), SE:1569
        return o

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 agent_index=None, is_multi_agent=False, seed=22, initial_piles=None, piles_per_type=8):
        import torch
        from ..sim import SdxError
        if initial_piles is not None:
            raise SdxError("BlockAssemblySearch starts every episode from the spawn lattice (SE:1391-1396); it takes no saved piles")
        # the base class would settle synthetic piles for GraspSim; this task's "saved pile" is the lattice itself (the library's
        # default pile), so an explicit one-slot lattice is handed to it
        from ..scene import load_scene
        sc = load_scene()
        lattice = torch.zeros(8, 1, 132, 13)
        lattice[:, :, :, 6] = 1.0
        for i, (fs, pos) in enumerate(zip(sc.raw["free_spawn"], sc.spawn_positions())):   # (lifted above the floor slab, scene.to_desc)
            lattice[:, 0, i, 0:3] = torch.tensor(pos)
            lattice[:, 0, i, 3:7] = torch.tensor(fs["quat"])
        for i, fb in enumerate(sc.raw["fixed_bricks"]):
            lattice[:, 0, 72 + i, 0:3] = torch.tensor(fb["pos"])
        super().__init__(cfg, sim_params, physics_engine, device_type, device_id, headless, agent_index, is_multi_agent, seed,
                         lattice, piles_per_type)
        s = self.sim
        # RetriGraspTValue(650, 2) with torch-default init (SE:395-400); evaluated every step on the ten-frame buffer (SE:1133-1166)
        g = torch.Generator().manual_seed(seed + 1)
        flat = []
        for fin, fout in ((650, 1024), (1024, 512), (512, 128), (128, 2)):
            b = 1.0 / (fin ** 0.5)
            flat.append(((torch.rand(fout, fin, generator=g) * 2 - 1) * b).reshape(-1))
            flat.append((torch.rand(fout, generator=g) * 2 - 1) * b)
        s.set_retri_tvalue_weights(torch.cat(flat).numpy())
        self.t_value_obs_buf = s.TVALUE_OBS                                       # [N, 652]: 650 + 2 padding columns
        self.segmentation_pixels, self.emergence_reward = s.SEG_PIXELS, s.EMERGENCE
        self.extras["emergence_reward"] = s.EMERGENCE                          # SE:965

    def render_segmentation(self):
        self.sim.render_segmentation()
        return self.sim.SEG_IMAGE
