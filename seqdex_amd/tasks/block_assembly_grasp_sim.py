"""Host-side mirror of the reference task class `BlockAssemblyGraspSim`
(tasks/block_assembly/allegro_hand_block_assembly_grasp_sim.py:94, base class tasks/hand_base/base_task.py:22):
same constructor signature, buffers, and step()/pre_physics_step()/post_physics_step()/compute_observations()/
reset_idx() methods, but every method is one call into libseqdex_hip.so - there is no torch arithmetic here.
"""
import numpy as np
import torch

from .. import _abi
from ..piles import generate_piles
from ..sim import SdxSim


class BlockAssemblyGraspSim:
    TASK_KIND = 0                     # sdx_scene_desc.task_kind
    ONE_FRAME_NUM_OBS = _abi.OBS_FRAME
    STACK_OBS = 3                     # GS:189

    def _scene_overrides(self, scene):
        """task-specific entries of sdx_scene_desc (hook for the other BlockAssembly* tasks)"""
        return {"grasp_tvalue_gate": float(getattr(self, "harvest_tvalue_gate", 0.8))}     # GS:1406

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 agent_index=None, is_multi_agent=False, seed=22, initial_piles=None, piles_per_type=8, harvest_tvalue_gate=None):
        """harvest_tvalue_gate: the transition value above which a finished episode's grasp state is harvested (GS:1406: 0.8)"""
        if harvest_tvalue_gate is not None:
            self.harvest_tvalue_gate = float(harvest_tvalue_gate)
        self.cfg = cfg
        env = cfg["env"]
        self.num_envs = env["numEnvs"]
        self.max_episode_length = env["episodeLength"]                       # GS:147
        self.stack_obs = self.STACK_OBS
        self.one_frame_num_obs, self.one_frame_num_states = self.ONE_FRAME_NUM_OBS, _abi.STATE_FRAME   # GS:207-208
        cfg["env"]["numObservations"] = self.ONE_FRAME_NUM_OBS * self.stack_obs   # GS:209-211
        cfg["env"]["numStates"] = _abi.NUM_STATES
        cfg["env"]["numActions"] = _abi.NUM_ACTIONS
        self.num_obs, self.num_states, self.num_actions = self.ONE_FRAME_NUM_OBS * self.stack_obs, _abi.NUM_STATES, _abi.NUM_ACTIONS
        self.control_freq_inv = env.get("controlFrequencyInv", 1)
        if device_type not in ("cuda", "GPU"):
            raise RuntimeError("seqdex_amd runs the task on the GPU only (no CPU pipeline; see DESIGN.md §5)")
        self.device = "cuda:%d" % device_id                                   # BT:31-33
        self.headless = headless
        overrides = {}
        simc = cfg.get("sim", {})
        if "substeps" in simc:
            overrides["substeps"] = int(simc["substeps"])
        px = simc.get("physx", {})
        if "num_position_iterations" in px:
            overrides["solver_iters"] = int(px["num_position_iterations"])
        if "contact_offset" in px:
            overrides["contact_offset"] = float(px["contact_offset"])
        overrides["max_episode_length"] = float(self.max_episode_length)
        overrides["act_moving_average"] = float(env.get("actionsMovingAverage", 1.0))
        from ..scene import load_scene
        scene = load_scene()
        overrides["task_kind"] = self.TASK_KIND
        overrides.update(self._scene_overrides(scene))
        self.sim = SdxSim(self.num_envs, device=self.device, seed=seed, scene=scene, **overrides)
        s = self.sim
        # buffers of BaseTask (BT:57-69) are zero-copy views of library-owned memory
        self.obs_buf, self.states_buf = s.OBS, s.STATES
        self.rew_buf, self.reset_buf = s.REW, s.RESET
        self.progress_buf, self.randomize_buf = s.PROGRESS, s.RANDOMIZE
        self.root_state_tensor, self.dof_state = s.ROOT, s.DOF                # GS:313-322
        self.rigid_body_states, self.contact_tensor = s.RB, s.CONTACT
        self.jacobian_eef, self.cur_targets, self.prev_targets = s.JAC_EEF, s.TARGETS, s.PREV_TARGETS
        self.actions = s.ACTIONS
        self.segmentation_target_init_pos, self.segmentation_target_init_rot = s.INIT_POS, s.INIT_ROT
        self.successes, self.consecutive_successes = s.SUCCESSES, s.CONS_SUCCESSES
        self.meta_rew_buf, self.arm_hand_finger_dist, self.tvalue = s.META_REW, s.FINGER_DIST, s.TVALUE
        self.contacts = s.ARM_CONTACTS
        self.extras = {"student_obs_buf": s.STUDENT_OBS, "success_buf": s.SUCCESS_BUF,                 # GS:458-459
                       "emergence_reward": torch.zeros(self.num_envs, device=self.device),             # GS:1071-1073
                       "heap_movement_penalty": torch.zeros(self.num_envs, device=self.device),
                       "meta_reward": s.META_REW}
        self.arm_hand_dof_lower_limits = torch.as_tensor(s.scene.lower, device=self.device)
        self.arm_hand_dof_upper_limits = torch.as_tensor(s.scene.upper, device=self.device)
        self.total_steps = 0
        # saved pile states: the reference unpickles intermediate_state/...tvalue.pkl (GS:412-413), produced by the
        # Search->Orient stages; that artefact is not shipped, so piles are settled with the engine itself.
        if isinstance(initial_piles, str):                                    # the reference's pickle: list[8] of [slots, 132, 13] (GS:412-413)
            from ..piles import load_pile_pickle
            initial_piles = load_pile_pickle(initial_piles)
        if initial_piles is None:
            initial_piles = generate_piles(piles_per_type, device=self.device, seed=seed)
        s.load_initial_states(initial_piles)
        # GraspInsertTValue(4, 2) with torch-default init (GS:417-419); weights can be replaced via set_tvalue_weights
        g = torch.Generator().manual_seed(seed)
        flat = []
        for fin, fout in ((4, 256), (256, 128), (128, 64), (64, 2)):
            b = 1.0 / np.sqrt(fin)
            flat.append(((torch.rand(fout, fin, generator=g) * 2 - 1) * b).reshape(-1))
            flat.append((torch.rand(fout, generator=g) * 2 - 1) * b)
        s.set_tvalue_weights(torch.cat(flat).numpy())

    # ------------------------------------------------------------------ BaseTask.step, BT:130-150
    def step(self, actions):
        self.sim.step(actions)
        self.total_steps += 1

    def pre_physics_step(self, actions):          # GS:1555-1638
        self.sim.pre_physics(actions)

    def post_physics_step(self):                  # GS:1640-1645
        self.sim.post_physics()

    def compute_observations(self):               # GS:1090-1218
        self.sim.compute_observations()

    def reset_idx(self, env_ids, goal_env_ids=None):   # GS:1361-1553
        mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        mask[env_ids] = 1
        self.sim.reset_idx(mask)

    # ------------------------------------------------------------------ hand-off to BlockAssemblyInsertSim
    def grasp_terminal_states(self):
        """the harvested terminal states in the reference's hand-off layout (GS:391-417,1440-1450): two lists of 8 tensors,
        object root states [K_t, 1, 13] and hand joint states [K_t, 23, 2] per brick-type group (K_t = filled ring slots)."""
        s = self.sim
        cnt = s.HARVEST_COUNT.cpu().numpy()
        # serial (step, env) order of the appends (SdxSim.ring_rows): InsertSim's resets index these lists by position
        obj = [s.ring_rows(s.HARVEST_OBJ[t], s.HARVEST_KEYS[t], cnt[t]).unsqueeze(1) for t in range(8)]
        hand = [s.ring_rows(s.HARVEST_HAND[t], s.HARVEST_KEYS[t], cnt[t]) for t in range(8)]
        return obj, hand

    def save_grasp_terminal_states(self, path):
        """np.savez of grasp_terminal_states(): obj_<t> / hand_<t> arrays (our stand-in for the two pickles of GS:1447-1450)."""
        obj, hand = self.grasp_terminal_states()
        np.savez(path, **{"obj_%d" % t: obj[t].cpu().numpy() for t in range(8)},
                 **{"hand_%d" % t: hand[t].cpu().numpy() for t in range(8)})

    def get_states(self):                         # BT:152-153
        return self.states_buf

    def render(self, sync_frame_time=False):      # headless: BT:155-177 are viewer calls
        return None
