"""Mirror of the reference's rl_games-facing VecEnv adapter (tasks/hand_base/vec_task_rlgames.py): `VecTask` (VR:17-104)
and `RLgamesVecTaskPython` (VR:160-213).  The +-5 observation clamps and the +-1 action clamp are done inside the HIP
kernels (SDX_T_OBS_CLAMPED / SDX_T_STATES_CLAMPED, k_pre_physics), so step() launches nothing extra."""
import numpy as np
import torch


class Box:
    """stand-in for gym.spaces.Box (gym is not a dependency): low/high/shape only."""

    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float32), np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32


class VecTask:
    def __init__(self, task, rl_device, clip_observations=5.0, clip_actions=1.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_agents = 1
        self.num_observations = task.num_obs
        self.num_states = task.num_states
        self.num_actions = task.num_actions
        self.obs_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)          # VR:27
        self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)  # VR:28
        self.act_space = Box(np.ones(self.num_actions) * -1.0, np.ones(self.num_actions) * 1.0)        # VR:29
        if clip_observations != 5.0 or clip_actions != 1.0:
            raise ValueError("clip values are fixed at the reference defaults 5.0 / 1.0 (VR:18), fused in the kernels")
        self.clip_obs, self.clip_actions = clip_observations, clip_actions
        self.rl_device = task.device                                                                   # VR:33
        self.info = {"action_space": self.act_space, "observation_space": self.obs_space,
                     "state_space": self.state_space, "agents": 1}                                     # VR:37-41

    def has_action_masks(self):
        return False

    def seed(self, seed):
        pass

    def set_train_info(self, env_frames, *args, **kwargs):
        pass

    def get_env_state(self):
        return None

    def set_env_state(self, env_state):
        pass

    def step(self, actions):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    @property
    def get_number_of_agents(self):
        return self.num_agents

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_acts(self):
        return self.num_actions

    @property
    def num_obs(self):
        return self.num_observations

    def get_env_info(self):
        return self.info


class RLgamesVecTaskPython(VecTask):
    def get_state(self):
        return self.task.sim.STATES_CLAMPED

    def _obs_dict(self):
        return {"obs": self.task.sim.OBS_CLAMPED, "states": self.task.sim.STATES_CLAMPED}               # VR:171-172

    def step(self, actions):
        self.task.step(actions)                      # clamp to +-1 happens in k_pre_physics (VR:166)
        return self._obs_dict(), self.task.rew_buf, self.task.reset_buf, self.task.extras

    def reset(self):
        a = 0.01 * (1 - 2 * torch.rand([self.task.num_envs, self.task.num_actions], dtype=torch.float32,
                                       device=self.rl_device))                                          # VR:180
        self.task.step(a)
        return self._obs_dict()
