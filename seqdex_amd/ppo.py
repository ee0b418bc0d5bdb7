"""Python binding of the sdxp_* C ABI (include/seqdex.h): the PPO inner loops of rl_games' A2CAgent on the GPU.
torch is plumbing (pointers, streams); arithmetic is in libseqdex_hip.so (seqdex_amd/csrc/sdxp_kernels.hip)."""
import ctypes as C

import torch

from . import _abi
from .sim import SdxError, _stream_ptr, wrap_device_pointer


class Ctrl(C.Structure):
    """host mirror of SdxpCtrl (csrc/sdxp_types.h)"""
    _fields_ = [("step", C.c_int32), ("mb_index", C.c_int32), ("mini_epoch", C.c_int32), ("ac_pending", C.c_int32),
                ("cv_pending", C.c_int32), ("ac_t", C.c_int32), ("cv_t", C.c_int32), ("n_mb", C.c_int32),
                ("ac_lr", C.c_float), ("cv_lr", C.c_float), ("ac_lr_applied", C.c_float), ("cv_lr_applied", C.c_float),
                ("ac_gscale", C.c_float), ("cv_gscale", C.c_float), ("ac_gnorm", C.c_float), ("cv_gnorm", C.c_float),
                ("ac_bc1", C.c_float), ("ac_bc2", C.c_float), ("cv_bc1", C.c_float), ("cv_bc2", C.c_float),
                ("last_kl", C.c_float), ("sum_a_loss", C.c_float), ("sum_c_loss", C.c_float), ("sum_b_loss", C.c_float),
                ("sum_kl", C.c_float), ("sum_cv_loss", C.c_float), ("sum_entropy", C.c_float), ("acc", C.c_float * 8),
                ("games_sum_rew", C.c_float), ("games_sum_len", C.c_float), ("games_cnt", C.c_float), ("pad0", C.c_float),
                ("gn2_ac", C.c_float), ("gn2_cv", C.c_float), ("world", C.c_int32), ("ll_tag", C.c_uint32),
                ("prev_mb", C.c_int32), ("prev_mini_epoch", C.c_int32), ("n2_part", C.c_float * 4),
                ("gx", C.c_float * (3 * 3 * 64)), ("rms_count", C.c_double),
                ("ac_b1pow", C.c_double), ("ac_b2pow", C.c_double), ("cv_b1pow", C.c_double), ("cv_b2pow", C.c_double)]


def make_config(num_actors, params=None, world_size=1, obs_dim=None, state_dim=None):
    """sdxp_config from the `params.config` block of cfg/lego/ppo_continuous_grasp.yaml (YG) + network units."""
    p = params or {}
    cfgd = p.get("config", {})
    cv = cfgd.get("central_value_config", {})
    units = p.get("network", {}).get("mlp", {}).get("units", [1024, 512, 256])
    c = _abi.PPOConfig()
    c.num_actors = num_actors
    c.horizon = cfgd.get("horizon_length", 8)
    c.minibatch = cfgd.get("minibatch_size", 4)
    c.mini_epochs = cfgd.get("mini_epochs", 5)
    c.cv_minibatch = cv.get("minibatch_size", c.minibatch)
    c.cv_mini_epochs = cv.get("mini_epochs", c.mini_epochs)
    raw_obs = obs_dim or _abi.NUM_OBS
    c.obs_dim = (raw_obs + 3) // 4 * 4          # the kernels read rows with float4 accesses: network input zero-padded to x4
    c.obs_cols = raw_obs if raw_obs != c.obs_dim else 0
    c.state_dim, c.act_dim = state_dim or _abi.NUM_STATES, _abi.NUM_ACTIONS
    c.units[:] = units
    c.gamma, c.tau = cfgd.get("gamma", 0.99), cfgd.get("tau", 0.95)
    c.lr, c.cv_lr = float(cfgd.get("learning_rate", 3e-4)), float(cv.get("learning_rate", 1e-3))
    c.e_clip, c.grad_norm = cfgd.get("e_clip", 0.1), cfgd.get("grad_norm", 1.0)
    c.critic_coef, c.entropy_coef = cfgd.get("critic_coef", 1.0), cfgd.get("entropy_coef", 0.0)
    c.bounds_loss_coef = cfgd.get("bounds_loss_coef", 1e-3)
    c.kl_threshold = cfgd.get("kl_threshold", 0.02)
    c.clip_value = int(bool(cfgd.get("clip_value", True)))
    c.truncate_grads = int(bool(cfgd.get("truncate_grads", True)))
    c.normalize_advantage = int(bool(cfgd.get("normalize_advantage", True)))
    c.cv_normalize_input = int(bool(cv.get("normalize_input", True)))
    c.adaptive_lr = int(cfgd.get("lr_schedule", "adaptive") == "adaptive")
    c.world_size = world_size
    c.mixed_precision = int(bool(cfgd.get("mixed_precision", False)))   # bf16 trunk GEMMs on the large-minibatch path
    return c


class SdxPPO:
    def __init__(self, num_actors, params=None, device="cuda:0", seed=22, config=None, world_size=1, obs_dim=None, state_dim=None):
        if not torch.cuda.is_available():
            raise SdxError("seqdex_amd needs a ROCm GPU (gfx950); there is no CPU fallback for the product path")
        self.lib = _abi.load_library()
        self.device = torch.device(device)
        self.cfg = config or make_config(num_actors, params, world_size, obs_dim, state_dim)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.sdxp_create(C.byref(self.cfg), idx, C.c_uint64(seed), C.byref(h))
        if rc != 0:
            raise SdxError("sdxp_create failed (%d): %s" % (rc, self.lib.sdxp_last_error(None).decode()))
        self.h = h
        self.t = {}
        for name, tid in _abi.TP.items():
            ptr, shape, ndim, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
            self._check(self.lib.sdxp_tensor(self.h, tid, C.byref(ptr), shape, C.byref(ndim), C.byref(dt)))
            self.t[name] = wrap_device_pointer(ptr.value, [shape[i] for i in range(ndim.value)], dt.value, self.device)
        self.num_actors, self.horizon = self.cfg.num_actors, self.cfg.horizon
        self.actions = torch.zeros(num_actors, self.cfg.act_dim, device=self.device)

    def _check(self, rc):
        if rc != 0:
            raise SdxError("libseqdex_hip (ppo) error %d: %s" % (rc, self.lib.sdxp_last_error(self.h).decode()))

    @staticmethod
    def _p(t, dtype=torch.float32):
        if t is None:
            return C.c_void_p(0)
        assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.dtype, dtype)
        return C.c_void_p(t.data_ptr())

    def act(self, t, obs, states, dones=None, eps=None):
        self._check(self.lib.sdxp_act(self.h, t, self._p(obs), self._p(states), self._p(dones, torch.int64), self._p(eps),
                                      self._p(self.actions), _stream_ptr(self.device)))
        return self.actions

    def store_rewards(self, t, rew, dones_after=None):
        self._check(self.lib.sdxp_store_rewards(self.h, t, self._p(rew), self._p(dones_after, torch.int64),
                                                _stream_ptr(self.device)))

    def finish_rollout(self, last_states, last_dones=None):
        self._check(self.lib.sdxp_finish_rollout(self.h, self._p(last_states), self._p(last_dones, torch.int64),
                                                 _stream_ptr(self.device)))

    def get_values(self, states, out=None):
        """central value of `states` [N, state_dim] -> f32 [N] (sdxp_get_values)"""
        if out is None:
            out = torch.empty(self.num_actors, device=self.device)
        self._check(self.lib.sdxp_get_values(self.h, self._p(states), self._p(out), _stream_ptr(self.device)))
        return out

    def discount_values(self, last_values, last_dones=None):
        """GAE over the experience buffer (PS:331-336): raw advantages -> t["ADVANTAGES"], returns -> t["RETURNS"]"""
        self._check(self.lib.sdxp_discount_values(self.h, self._p(last_values), self._p(last_dones, torch.int64),
                                                  _stream_ptr(self.device)))

    def prepare_dataset(self):
        """advantage normalisation in place (RC:1645-1651)"""
        self._check(self.lib.sdxp_prepare_dataset(self.h, _stream_ptr(self.device)))

    def update(self):
        self._check(self.lib.sdxp_update(self.h, _stream_ptr(self.device)))

    def update_checked(self):
        """update() + wait + verdict; if the persistent kernel could not run (its 256 workgroups were not co-resident) the library
        has restored its inputs and switched to the hipGraph path: repeat the epoch there.  Returns the implementation used."""
        self.update()
        if self.lib.sdxp_update_status(self.h, _stream_ptr(self.device)) != 0:
            self.update()
            self._check(self.lib.sdxp_update_status(self.h, _stream_ptr(self.device)))
        return self.update_impl()

    def update_status(self):
        """wait for the launched update / multi-rank steps; raises SdxError if a persistent launch could not run"""
        self._check(self.lib.sdxp_update_status(self.h, _stream_ptr(self.device)))

    def update_impl(self):
        """'persistent' (one launch per epoch, register-resident weights), 'graph' (hipGraph of the multi-kernel rank-MB step) or
        'gemm' (minibatch_size > 8: fp32-MFMA GEMM step with explicit gradients, sdxp_bigmb.hip)"""
        return {1: "persistent", 2: "gemm"}.get(self.lib.sdxp_update_impl(self.h), "graph")

    # ---- explicit-gradient path for world_size > 1 (gradients all-reduced by the caller between the two calls)
    def backward(self, which, mb):
        self._check(self.lib.sdxp_backward(self.h, which, mb, _stream_ptr(self.device)))

    def apply(self, which, kl=float("nan")):
        self._check(self.lib.sdxp_apply(self.h, which, C.c_float(kl), _stream_ptr(self.device)))

    def backward_factors(self, mb):
        """multi-rank, factor exchange: this rank's rank-MB factors of minibatch `mb` -> t["FACTORS"] (mb < 0: begin of the epoch)"""
        self._check(self.lib.sdxp_backward_factors(self.h, mb, _stream_ptr(self.device)))

    def grads_from_factors(self):
        """rebuild the SUM over ranks of the minibatch gradients from t["FACTORS_ALL"] (all-gathered by the caller)"""
        self._check(self.lib.sdxp_grads_from_factors(self.h, _stream_ptr(self.device)))

    def apply_factors(self):
        """grads_from_factors() + apply(0, -inf) + apply(1) in three launches"""
        self._check(self.lib.sdxp_apply_factors(self.h, _stream_ptr(self.device)))

    def kl_view(self):
        """1-element f32 view of SdxpCtrl.last_kl inside the STATS tensor (for the scalar KL all-reduce, PS:308-310)"""
        off = Ctrl.last_kl.offset // 4
        return self.t["STATS"][off:off + 1]

    def ctrl(self):
        raw = self.t["STATS"].cpu().numpy().tobytes()
        return Ctrl.from_buffer_copy(raw[:C.sizeof(Ctrl)])

    def get_state(self):
        """optimiser state of the control block: dict(rms_count, ac_t, cv_t, ac_lr, cv_lr) (sdxp_get_state; blocking)"""
        st = _abi.OptState()
        self._check(self.lib.sdxp_get_state(self.h, C.byref(st), _stream_ptr(self.device)))
        return dict(rms_count=st.rms_count, ac_t=st.ac_t, cv_t=st.cv_t, ac_lr=st.ac_lr, cv_lr=st.cv_lr)

    def set_state(self, rms_count=None, ac_t=None, cv_t=None, ac_lr=None, cv_lr=None):
        """write the given items back (the others keep their current values); bias-correction powers follow the step counters"""
        cur = self.get_state()
        for k, v in dict(rms_count=rms_count, ac_t=ac_t, cv_t=cv_t, ac_lr=ac_lr, cv_lr=cv_lr).items():
            if v is not None:
                cur[k] = v
        st = _abi.OptState(float(cur["rms_count"]), int(cur["ac_t"]), int(cur["cv_t"]), float(cur["ac_lr"]), float(cur["cv_lr"]))
        self._check(self.lib.sdxp_set_state(self.h, C.byref(st), _stream_ptr(self.device)))

    def param_count(self, which=0):
        return int(self.lib.sdxp_param_count(self.h, which))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.t.clear()
            self.lib.sdxp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
