"""Thin Python binding of the sdx_* C ABI (include/seqdex.h): owns a handle and exposes the library-owned
device buffers as zero-copy torch tensors — the equivalent of `gymtorch.wrap_tensor(gym.acquire_*_tensor(sim))`
(GS:237-241,313-322).  torch is plumbing here (device pointers, streams); all arithmetic is in libseqdex_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from .scene import load_scene

_TORCH_DTYPE = {0: (torch.float32, "<f4"), 1: (torch.int64, "<i8"), 2: (torch.int32, "<i4"), 3: (torch.uint8, "|u1"),
                4: (torch.float64, "<f8"), 5: (torch.int16, "<i2")}


class _DevArray:
    """minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer (ROCm uses the same protocol)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_device_pointer(ptr, shape, dtype_code, device):
    tdt, typestr = _TORCH_DTYPE[dtype_code]
    t = torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)
    assert t.data_ptr() == ptr and t.dtype == tdt
    return t


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class SdxError(RuntimeError):
    pass


class SdxSim:
    """One simulator+task instance on one GPU (one process per GPU; envs shard across ranks)."""

    def __init__(self, num_envs, device="cuda:0", seed=22, scene=None, desc=None, **desc_overrides):
        """desc: a ready sdx_scene_desc (Scene.to_desc(...), possibly edited by the caller) instead of the scene's default one"""
        if not torch.cuda.is_available():
            raise SdxError("seqdex_amd needs a ROCm GPU (gfx950); there is no CPU fallback for the product path")
        self.lib = _abi.load_library()
        self.scene = scene or load_scene()
        self.device = torch.device(device)
        self.num_envs = int(num_envs)
        if desc is not None and desc_overrides:
            raise SdxError("SdxSim: pass either a ready `desc` or overrides for Scene.to_desc(), not both (%s would be ignored)"
                           % ", ".join(sorted(desc_overrides)))
        self._desc = desc if desc is not None else self.scene.to_desc(**desc_overrides)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.sdx_create(C.byref(self._desc), self.num_envs, idx, C.c_uint64(seed), C.byref(h))
        if rc != 0:
            raise SdxError("sdx_create failed (%d): %s" % (rc, self.lib.sdx_last_error(None).decode()))
        self.h = h
        self.ring_wrapped = False     # set by ring_rows() when a ring it reads has wrapped (its rows are then in claim order, not serial order)
        self._tensors = {}
        for name, tid in _abi.T.items():
            self._tensors[name] = self._wrap(tid)

    def _wrap(self, tid):
        ptr, shape, ndim, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
        self._check(self.lib.sdx_tensor(self.h, tid, C.byref(ptr), shape, C.byref(ndim), C.byref(dt)))
        return wrap_device_pointer(ptr.value, [shape[i] for i in range(ndim.value)], dt.value, self.device)

    def _check(self, rc):
        if rc != 0:
            raise SdxError("libseqdex_hip error %d: %s" % (rc, self.lib.sdx_last_error(self.h).decode()))

    def tensor(self, name):
        return self._tensors[name]

    def __getattr__(self, name):
        t = self.__dict__.get("_tensors", {})
        if name.upper() in t:
            return t[name.upper()]
        raise AttributeError(name)

    def ring_rows(self, rows, keys, count):
        """the filled rows of one ring in SERIAL order: `rows` [slots, ...] and `keys` [slots] views of a ring and its key tensor
        (SDX_T_*_KEYS: step << 24 | env of the append), `count` appends so far.  The kernels claim ring slots with atomics, so the slot order
        is the hardware's; sorted by key the rows come out as a loop over steps and envs would have written them - the same on every
        run.  A ring that has WRAPPED (count > slots) keeps whichever rows landed last in each slot - which ones depends on the order the
        slots were claimed in, the serial-order guarantee is gone: `self.ring_wrapped` records it (callers that promise determinism
        assert it stayed False; size the run so that it does)."""
        if int(count) > rows.shape[0] and self is not None:
            if not self.ring_wrapped:      # said once per simulator: every consumer of the ring (terminal states, piles, T-value data sets) is affected
                import warnings
                warnings.warn("seqdex_amd: a terminal-state ring wrapped (%d appends into %d slots): its rows are in slot-claim order, not in serial "
                              "(step, env) order - the run is no longer bit-reproducible; harvest more often or size the run to the ring"
                              % (int(count), rows.shape[0]), RuntimeWarning, stacklevel=2)
            self.ring_wrapped = True
        k = int(min(int(count), rows.shape[0]))
        if k == 0:
            return rows[:0].clone()
        order = torch.argsort(keys[:k], stable=True)
        return rows[:k].index_select(0, order)

    # ------------------------------------------------------------------ C ABI calls
    def load_initial_states(self, piles):
        if torch.is_tensor(piles):
            piles = piles.detach().cpu().numpy()
        piles = np.ascontiguousarray(piles, dtype=np.float32)
        assert piles.ndim == 4 and piles.shape[0] == 8 and piles.shape[2:] == (132, 13), piles.shape
        self._check(self.lib.sdx_load_initial_states(self.h, piles.ctypes.data_as(C.c_void_p), piles.shape[1]))

    def set_tvalue_weights(self, state_dict_or_flat):
        if isinstance(state_dict_or_flat, dict):
            parts = []
            for n in ["linear1", "linear2", "linear3", "output_layer"]:
                parts.append(np.asarray(state_dict_or_flat[n + ".weight" if n + ".weight" in state_dict_or_flat
                                                           else n + "_weight"], dtype=np.float32).ravel())
                parts.append(np.asarray(state_dict_or_flat[n + ".bias" if n + ".bias" in state_dict_or_flat
                                                           else n + "_bias"], dtype=np.float32).ravel())
            flat = np.concatenate(parts)
        else:
            flat = np.ascontiguousarray(state_dict_or_flat, dtype=np.float32)
        assert flat.size == _abi.TV_PARAMS
        self._check(self.lib.sdx_set_tvalue_weights(self.h, flat.ctypes.data_as(C.c_void_p), flat.size))

    def set_retri_tvalue_weights(self, state_dict_or_flat):
        """BlockAssemblySearch: RetriGraspTValue(650, 2) parameters (state_dict with linear1/2/3 + output_layer, or the flat packing)"""
        if isinstance(state_dict_or_flat, dict):
            parts = []
            for n in ["linear1", "linear2", "linear3", "output_layer"]:
                parts.append(np.asarray(state_dict_or_flat[n + ".weight"], dtype=np.float32).ravel())
                parts.append(np.asarray(state_dict_or_flat[n + ".bias"], dtype=np.float32).ravel())
            flat = np.concatenate(parts)
        else:
            flat = np.ascontiguousarray(state_dict_or_flat, dtype=np.float32)
        assert flat.size == _abi.RETRI_TV_PARAMS, flat.size
        self._check(self.lib.sdx_set_retri_tvalue_weights(self.h, flat.ctypes.data_as(C.c_void_p), flat.size))

    def _act_ptr(self, actions):
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert actions.shape == (self.num_envs, _abi.NUM_ACTIONS)
        return C.c_void_p(actions.data_ptr())

    def step(self, actions):
        self._check(self.lib.sdx_step(self.h, self._act_ptr(actions), _stream_ptr(self.device)))

    def pre_physics(self, actions):
        self._check(self.lib.sdx_pre_physics(self.h, self._act_ptr(actions), _stream_ptr(self.device)))

    def simulate(self):
        self._check(self.lib.sdx_simulate(self.h, _stream_ptr(self.device)))

    def post_physics(self):
        self._check(self.lib.sdx_post_physics(self.h, _stream_ptr(self.device)))

    def compute_observations(self):
        self._check(self.lib.sdx_compute_observations(self.h, _stream_ptr(self.device)))

    def render_segmentation(self):
        """BlockAssemblySearch: gym.render_all_camera_sensors + pixel statistics -> SEG_IMAGE, SEG_PIXELS, EMERGENCE"""
        self._check(self.lib.sdx_render_segmentation(self.h, _stream_ptr(self.device)))

    def refresh_kinematics(self):
        self._check(self.lib.sdx_refresh_kinematics(self.h, _stream_ptr(self.device)))

    def reset_idx(self, env_mask, pile_choice=None):
        assert env_mask.is_cuda and env_mask.dtype == torch.uint8 and env_mask.numel() == self.num_envs
        pc = C.c_void_p(0)
        if pile_choice is not None:
            assert pile_choice.is_cuda and pile_choice.dtype == torch.int32 and pile_choice.numel() == self.num_envs
            pc = C.c_void_p(pile_choice.data_ptr())
        self._check(self.lib.sdx_reset_idx(self.h, C.c_void_p(env_mask.data_ptr()), pc, _stream_ptr(self.device)))

    def set_indexed(self, name, src, actor_ids):
        """gym.set_{actor_root_state,dof_state,dof_position_target}_tensor_indexed: name in ROOT / DOF / TARGETS, src the full-size
        tensor (the library's own view after in-place edits, or a tensor of the caller's), actor_ids int32 sim-domain actor indices"""
        assert name in ("ROOT", "DOF", "TARGETS"), name
        own = self._tensors[name]
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.numel() == own.numel(), (src.shape, own.shape)
        assert actor_ids.is_cuda and actor_ids.dtype == torch.int32 and actor_ids.is_contiguous()
        self._check(self.lib.sdx_set_indexed(self.h, _abi.T[name], C.c_void_p(src.data_ptr()), C.c_void_p(actor_ids.data_ptr()),
                                             int(actor_ids.numel()), _stream_ptr(self.device)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self._tensors.clear()
            self.lib.sdx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
