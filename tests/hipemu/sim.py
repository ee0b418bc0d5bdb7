"""TEST INFRASTRUCTURE ONLY: seqdex_amd.sim.SdxSim on top of the EMULATED simulator library (tests/hipemu: the product's sdx_capi /
task / physics / camera sources compiled by g++ against the SIMT emulator).  Same methods and tensor names as SdxSim, tensors are CPU
torch views of the library's buffers; it lets the golden-vector tests of the task kernels run in the `-m "not gpu"` suite.  This is not a
CPU path of the product: nothing under seqdex_amd/ imports it."""
import ctypes as C

import numpy as np
import torch

from seqdex_amd import _abi
from seqdex_amd.scene import load_scene
from seqdex_amd.sim import SdxError, SdxSim

from . import sim_lib

_NP = {0: np.float32, 1: np.int64, 2: np.int32, 3: np.uint8, 4: np.float64, 5: np.int16}     # sdx_dtype codes of include/seqdex.h


class EmuSim(SdxSim):
    def __init__(self, num_envs, device="cpu", seed=22, scene=None, **desc_overrides):   # noqa: the base constructor wants a GPU
        self.lib = sim_lib()
        self.scene = scene or load_scene()
        self.device = torch.device("cpu")
        self.num_envs = int(num_envs)
        self._desc = self.scene.to_desc(**desc_overrides)
        h = C.c_void_p()
        rc = self.lib.sdx_create(C.byref(self._desc), self.num_envs, 0, C.c_uint64(seed), C.byref(h))
        if rc != 0:
            raise SdxError("emulated sdx_create failed (%d): %s" % (rc, self.lib.sdx_last_error(None).decode()))
        self.h = h
        self._tensors = {}
        for name, tid in _abi.T.items():
            self._tensors[name] = self._wrap(tid)

    def _wrap(self, tid):
        ptr, shape, ndim, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
        self._check(self.lib.sdx_tensor(self.h, tid, C.byref(ptr), shape, C.byref(ndim), C.byref(dt)))
        shp = [shape[i] for i in range(ndim.value)]
        n = int(np.prod(shp))
        npdt = np.dtype(_NP[dt.value])
        buf = (C.c_char * (n * npdt.itemsize)).from_address(ptr.value)
        return torch.from_numpy(np.frombuffer(buf, dtype=npdt, count=n).reshape(shp))

    # the base class insists on CUDA tensors and hands the current CUDA stream over; here everything is host memory, no streams
    def _act_ptr(self, actions):
        assert actions.dtype == torch.float32 and actions.is_contiguous() and actions.shape == (self.num_envs, _abi.NUM_ACTIONS)
        return C.c_void_p(actions.data_ptr())

    def step(self, actions):
        self._check(self.lib.sdx_step(self.h, self._act_ptr(actions), None))

    def pre_physics(self, actions):
        self._check(self.lib.sdx_pre_physics(self.h, self._act_ptr(actions), None))

    def simulate(self):
        self._check(self.lib.sdx_simulate(self.h, None))

    def post_physics(self):
        self._check(self.lib.sdx_post_physics(self.h, None))

    def compute_observations(self):
        self._check(self.lib.sdx_compute_observations(self.h, None))

    def render_segmentation(self):
        self._check(self.lib.sdx_render_segmentation(self.h, None))

    def refresh_kinematics(self):
        self._check(self.lib.sdx_refresh_kinematics(self.h, None))

    def reset_idx(self, env_mask, pile_choice=None):
        assert env_mask.dtype == torch.uint8 and env_mask.numel() == self.num_envs
        pc = C.c_void_p(0)
        if pile_choice is not None:
            assert pile_choice.dtype == torch.int32 and pile_choice.numel() == self.num_envs
            pc = C.c_void_p(pile_choice.data_ptr())
        self._check(self.lib.sdx_reset_idx(self.h, C.c_void_p(env_mask.data_ptr()), pc, None))

    def set_indexed(self, name, src, actor_ids):
        assert name in ("ROOT", "DOF", "TARGETS") and src.dtype == torch.float32 and src.is_contiguous()
        assert actor_ids.dtype == torch.int32 and actor_ids.is_contiguous()
        self._check(self.lib.sdx_set_indexed(self.h, _abi.T[name], C.c_void_p(src.data_ptr()), C.c_void_p(actor_ids.data_ptr()),
                                             int(actor_ids.numel()), None))
