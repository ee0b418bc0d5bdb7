"""Host-side mirror of the reference's transition-value trainer `TValue_Trainer`
(policy_sequencing/transition_value_trainer.py:127-248 = TT): same method names and protocol
(init_TValue_function / train_rollout, 512 + 512 samples per batch, +-0.05 noise and renormalisation, BCEWithLogitsLoss on one-hot
[failure, success] labels, Adam 1e-3, the last 100 successes held out for validation), every step inside libseqdex_hip.so
(csrc/sdx_tvtrain.hip).  The datasets are arrays [n, 4] of camera-frame target quaternions: the reference reads them from an HDF5 file
(h5py is not installed here); this class takes tensors / numpy arrays / an .npz with `success` and `failure`, or the rings a task
instance filled (`from_task`)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from .sim import SdxError, _stream_ptr, wrap_device_pointer

TV_T = dict(PARAMS=0, GRADS=1, ADAM_M=2, ADAM_V=3, BATCH=4, LOSS=5, OUTPUT=6)
LAYERS = (("linear1", 256, 4), ("linear2", 128, 256), ("linear3", 64, 128), ("output_layer", 2, 64))   # terminal_value_function.py:30-37


def flat_from_state_dict(sd):
    parts = []
    for name, out, inn in LAYERS:
        w = torch.as_tensor(sd[name + ".weight"]).float().reshape(out, inn)
        parts += [w.reshape(-1), torch.as_tensor(sd[name + ".bias"]).float().reshape(out)]
    return torch.cat(parts)


def state_dict_from_flat(flat):
    flat, sd, o = torch.as_tensor(flat).float().cpu(), {}, 0
    for name, out, inn in LAYERS:
        sd[name + ".weight"] = flat[o:o + out * inn].reshape(out, inn).clone(); o += out * inn
        sd[name + ".bias"] = flat[o:o + out].clone(); o += out
    assert o == _abi.TV_PARAMS
    return sd


class TValue_Trainer:
    def __init__(self, data, device="cuda:0", seed=0, valid_holdout=100):
        if not torch.cuda.is_available():
            raise SdxError("the T-value trainer runs on the GPU only (libseqdex_hip.so has no CPU path)")
        if isinstance(data, str):
            z = np.load(data)
            succ, fail = z["success"], z["failure"]
        else:
            succ, fail = data
        self.device = torch.device(device)
        succ = torch.as_tensor(succ, dtype=torch.float32).reshape(-1, 4).to(self.device)
        fail = torch.as_tensor(fail, dtype=torch.float32).reshape(-1, 4).to(self.device)
        if succ.shape[0] <= valid_holdout or fail.shape[0] == 0:
            raise ValueError("TValue_Trainer: need more than %d success rows and at least one failure row" % valid_holdout)
        self.valid_data = succ[-valid_holdout:].clone()                    # TT:170-171
        self.success_data = succ[:-valid_holdout].contiguous()
        self.failure_data = fail.contiguous()
        self.num_success_data, self.num_failure_data = self.success_data.shape[0], self.failure_data.shape[0]
        self.input_dim = 4
        self.lib = _abi.load_library()
        self.seed = seed
        self.h = None

    @classmethod
    def from_task(cls, task, **kw):
        """datasets = the SDX_T_TV_SUCCESS / SDX_T_TV_FAILURE rings that the task's reset kernels filled (GS:1404-1438, IS:1392-1410)"""
        s = task.sim
        cnt = s.TV_COUNT.cpu().numpy()
        # rows in serial (step, env) order, not in the order the ring slots happened to be claimed in: the fit's sampler indexes the
        # datasets by position, so the same outcomes must sit at the same positions on every run
        return cls((s.ring_rows(s.TV_SUCCESS, s.TV_KEYS[0], cnt[0]), s.ring_rows(s.TV_FAILURE, s.TV_KEYS[1], cnt[1])), device=str(s.device), **kw)

    def init_TValue_function(self, task_name="grasping_insertion", rollout=100000, state_dict=None, batch_size=1024, lr=0.001):
        self.batch_size, self.succ_batch_size, self.fail_batch_size = batch_size, batch_size // 2, batch_size // 2   # TT:190-192
        self.valid_batch_size = self.valid_data.shape[0]
        self.rollout, self.lr = rollout, lr
        self.t_value_save_path = "./intermediate_state/{}_t_value/".format(task_name)                       # TT:188
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.sdxtv_create(batch_size, idx, C.c_uint64(self.seed), C.byref(h))
        if rc != 0:
            raise SdxError("sdxtv_create failed (%d): %s" % (rc, self.lib.sdxtv_last_error(None).decode()))
        self.h = h
        self.t = {}
        for name, tid in TV_T.items():
            p, shp, nd, dt = C.c_void_p(), (C.c_int64 * 4)(), C.c_int32(), C.c_int32()
            self._check(self.lib.sdxtv_tensor(self.h, tid, C.byref(p), shp, C.byref(nd), C.byref(dt)))
            self.t[name] = wrap_device_pointer(p.value, list(shp)[:nd.value], dt.value, self.device)
        if state_dict is None:                                  # nn.Linear default init of GraspInsertTValue(4, 2) (TT:181)
            g = torch.Generator().manual_seed(self.seed)
            sd = {}
            for name, out, inn in LAYERS:
                b = 1.0 / np.sqrt(inn)
                sd[name + ".weight"] = (torch.rand(out, inn, generator=g) * 2 - 1) * b
                sd[name + ".bias"] = (torch.rand(out, generator=g) * 2 - 1) * b
            state_dict = sd
        self.load_state_dict(state_dict)
        self.success_buf = torch.zeros(batch_size, 2)                                                        # TT:203-205
        self.success_buf[:self.succ_batch_size, 1] = 1
        self.success_buf[self.succ_batch_size:, 0] = 1
        self.t_value_obs_buf = self.t["BATCH"]
        self.losses = []

    def _check(self, rc):
        if rc != 0:
            raise SdxError("sdxtv call failed (%d): %s" % (rc, self.lib.sdxtv_last_error(self.h).decode()))

    def load_state_dict(self, sd):
        self.t["PARAMS"].copy_(flat_from_state_dict(sd).to(self.device))

    def state_dict(self):
        return state_dict_from_flat(self.t["PARAMS"])

    # ---- the pieces of one iteration (exposed for the parity tests)
    def sample(self):
        self._check(self.lib.sdxtv_sample(self.h, C.c_void_p(self.success_data.data_ptr()), self.num_success_data,
                                          C.c_void_p(self.failure_data.data_ptr()), self.num_failure_data, _stream_ptr(self.device)))

    def step(self):
        self._check(self.lib.sdxtv_step(self.h, C.c_float(self.lr), _stream_ptr(self.device)))

    def predict(self, x):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        out = torch.empty(x.shape[0], 2, device=self.device)
        self._check(self.lib.sdxtv_predict(self.h, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(out.data_ptr()), _stream_ptr(self.device)))
        return out

    def validate(self):
        """TT:233-246: share of the held-out successes whose success logit beats the failure logit"""
        p = torch.sigmoid(self.predict(self.valid_data))
        return float((p[:, 0] < p[:, 1]).float().mean())

    def train_rollout(self, iters=None, validate_every=10000, save=False, verbose=False):
        iters = self.rollout if iters is None else iters
        done = 0
        while done < iters:
            n = min(validate_every, iters - done)
            self._check(self.lib.sdxtv_train(self.h, C.c_void_p(self.success_data.data_ptr()), self.num_success_data,
                                             C.c_void_p(self.failure_data.data_ptr()), self.num_failure_data, n, C.c_float(self.lr),
                                             _stream_ptr(self.device)))
            done += n
            torch.cuda.synchronize(self.device)
            loss = float(self.t["LOSS"][0])
            self.losses.append(loss)
            self.valid_t_value_success_rate = self.validate()
            if verbose:
                print("t_value_udpate_iter: ", done, "loss: ", loss, "valid_t_value_success_rate: ", self.valid_t_value_success_rate)
            if save:                                             # TT:247
                os.makedirs(self.t_value_save_path, exist_ok=True)
                torch.save(self.state_dict(), self.t_value_save_path + "/grasp_insert_TValue_{}_{}.pt".format(done, self.valid_t_value_success_rate))
        return self.losses[-1] if self.losses else None

    def close(self):
        if self.h is not None:
            self.lib.sdxtv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
