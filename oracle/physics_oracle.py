"""TEST INFRASTRUCTURE ONLY: ctypes wrapper around oracle/libsdx_oracle.so (oracle/physics_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "libsdx_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "physics_oracle.c")
        hdr = os.path.join(os.path.dirname(HERE), "include", "seqdex.h")      # the scene descriptor's layout lives there
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            build()
        _lib = C.CDLL(_SO)
        _lib.sdxo_contacts.restype = C.c_int
        _lib.sdxo_max_contacts.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class WarmState:
    """impulse cache of n envs between calls of simulate(); fresh = empty, like a freshly created simulator"""

    def __init__(self, n):
        cap = lib().sdxo_max_contacts()
        self.count = np.zeros(n, np.int32)
        self.key = np.zeros((n, cap), np.uint32)
        self.lam = np.zeros((n, 3, cap), np.float32)

    def clear(self):
        self.count[:] = 0


def simulate(desc, root, dof, targets, warm=None):
    """root [N,142,13], dof [N,23,2], targets [N,23] (float32, modified in place for root/dof).  warm: a WarmState that carries the
    solver's impulses from one call to the next (None: every call starts from an empty cache).
    Returns rb [N,165,13], contact [N,165,3], jac [N,6,7], ncontacts [N]."""
    n = root.shape[0]
    assert root.dtype == np.float32 and root.flags.c_contiguous and dof.flags.c_contiguous
    rb = np.zeros((n, 165, 13), np.float32)
    contact = np.zeros((n, 165, 3), np.float32)
    jac = np.zeros((n, 6, 7), np.float32)
    nc = np.zeros(n, np.int32)
    tg = np.ascontiguousarray(targets, np.float32)
    if warm is None:
        w = (None, None, None)
    else:
        assert warm.count.shape[0] == n
        w = (_p(warm.count), _p(warm.key), _p(warm.lam))
    lib().sdxo_simulate(C.byref(desc), C.c_int(n), _p(root), _p(dof), _p(tg), _p(rb), _p(contact), _p(jac), _p(nc), *w)
    return rb, contact, jac, nc


def kinematics(desc, dof):
    n = dof.shape[0]
    rb = np.zeros((n, 165, 13), np.float32)
    jac = np.zeros((n, 6, 7), np.float32)
    d = np.ascontiguousarray(dof, np.float32)
    lib().sdxo_kinematics(C.byref(desc), C.c_int(n), _p(d), _p(rb), _p(jac))
    return rb, jac


def mass_matrix(desc, q, h):
    H = np.zeros((23, 23), np.float32)
    Hi = np.zeros((23, 23), np.float32)
    qq = np.ascontiguousarray(q, np.float32)
    lib().sdxo_mass_matrix(C.byref(desc), _p(qq), C.c_float(h), _p(H), _p(Hi))
    return H, Hi


def contacts(desc, root_env, dof_env):
    cap = lib().sdxo_max_contacts()
    out = np.zeros((cap, 9), np.float32)
    r = np.ascontiguousarray(root_env, np.float32)
    d = np.ascontiguousarray(dof_env, np.float32)
    total = lib().sdxo_contacts(C.byref(desc), _p(r), _p(d), _p(out), C.c_int(cap))
    return out[:min(total, cap)], total
