"""TEST INFRASTRUCTURE ONLY: builds the product's HIP kernel SOURCES with g++ against a small SIMT emulator (one fiber per GPU
thread, wave collectives resolved per call site; tests/hipemu/include/hip/hip_runtime.h) and runs them on the CPU, so that kernel
logic can be checked against the oracles in the `-m "not gpu"` suite.  Nothing under seqdex_amd/ uses this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "seqdex_amd", "csrc")
_SO = os.path.join(HERE, "libsdx_emu.so")
_SRCS = [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "physics_driver.cpp"), os.path.join(CSRC, "sdx_physics.hip")]
_DEPS = _SRCS + [os.path.abspath(__file__), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(CSRC, "sdx_common.h"),
                 os.path.join(CSRC, "sdx_const_build.h"), os.path.join(ROOT, "include", "seqdex.h")]
# extra g++ flags for every emulator build (e.g. "-DSDX_D_SORT": a compile-time variant of a kernel under test); rebuild with force=True after changing it
_EXTRA = os.environ.get("SDX_EMU_CXXFLAGS", "").split()
_lib = None
# the whole simulator behind the C ABI of include/seqdex.h (sdx_capi + task + physics + camera sources) on the emulator
_SIM_SO = os.path.join(HERE, "libsdx_emu_sim.so")
_SIM_SRCS = [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "sim_driver.cpp")] + \
            [os.path.join(CSRC, f) for f in ("sdx_capi.hip", "sdx_task.hip", "sdx_physics.hip", "sdx_camera.hip")]
_sim_lib = None


def build(force=False):
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in _DEPS):
        return _SO
    objs = []
    for src in _SRCS:
        obj = os.path.join(HERE, os.path.basename(src) + ".emu.o")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-omit-frame-pointer", "-w", "-x", "c++",
                               "-I", os.path.join(HERE, "include"), "-I", CSRC] + _EXTRA + ["-c", src, "-o", obj])
        objs.append(obj)
    # -Bsymbolic: the product library may already be loaded RTLD_GLOBAL in this process and exports the same sdxk_* names
    subprocess.check_call(["g++", "-shared", "-Wl,-Bsymbolic", "-o", _SO] + objs)
    return _SO


def build_sim(force=False):
    deps = _SIM_SRCS + [os.path.abspath(__file__), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "seqdex.h")] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and os.path.exists(_SIM_SO) and all(os.path.getmtime(_SIM_SO) >= os.path.getmtime(d) for d in deps):
        return _SIM_SO
    objs = []
    for src in _SIM_SRCS:
        obj = os.path.join(HERE, os.path.basename(src) + ".emusim.o")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-omit-frame-pointer", "-w", "-x", "c++",
                               "-I", os.path.join(HERE, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "include")] + _EXTRA + ["-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-o", _SIM_SO] + objs)
    return _SIM_SO


_GEMM_SO = os.path.join(HERE, "libsdx_emu_gemm.so")
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"       # host x86 compile: the GEMM header uses ext_vector_type and __bf16, which g++ 11 lacks


def build_gemm(force=False):
    """seqdex_amd/csrc/sdx_gemm_nt.h (k_gemm_nt, k_stage) on the emulator: MFMA as a wave collective, global_load_lds as a per-lane copy"""
    srcs = [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "gemm_driver.cpp")]
    deps = srcs + [os.path.abspath(__file__), os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "include", "hipemu_mfma.h")] + \
        [os.path.join(CSRC, f) for f in ("sdx_gemm_nt.h", "sdx_gemm.h", "sdx_common.h")]
    if not force and os.path.exists(_GEMM_SO) and all(os.path.getmtime(_GEMM_SO) >= os.path.getmtime(d) for d in deps):
        return _GEMM_SO
    objs = []
    for src in srcs:
        obj = os.path.join(HERE, os.path.basename(src) + ".emugemm.o")
        subprocess.check_call([_CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-omit-frame-pointer", "-w", "-x", "c++",
                               "-I", os.path.join(HERE, "include"), "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call([_CLANG, "-shared", "-Wl,-Bsymbolic", "-o", _GEMM_SO] + objs)
    return _GEMM_SO


def sim_lib():
    """ctypes handle of the emulated simulator library with the prototypes of seqdex_amd/_abi.py::load_library"""
    global _sim_lib
    if _sim_lib is None:
        from seqdex_amd import _abi
        lib_ = C.CDLL(build_sim())
        vp, i64p, i32, i32p = C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)
        lib_.sdx_create.argtypes = [C.POINTER(_abi.SceneDesc), i32, i32, C.c_uint64, C.POINTER(vp)]
        lib_.sdx_destroy.argtypes = [vp]
        lib_.sdx_tensor.argtypes = [vp, i32, C.POINTER(vp), i64p, i32p, i32p]
        lib_.sdx_load_initial_states.argtypes = [vp, vp, i32]
        lib_.sdx_set_tvalue_weights.argtypes = [vp, vp, i32]
        lib_.sdx_set_retri_tvalue_weights.argtypes = [vp, vp, i32]
        for n in ["sdx_step", "sdx_pre_physics"]:
            getattr(lib_, n).argtypes = [vp, vp, vp]
        for n in ["sdx_simulate", "sdx_post_physics", "sdx_compute_observations", "sdx_refresh_kinematics", "sdx_render_segmentation"]:
            getattr(lib_, n).argtypes = [vp, vp]
        lib_.sdx_reset_idx.argtypes = [vp, vp, vp, vp]
        lib_.sdx_set_indexed.argtypes = [vp, i32, vp, vp, i32, vp]
        lib_.sdx_num_envs.argtypes = [vp]
        lib_.sdx_last_error.argtypes = [vp]
        lib_.sdx_last_error.restype = C.c_char_p
        _sim_lib = lib_
    return _sim_lib


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.emu_simulate.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def simulate(desc, root, dof, targets, warm=None):
    """same contract as oracle.physics_oracle.simulate (warm: an oracle.physics_oracle.WarmState of the emulated kernel's own cache -
    its key encoding differs from the oracle's, the two are not interchangeable), executed by the emulated k_physics"""
    n = root.shape[0]
    assert root.dtype == np.float32 and root.flags.c_contiguous and dof.flags.c_contiguous
    rb = np.zeros((n, 165, 13), np.float32)
    contact = np.zeros((n, 165, 3), np.float32)
    jac = np.zeros((n, 6, 7), np.float32)
    nc = np.zeros(n, np.int32)
    tg = np.ascontiguousarray(targets, np.float32)
    w = (None, None, None) if warm is None else (_p(warm.count), _p(warm.key), _p(warm.lam))
    lib().emu_simulate(C.byref(desc), C.c_int(n), _p(root), _p(dof), _p(tg), _p(rb), _p(contact), _p(jac), _p(nc), None, *w)
    return rb, contact, jac, nc


def contact_stats(reset=True):
    """SDX_T_CONTACT_STATS of the emulated k_physics launches since the last reset: [largest contact count, substeps that lost contacts,
    substeps rebuilt without speculative contacts, substeps whose pair lists overflowed]"""
    out = np.zeros(4, np.int32)
    lib().emu_cstats(_p(out), C.c_int(1 if reset else 0))
    return out
