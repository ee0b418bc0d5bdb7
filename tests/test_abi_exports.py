"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/seqdex.h declares; creating a
handle without a GPU fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "seqdex.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdx(?:p|tv)?_[a-z_0-9]+)\s*\(", txt)))


def test_header_declares_expected_surface():
    fn = declared_functions()
    for name in ["sdx_create", "sdx_step", "sdx_simulate", "sdx_tensor", "sdx_reset_idx", "sdxp_create", "sdxp_act",
                 "sdxp_update", "sdxp_backward", "sdxp_apply"]:
        assert name in fn


def test_library_exports_every_declared_symbol():
    from seqdex_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_abi.LIB_PATH)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(_abi.SDX_EXPORTS) == declared_functions()


def test_struct_layout_matches_header():
    """sizeof(sdx_scene_desc) / sizeof(sdxp_config) seen by the C compiler == the ctypes mirrors."""
    import subprocess
    import tempfile
    from seqdex_amd import _abi
    src = '#include <stdio.h>\n#include "seqdex.h"\nint main(){printf("%zu %zu\\n", sizeof(sdx_scene_desc), sizeof(sdxp_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        a, b = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert int(a) == C.sizeof(_abi.SceneDesc)
    assert int(b) == C.sizeof(_abi.PPOConfig)


def test_no_cpu_fallback():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from seqdex_amd.sim import SdxError, SdxSim
    with pytest.raises(SdxError):
        SdxSim(4)
    # and straight through the C ABI: sdx_create reports SDX_ERR_NO_DEVICE (-3) or a HIP error, never success
    from seqdex_amd import _abi
    from seqdex_amd.scene import load_scene
    lib = _abi.load_library()
    h = C.c_void_p()
    d = load_scene().to_desc()
    rc = lib.sdx_create(C.byref(d), 4, 0, C.c_uint64(1), C.byref(h))
    assert rc < 0 and not h.value
    assert b"no" in lib.sdx_last_error(None).lower() or rc == -2


def test_product_path_does_not_import_oracle():
    """nothing under seqdex_amd/ (the product) may import, link or execute oracle/."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "seqdex_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|libsdx_oracle|#include\s+\"[^\"]*oracle", t, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
