"""The C-ABI library loads without a GPU and exports every function include/rt_amd.h declares."""
import ctypes
import os
import re

from cuda_raytracing_amd import abi
from helpers import ROOT


def declared_functions(experiments=False):
    """Functions the header declares; those inside #ifdef RT_EXPERIMENTS only on request (A/B timing builds carry them)."""
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    exp = re.findall(r"#ifdef RT_EXPERIMENTS(.*?)#endif", text, flags=re.S)
    if not experiments:
        text = re.sub(r"#ifdef RT_EXPERIMENTS.*?#endif", "", text, flags=re.S)
    else:
        text = "".join(exp)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(abi.EXPORTED_SYMBOLS)
    assert declared_functions(experiments=True) == sorted(abi._OPTIONAL_SIGNATURES)


def test_shipped_library_carries_no_experiment_entry_points():
    if os.environ.get("RT_AMD_LIB"):
        return  # an A/B build was selected on purpose
    lib = ctypes.CDLL(abi.LIB_PATH)
    for name in declared_functions(experiments=True):
        assert not hasattr(lib, name), name


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(abi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_struct_sizes_match_the_header():
    # sizes the device code static_asserts; a drift would corrupt the scene upload
    assert ctypes.sizeof(abi.BvhNode) == 64 and ctypes.sizeof(abi.BvhLeaf) == 96
    assert ctypes.sizeof(abi.Material) == 104 and ctypes.sizeof(abi.Settings) == 40
    assert ctypes.sizeof(abi.Primitive) == 408 and ctypes.sizeof(abi.Light) == 112 and ctypes.sizeof(abi.Camera) == 136
    assert ctypes.sizeof(abi.SceneDesc) == 256 and ctypes.sizeof(abi.Stats) == 88


def test_last_error_is_thread_local_string():
    lib = abi.load_library()
    out = ctypes.c_void_p()
    rc = lib.rt_host_scene_load(b"/nonexistent/scene.txt", None, ctypes.byref(out))
    assert rc == -2 and b"/nonexistent/scene.txt" in lib.rt_last_error()
