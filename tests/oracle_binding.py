"""ctypes binding of oracle/liboracle.so (the CPU restatement).  Test infrastructure: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

from cuda_raytracing_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
_P = C.POINTER
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            raise OSError(f"{ORACLE_LIB} is not built: run `make -C oracle liboracle.so`")
        lib = C.CDLL(ORACLE_LIB)
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [_P(abi.SceneDesc), _P(abi.Settings)] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, _P(abi.Stats), C.c_int]
        lib.orc_trace_primary.restype = C.c_int
        lib.orc_trace_primary.argtypes = [_P(abi.SceneDesc), _P(abi.Settings)] + [C.c_int] * 8 + [C.c_void_p] * 5 + [C.c_int]
        lib.orc_philox4x32_10.restype = None
        lib.orc_philox4x32_10.argtypes = [_P(C.c_uint32 * 4), C.c_uint32, C.c_uint32]
        lib.orc_frand_at.restype = C.c_float
        lib.orc_frand_at.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.orc_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def settings(spp=1, dof=False, focal=10.0, aperture=10.0, seed=1, flags=abi.RT_MODE_CPU_REF):
    return abi.Settings(dof_enabled=int(dof), spp=spp, focal_distance=focal, aperture=aperture, seed=seed, flags=flags)


def render(scene, st, tex_w, tex_h, rect=None, cam=None, threads=0):
    """Oracle render of a rectangle.  Returns (rgba [tex_h, tex_w, 4] u8, rgb [h, w, 3] f64, Stats)."""
    x0, y0, w, h = rect or (0, 0, tex_w, tex_h)
    cw, ch = cam or (tex_w, tex_h)
    rgba = np.zeros((tex_h, tex_w, 4), dtype=np.uint8)
    rgb = np.zeros((h, w, 3), dtype=np.float64)
    stats = abi.Stats()
    rc = load().orc_render(scene.desc_ptr, C.byref(st), tex_w, tex_h, cw, ch, x0, y0, w, h, rgba.ctypes.data, rgb.ctypes.data,
                           C.byref(stats), threads)
    assert rc == 0, rc
    return rgba, rgb, stats


def trace_primary(scene, st, tex_w, tex_h, rect=None, cam=None, visit_cap=0):
    x0, y0, w, h = rect or (0, 0, tex_w, tex_h)
    cw, ch = cam or (tex_w, tex_h)
    hit = np.zeros((h, w), dtype=np.int32)
    t = np.zeros((h, w), dtype=np.float64)
    leaf = np.zeros((h, w), dtype=np.int32)
    log = np.zeros((h, w, max(visit_cap, 1)), dtype=np.uint32)
    cnt = np.zeros((h, w), dtype=np.uint32)
    rc = load().orc_trace_primary(scene.desc_ptr, C.byref(st), tex_w, tex_h, cw, ch, x0, y0, w, h, hit.ctypes.data, t.ctypes.data,
                                  leaf.ctypes.data, log.ctypes.data if visit_cap else None, cnt.ctypes.data, visit_cap)
    assert rc == 0, rc
    return hit, t, leaf, log, cnt
