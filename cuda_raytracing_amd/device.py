"""Device scene + render calls: object wrapper over the device half of the C ABI.

`DeviceScene` plays the role of the reference's `SceneContainer` (CR/Renderer.h:19-26): it owns
the uploaded scene and frees it on close.  Rendering goes through `rt_render_tile` /
`rt_render_frame` only; there is no CPU path here, and construction fails if the HIP library or a
gfx950 device is missing.
"""
import ctypes as C

import numpy as np

from . import abi


class DeviceScene:
    def __init__(self, host_scene, device=0, tree_builder=None):
        """tree_builder: None = the library's default, "host", "device" (Morton order) or "device_sah" (rt_set_tree_builder)."""
        import torch  # noqa: F401  plumbing: loads the HIP runtime that librt_amd.so then binds to

        self._lib = abi.load_library()
        abi.check(self._lib.rt_init(device))
        if tree_builder is not None:
            abi.check(self._lib.rt_set_tree_builder({"host": 0, "device": 1, "device_sah": 2}[tree_builder]))
        self.device = device
        self._h = C.c_void_p()
        abi.check(self._lib.rt_scene_create(host_scene.desc_ptr, C.byref(self._h)))
        self.host = host_scene

    def check_trees(self):
        """(number of four-wide trees, their surface-area measure); raises if a tree is broken (rt_scene_check_trees)."""
        n, cost = C.c_int(0), C.c_double(0.0)
        abi.check(self._lib.rt_scene_check_trees(self._h, C.byref(n), C.byref(cost)))
        return n.value, cost.value

    @property
    def tree_build_ms(self):
        return float(self._lib.rt_scene_tree_build_ms(self._h))

    @property
    def shared_walk(self):
        """(choice, plain_ms, shared_ms) of the shadow-walk variant: choice -1 = not made yet, 0 = plain, 1 = shared."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        choice = int(self._lib.rt_scene_shared_walk(self._h, C.byref(a), C.byref(b)))
        return choice, a.value, b.value

    @property
    def deferred_shadows(self):
        """(in use, calls that fell back to the one-kernel render) of the three-kernel render; (0, 0) with a library that
        was not built with -DRT_EXPERIMENTS (the shipped one): it has no such render."""
        if not abi.has_experiments(self._lib):
            return 0, 0
        n = C.c_int(0)
        on = int(self._lib.rt_scene_deferred_shadows(self._h, C.byref(n)))
        return on, n.value

    @property
    def waves_per_simd(self):
        return int(self._lib.rt_scene_waves_per_simd(self._h))

    def close(self):
        if self._h:
            self._lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_handle):
        abi.check(self._lib.rt_set_stream(C.c_void_p(hip_stream_handle)))

    @staticmethod
    def _ptr(out):
        """Destination pointer of a numpy array (host) or a torch CUDA uint8 tensor (device)."""
        if isinstance(out, np.ndarray):
            assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
            return out.ctypes.data
        assert out.is_contiguous() and out.element_size() == 1
        return out.data_ptr()

    def render_tile(self, st, tex_w, tex_h, rect, cam=None, out=None):
        """launchRayTraceKernel (CR/RayTracer.cuh:12): render rectangle (x0, y0, w, h)."""
        x0, y0, w, h = rect
        cw, ch = cam or (tex_w, tex_h)
        if out is None:
            out = np.zeros((tex_h, tex_w, 4), dtype=np.uint8)
        stats = abi.Stats()
        abi.check(self._lib.rt_render_tile(self._h, C.byref(st), tex_w, tex_h, cw, ch, x0, y0, w, h, self._ptr(out), C.byref(stats)))
        return out, stats

    def render_frame(self, st, tex_w, tex_h, cam=None, tile=32, part=0, n_parts=1, out=None):
        """launchRayTraceKernelParallel (CR/RayTracer.cuh:13) for the tiles owned by `part`."""
        cw, ch = cam or (tex_w, tex_h)
        if out is None:
            out = np.zeros((tex_h, tex_w, 4), dtype=np.uint8)
        stats = abi.Stats()
        abi.check(self._lib.rt_render_frame(self._h, C.byref(st), tex_w, tex_h, cw, ch, tile, part, n_parts, self._ptr(out), C.byref(stats)))
        return out, stats

    def render_tile_noalloc(self, st, tex_w, tex_h, rect, out, cam=None):
        """rt_render_tile_noalloc: like render_tile, into a device image (`out`: torch CUDA uint8 tensor), without any
        device allocation on the library's side."""
        x0, y0, w, h = rect
        cw, ch = cam or (tex_w, tex_h)
        stats = abi.Stats()
        abi.check(self._lib.rt_render_tile_noalloc(self._h, C.byref(st), tex_w, tex_h, cw, ch, x0, y0, w, h, self._ptr(out), C.byref(stats)))
        return out, stats

    def trace_primary(self, st, tex_w, tex_h, rect=None, cam=None, visit_cap=0):
        x0, y0, w, h = rect or (0, 0, tex_w, tex_h)
        cw, ch = cam or (tex_w, tex_h)
        hit = np.zeros((h, w), dtype=np.int32)
        t = np.zeros((h, w), dtype=np.float64)
        leaf = np.zeros((h, w), dtype=np.int32)
        log = np.zeros((h, w, max(visit_cap, 1)), dtype=np.uint32)
        cnt = np.zeros((h, w), dtype=np.uint32)
        abi.check(self._lib.rt_trace_primary(self._h, C.byref(st), tex_w, tex_h, cw, ch, x0, y0, w, h, hit.ctypes.data, t.ctypes.data,
                                             leaf.ctypes.data, log.ctypes.data if visit_cap else None, cnt.ctypes.data, visit_cap))
        return hit, t, leaf, log, cnt


class SceneGroup:
    """rt_scene_group: the scene on several GPUs of this process; `render_frame` is rt_render_frame_multi."""

    def __init__(self, host_scene, n_gpus=None, devices=None):
        import torch  # noqa: F401  plumbing: loads the HIP runtime that librt_amd.so then binds to

        self._lib = abi.load_library()
        if devices is not None:
            n_gpus = len(devices)
            arr = (C.c_int * n_gpus)(*devices)
        else:
            n_gpus = n_gpus or int(self._lib.rt_device_count())
            arr = None
        self._h = C.c_void_p()
        abi.check(self._lib.rt_scene_group_create(host_scene.desc_ptr, n_gpus, arr, C.byref(self._h)))
        self.host = host_scene

    @property
    def size(self):
        return int(self._lib.rt_scene_group_size(self._h))

    @property
    def member_ms(self):
        return [float(self._lib.rt_scene_group_member_ms(self._h, k)) for k in range(self.size)]

    def render_frame(self, st, tex_w, tex_h, cam=None, tile=32, out=None):
        cw, ch = cam or (tex_w, tex_h)
        if out is None:
            out = np.zeros((tex_h, tex_w, 4), dtype=np.uint8)
        stats = abi.Stats()
        abi.check(self._lib.rt_render_frame_multi(self._h, C.byref(st), tex_w, tex_h, cw, ch, tile, DeviceScene._ptr(out), C.byref(stats)))
        return out, stats

    def close(self):
        if self._h:
            self._lib.rt_scene_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
