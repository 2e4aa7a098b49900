"""ctypes mirror of include/rt_amd.h and loader of the in-tree librt_amd.so.

Plumbing only: every structure here is a field-for-field copy of the C header, and every
function is bound with its C signature.  There is no Python fallback for any of them: if the
shared library is missing, `load_library()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
RT_MAX_DEPTH = 10
RT_BEZIER_MAX_CTRL = 10
RT_REF_LEAF = 0x80000000
RT_REF_END = 0xFFFFFFFF

RT_PRIM_SPHERE, RT_PRIM_PLANE, RT_PRIM_SQUARE, RT_PRIM_CYLINDER, RT_PRIM_BEZIER, RT_PRIM_TRIANGLE, RT_PRIM_MESH = range(7)
RT_LIGHT_POINT, RT_LIGHT_SQUARE, RT_LIGHT_SPHERE = range(3)

RT_F_NORMALIZE_PRIMARY = 0x001
RT_F_SKY_Y = 0x002
RT_F_TIE_LATER_WINS = 0x004
RT_F_SHADOW_SELF_SKIP = 0x008
RT_F_SQUARE_UV_NO_ORIGIN = 0x010
RT_F_TEX_U8_TAPS = 0x020
RT_F_PIXEL_MAP_TRUNC = 0x040
RT_F_MESH_RAW_DIR = 0x080
RT_F_BEZIER_STACKED = 0x100
RT_F_STRICT_TRAVERSAL = 0x200
RT_MODE_CPU_REF = 0
RT_MODE_GPU_REF = (
    RT_F_NORMALIZE_PRIMARY | RT_F_SKY_Y | RT_F_TIE_LATER_WINS | RT_F_SHADOW_SELF_SKIP | RT_F_SQUARE_UV_NO_ORIGIN
    | RT_F_TEX_U8_TAPS | RT_F_PIXEL_MAP_TRUNC | RT_F_MESH_RAW_DIR | RT_F_BEZIER_STACKED
)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Material(C.Structure):
    _fields_ = [("color", Vec3), ("absor", Vec3), ("refl", C.c_double), ("refr", C.c_double), ("diff", C.c_double),
                ("spec", C.c_double), ("rindex", C.c_double), ("drefl", C.c_double), ("texture", C.c_int32), ("pad_", C.c_int32)]


class _Sphere(C.Structure):
    _fields_ = [("O", Vec3), ("De", Vec3), ("Dc", Vec3), ("R", C.c_double)]


class _Plane(C.Structure):
    _fields_ = [("N", Vec3), ("Dx", Vec3), ("Dy", Vec3), ("R", C.c_double)]


class _Square(C.Structure):
    _fields_ = [("O", Vec3), ("Dx", Vec3), ("Dy", Vec3)]


class _Cylinder(C.Structure):
    _fields_ = [("O1", Vec3), ("O2", Vec3), ("R", C.c_double)]


class _Bezier(C.Structure):
    _fields_ = [("O1", Vec3), ("O2", Vec3), ("N", Vec3), ("Nx", Vec3), ("Ny", Vec3), ("R_c", C.c_double), ("degree", C.c_int32),
                ("pad_", C.c_int32), ("R", C.c_double * RT_BEZIER_MAX_CTRL), ("Z", C.c_double * RT_BEZIER_MAX_CTRL)]


class _Triangle(C.Structure):
    _fields_ = [("O1", Vec3), ("O2", Vec3), ("O3", Vec3), ("N", Vec3)]


class _MeshRef(C.Structure):
    _fields_ = [("mesh", C.c_int32), ("pad_", C.c_int32)]


class _PrimUnion(C.Union):
    _fields_ = [("sphere", _Sphere), ("plane", _Plane), ("square", _Square), ("cylinder", _Cylinder), ("bezier", _Bezier),
                ("triangle", _Triangle), ("mesh", _MeshRef)]


class Primitive(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_light", C.c_int32), ("material", Material), ("u", _PrimUnion)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("light_primitive", C.c_int32), ("color", Vec3), ("O", Vec3), ("Dx", Vec3), ("Dy", Vec3),
                ("R", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("O", Vec3), ("N", Vec3), ("Dx", Vec3), ("Dy", Vec3), ("W", C.c_int32), ("H", C.c_int32),
                ("shade_quality", C.c_double), ("drefl_quality", C.c_double), ("lens_W", C.c_double), ("lens_H", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("left", C.c_uint32), ("skip", C.c_uint32),
                ("right", C.c_uint32), ("preorder", C.c_uint32)]


class BvhLeaf(C.Structure):
    _fields_ = [("O1", Vec3), ("O2", Vec3), ("O3", Vec3), ("skip", C.c_uint32), ("preorder", C.c_uint32), ("face", C.c_uint32),
                ("pad_", C.c_uint32), ("pad2_", C.c_double)]


class Mesh(C.Structure):
    _fields_ = [("root", C.c_uint32), ("node_base", C.c_uint32), ("node_count", C.c_uint32), ("leaf_base", C.c_uint32),
                ("leaf_count", C.c_uint32), ("depth", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("background_top", Vec3), ("background_bottom", Vec3), ("camera", Camera), ("n_lights", C.c_int32),
                ("n_primitives", C.c_int32), ("n_textures", C.c_int32), ("n_meshes", C.c_int32), ("n_nodes", C.c_uint32),
                ("n_leaves", C.c_uint32), ("lights", C.POINTER(Light)), ("primitives", C.POINTER(Primitive)),
                ("textures", C.POINTER(Texture)), ("meshes", C.POINTER(Mesh)), ("nodes", C.POINTER(BvhNode)),
                ("leaves", C.POINTER(BvhLeaf))]


class Settings(C.Structure):
    _fields_ = [("dof_enabled", C.c_int32), ("spp", C.c_int32), ("focal_distance", C.c_double), ("aperture", C.c_double),
                ("seed", C.c_uint64), ("flags", C.c_uint32), ("pad_", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("texture_fetches", C.c_uint64), ("pixels", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("rays_shadow_traced", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    @property
    def rays(self):
        """Rays the reference traces for this work: every shadow sample counts, traced here or decided without a ray."""
        return self.rays_primary + self.rays_secondary + self.rays_shadow

    @property
    def rays_traced(self):
        """Rays that were actually intersected with the scene."""
        return self.rays_primary + self.rays_secondary + self.rays_shadow_traced


# sizes the C side static_asserts as well
assert C.sizeof(BvhNode) == 64 and C.sizeof(BvhLeaf) == 96 and C.sizeof(Vec3) == 24

_P = C.POINTER
_SIGNATURES = {
    "rt_host_scene_load": (C.c_int, [C.c_char_p, C.c_char_p, _P(C.c_void_p)]),
    "rt_host_scene_set_viewport": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rt_host_scene_desc": (_P(SceneDesc), [C.c_void_p]),
    "rt_host_scene_warning_count": (C.c_int, [C.c_void_p]),
    "rt_host_scene_warning": (C.c_char_p, [C.c_void_p, C.c_int]),
    "rt_host_scene_free": (None, [C.c_void_p]),
    "rt_bmp_save_rgba": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "rt_init": (C.c_int, [C.c_int]),
    "rt_set_stream": (C.c_int, [C.c_void_p]),
    "rt_set_tree_builder": (C.c_int, [C.c_int]),
    "rt_scene_tree_build_ms": (C.c_double, [C.c_void_p]),
    "rt_scene_check_trees": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "rt_scene_shared_walk": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rt_scene_waves_per_simd": (C.c_int, [C.c_void_p]),
    "rt_scene_create": (C.c_int, [_P(SceneDesc), _P(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render_tile": (C.c_int, [C.c_void_p, _P(Settings)] + [C.c_int] * 8 + [C.c_void_p, _P(Stats)]),
    "rt_render_frame": (C.c_int, [C.c_void_p, _P(Settings)] + [C.c_int] * 7 + [C.c_void_p, _P(Stats)]),
    "rt_render_tile_noalloc": (C.c_int, [C.c_void_p, _P(Settings)] + [C.c_int] * 8 + [C.c_void_p, _P(Stats)]),
    "rt_device_count": (C.c_int, []),
    "rt_scene_group_create": (C.c_int, [_P(SceneDesc), C.c_int, _P(C.c_int), _P(C.c_void_p)]),
    "rt_scene_group_destroy": (None, [C.c_void_p]),
    "rt_scene_group_size": (C.c_int, [C.c_void_p]),
    "rt_scene_group_member_ms": (C.c_double, [C.c_void_p, C.c_int]),
    "rt_render_frame_multi": (C.c_int, [C.c_void_p, _P(Settings)] + [C.c_int] * 5 + [C.c_void_p, _P(Stats)]),
    "rt_member_pixels": (C.c_longlong, [C.c_int] * 5),
    "rt_pack_tiles": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "rt_unpack_tiles": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "rt_unpack_tiles_all": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rt_trace_primary": (C.c_int, [C.c_void_p, _P(Settings)] + [C.c_int] * 8 + [C.c_void_p] * 5 + [C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_build_id": (C.c_char_p, []),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
# only in libraries built with -DRT_EXPERIMENTS (A/B timing builds): bound when present
_OPTIONAL_SIGNATURES = {
    "rt_scene_deferred_shadows": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
}

_lib = None


def load_library(path=None):
    """Load librt_amd.so (once) and bind the C ABI.  Raises OSError when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _OPTIONAL_SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def has_experiments(lib=None):
    """True if the loaded library was built with -DRT_EXPERIMENTS (it then carries the three-kernel render)."""
    return hasattr(lib or load_library(), "rt_scene_deferred_shadows")


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def check(code):
    if code != RT_OK:
        raise RtError(code, load_library().rt_last_error().decode("utf-8", "replace"))
    return code
