"""Host scene model: thin object wrapper over the C loader in librt_amd.so.

Mirrors what `Raytracer::SetInput` + `Raytracer::CreateAll` give a caller of the reference
(CR/raytracer.h:33-35): a parsed scene whose primitives, lights, camera and per-mesh BVH can be
inspected, and which the device upload consumes.
"""
import ctypes as C
import os

import numpy as np

from . import abi

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
SCENE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")


class HostScene:
    """A scene parsed from the reference's text grammar (owns the C-side object)."""

    def __init__(self, scene_path, asset_dir=ASSET_DIR):
        self._lib = abi.load_library()
        self._h = C.c_void_p()
        ad = asset_dir.encode() if asset_dir is not None else None
        abi.check(self._lib.rt_host_scene_load(os.fspath(scene_path).encode(), ad, C.byref(self._h)))
        self.path = os.fspath(scene_path)

    def close(self):
        if self._h:
            self._lib.rt_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_viewport(self, width, height):
        abi.check(self._lib.rt_host_scene_set_viewport(self._h, width, height))

    @property
    def desc_ptr(self):
        return self._lib.rt_host_scene_desc(self._h)

    @property
    def desc(self):
        return self.desc_ptr.contents

    @property
    def warnings(self):
        n = self._lib.rt_host_scene_warning_count(self._h)
        return [self._lib.rt_host_scene_warning(self._h, i).decode() for i in range(n)]

    # ---- numpy views (valid while the scene is alive) ----
    def nodes(self):
        d = self.desc
        if d.n_nodes == 0:
            return np.zeros(0, dtype=NODE_DTYPE)
        buf = (abi.BvhNode * d.n_nodes).from_address(C.addressof(d.nodes.contents))
        return np.frombuffer(buf, dtype=NODE_DTYPE)

    def leaves(self):
        d = self.desc
        if d.n_leaves == 0:
            return np.zeros(0, dtype=LEAF_DTYPE)
        buf = (abi.BvhLeaf * d.n_leaves).from_address(C.addressof(d.leaves.contents))
        return np.frombuffer(buf, dtype=LEAF_DTYPE)

    def texture(self, i):
        t = self.desc.textures[i]
        buf = (C.c_uint8 * (t.width * t.height * 3)).from_address(C.addressof(t.rgb.contents))
        return np.frombuffer(buf, dtype=np.uint8).reshape(t.height, t.width, 3)

    def canonical_bvh(self, mesh_index):
        """The mesh's tree as [n, 16] float64 rows in the reference's recursive visit order:
        [is_leaf, box min xyz, box max xyz, O1, O2, O3] (interior rows carry zeros for the
        triangle; leaf rows carry the triangle's own bounds)."""
        m = self.desc.meshes[mesh_index]
        nodes = self.nodes()[m.node_base:m.node_base + m.node_count]
        leaves = self.leaves()[m.leaf_base:m.leaf_base + m.leaf_count]
        out = np.zeros((m.node_count + m.leaf_count, 16))
        out[nodes["preorder"], 1:4] = nodes["bmin"]
        out[nodes["preorder"], 4:7] = nodes["bmax"]
        tri = np.stack([leaves["O1"], leaves["O2"], leaves["O3"]], 1)  # [n,3,3]
        lp = leaves["preorder"]
        out[lp, 0] = 1.0
        out[lp, 1:4] = tri.min(1)
        out[lp, 4:7] = tri.max(1)
        out[lp, 7:16] = tri.reshape(-1, 9)
        return out


NODE_DTYPE = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("left", "<u4"), ("skip", "<u4"), ("right", "<u4"), ("preorder", "<u4")])
LEAF_DTYPE = np.dtype([("O1", "<f8", 3), ("O2", "<f8", 3), ("O3", "<f8", 3), ("skip", "<u4"), ("preorder", "<u4"), ("face", "<u4"),
                       ("pad_", "<u4"), ("pad2_", "<f8")])
assert NODE_DTYPE.itemsize == 64 and LEAF_DTYPE.itemsize == 96


def save_bmp(path, rgba):
    """Write an [H, W, 4] uint8 image (row 0 = bottom) as the reference's BmpSave does."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, _ = rgba.shape
    abi.check(abi.load_library().rt_bmp_save_rgba(os.fspath(path).encode(), rgba.ctypes.data, w, h))
