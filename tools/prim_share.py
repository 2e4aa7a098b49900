#!/usr/bin/env python3
"""Which primitive the primary rays of a configuration hit, and the ray totals of a frame (run on the GPU box).
Usage: python tools/prim_share.py [scene] [width] [height] [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

name = sys.argv[1] if len(sys.argv) > 1 else "c4_lucy"
W, H, spp = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((2, 1920), (3, 1080), (4, 32)))
hs = HostScene(os.path.join(SCENE_DIR, name + ".txt"))
st = abi.Settings(dof_enabled=1, spp=spp, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
with DeviceScene(hs, device=0) as ds:
    hit = ds.trace_primary(st, W, H)[0]
    ids, n = np.unique(hit, return_counts=True)
    types = {-1: "background"}
    for i in range(hs.desc.n_primitives):
        types[i] = f"prim {i} type {hs.desc.primitives[i].type}"
    for i, c in zip(ids, n):
        print(f"{types[int(i)]}: {c / hit.size:.1%} of the primary rays")
    ds.render_frame(st, W, H)
    _, s = ds.render_frame(st, W, H)
    print({k: getattr(s, k) for k in ("rays_primary", "rays_secondary", "rays_shadow", "rays_shadow_traced", "shaded_hits", "texture_fetches", "nodes_visited", "tris_tested")},
          f"kernel {s.kernel_ms:.2f} ms")
