#!/usr/bin/env python3
"""Scene loading and creation times with the three tree builders, and what a frame costs on each tree (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

for n in ("c4_lucy", "c5_final", "t_icosa"):
    t0 = time.perf_counter()
    hs = HostScene(os.path.join(SCENE_DIR, n + ".txt"))
    t1 = time.perf_counter()
    DeviceScene(hs, device=0).close()  # first call of the process: device set-up included
    st = abi.Settings(dof_enabled=0, spp=8, focal_distance=1.0, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
    first = None
    for builder in ("host", "device", "device_sah"):
        t2 = time.perf_counter()
        ds = DeviceScene(hs, device=0, tree_builder=builder)
        t3 = time.perf_counter()
        ds.render_frame(st, 1920, 1080)
        ms = []
        for _ in range(3):
            img, stats = ds.render_frame(st, 1920, 1080)
            ms.append(stats.kernel_ms)
        first = img if first is None else first
        print(f"{n}: load {t1 - t0:.3f} s; create with {builder} trees {1e3 * (t3 - t2):.1f} ms (device part of the tree build {ds.tree_build_ms:.2f} ms); "
              f"1920x1080x8 frame {min(ms):.2f} ms, same image as with the host's trees: {bool(np.array_equal(img, first))}; "
              f"trees, surface-area measure: {ds.check_trees()}", flush=True)
        ds.close()
