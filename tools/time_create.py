#!/usr/bin/env python3
"""Scene loading and creation times with both tree builders (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

for n in ("c4_lucy", "c5_final", "t_icosa"):
    t0 = time.perf_counter()
    hs = HostScene(os.path.join(SCENE_DIR, n + ".txt"))
    t1 = time.perf_counter()
    DeviceScene(hs, device=0).close()  # first call of the process: device set-up included
    for builder in ("host", "device"):
        t2 = time.perf_counter()
        ds = DeviceScene(hs, device=0, tree_builder=builder)
        t3 = time.perf_counter()
        print(f"{n}: load {t1 - t0:.3f} s; create with {builder} trees {1e3 * (t3 - t2):.1f} ms (device part of the tree build {ds.tree_build_ms:.2f} ms)")
        ds.close()
