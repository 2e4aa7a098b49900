#!/usr/bin/env python3
"""Kernel time of small rectangles of the headline frame: what a launch costs before the first ray (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

W, H = 1920, 1080
hs = HostScene(os.path.join(SCENE_DIR, "c4_lucy.txt"))
st = abi.Settings(dof_enabled=1, spp=32, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
ds = DeviceScene(hs, device=0)
image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
ds.render_frame(st, W, H, out=image)
for side in (8, 32, 64, 128, 256, 512):
    for x0, y0, what in ((0, 0, "background corner"), (900, 500, "centre")):
        ms = []
        for _ in range(5):
            _, s = ds.render_tile(st, W, H, (x0, y0, side, side), out=image)
            ms.append(s.kernel_ms)
        print(f"{side}x{side} at {what}: kernel {min(ms) * 1e3:.0f} us (median {sorted(ms)[2] * 1e3:.0f}), total call {s.total_ms * 1e3:.0f} us, "
              f"{side * side * 32} paths", flush=True)
