#!/usr/bin/env python3
"""Kernel time of every 120x120 tile of the headline frame, rendered one tile per call (ms, row 0 = bottom)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

W, H, T = 1920, 1080, 120
scene = sys.argv[1] if len(sys.argv) > 1 else "c4_lucy"
hs = HostScene(os.path.join(SCENE_DIR, scene + ".txt"))
st = abi.Settings(dof_enabled=1, spp=32, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
ds = DeviceScene(hs, device=0)
ds.set_stream(torch.cuda.current_stream().cuda_stream)
image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
ds.render_tile(st, W, H, (0, 0, T, T), out=image)
tot = 0.0
for y0 in range(H - T, -1, -T):
    row = []
    for x0 in range(0, W, T):
        _, s = ds.render_tile(st, W, H, (x0, y0, T, T), out=image)
        row.append(s.kernel_ms)
        tot += s.kernel_ms
    print(" ".join(f"{v:5.2f}" for v in row), flush=True)
print("sum", round(tot, 2))
