#!/usr/bin/env python3
"""Device memory before and after 40 create / render / destroy cycles of a scene (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene
hs = HostScene(os.path.join(SCENE_DIR, "c4_lucy.txt"))
st = abi.Settings(dof_enabled=1, spp=4, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
free0 = None
for i in range(40):
    with DeviceScene(hs, device=0, tree_builder="device" if i % 2 else "host") as ds:
        ds.render_frame(st, 640, 360)
        ds.render_tile(st, 640, 360, (5, 7, 100, 50))
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if i == 4: free0 = free
print("free after 5 iterations", free0, "after 40", free, "delta MB", (free0 - free) / 1e6)
