#!/usr/bin/env python3
"""What cutting the headline frame into 8 or 16 launches costs: the sum of the kernel times of all parts against the one-launch
frame (run on the GPU box; RT_AMD_LIB selects the library build)."""
import os, sys
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
for n in (1, 8, 16):
    best = None
    for rep in range(3):
        ks = []
        for r in range(n):
            _, s = ds.render_frame(st, W, H, tile=32, part=r, n_parts=n, out=image)
            ks.append(s.kernel_ms)
        tot = sum(ks)
        best = tot if best is None else min(best, tot)
    print(f"{os.environ.get('RT_AMD_LIB','-')} parts {n}: sum of kernel times {best:.2f} ms", flush=True)
