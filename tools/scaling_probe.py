#!/usr/bin/env python3
"""Predict the multi-GPU frame time on one GPU: render every rank's share of the headline frame in turn and
report the slowest share per world size (the ranks run concurrently on a real node, so max = frame time
without the gather)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.device import DeviceScene
from cuda_raytracing_amd.host import SCENE_DIR, HostScene

W, H = 1920, 1080
hs = HostScene(os.path.join(SCENE_DIR, "c4_lucy.txt"))
st = abi.Settings(dof_enabled=1, spp=32, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
ds = DeviceScene(hs, device=0)
ds.set_stream(torch.cuda.current_stream().cuda_stream)
image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
ds.render_frame(st, W, H, tile=32, part=0, n_parts=1, out=image)
for n in (1, 2, 4, 8):
    ks, ws = [], []
    for r in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, s = ds.render_frame(st, W, H, tile=32, part=r, n_parts=n, out=image)
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t0) * 1e3)
        ks.append(s.kernel_ms)
    print(f"world {n}: kernel ms per rank min {min(ks):.2f} max {max(ks):.2f} sum {sum(ks):.2f}; wall per call max {max(ws):.2f}; "
          f"predicted efficiency {ks and (sum(ks) / n) / max(ws) * 100:.0f}% of ideal split", flush=True)

# fixed cost of the exchange on rank 0 besides the transfer itself: one pack kernel per rank, n - 1 unpack kernels on rank 0
lib = abi.load_library()
for n in (2, 4, 8):
    most = max(int(lib.rt_member_pixels(W, H, 32, r, n)) for r in range(n))
    landing = torch.zeros((n, most), dtype=torch.int32, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(2):  # second round is timed
        ev[0].record()
        abi.check(lib.rt_pack_tiles(image.data_ptr(), W, H, 32, 1, n, landing[1].data_ptr()))
        ev[1].record()
        for r in range(1, n):
            abi.check(lib.rt_unpack_tiles(landing[r].data_ptr(), W, H, 32, r, n, image.data_ptr()))
        ev[2].record()
        abi.check(lib.rt_unpack_tiles_all(landing[1].data_ptr(), most * 4, W, H, 32, 1, n, image.data_ptr()))
        ev[3].record()
        torch.cuda.synchronize()
    print(f"world {n}: pack kernel {ev[0].elapsed_time(ev[1]) * 1e3:.0f} us, {n - 1} unpack kernels {ev[1].elapsed_time(ev[2]) * 1e3:.0f} us, "
          f"the same as one launch (what the exchange uses) {ev[2].elapsed_time(ev[3]) * 1e3:.0f} us ({most * 4 / 1e6:.2f} MB per rank)", flush=True)
