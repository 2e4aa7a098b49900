import sys, time, os
sys.path.insert(0, '.')
import torch
from cuda_raytracing_amd.host import HostScene, SCENE_DIR
from cuda_raytracing_amd.device import DeviceScene
for n in ("c4_lucy", "c5_final", "t_icosa"):
    t0 = time.perf_counter(); hs = HostScene(os.path.join(SCENE_DIR, n + ".txt")); t1 = time.perf_counter()
    ds = DeviceScene(hs, device=0); t2 = time.perf_counter()
    ds2 = DeviceScene(hs, device=0); t3 = time.perf_counter()
    print(n, "load %.3f s, create %.3f s, second create %.3f s" % (t1 - t0, t2 - t1, t3 - t2))
