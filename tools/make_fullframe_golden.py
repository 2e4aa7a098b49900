#!/usr/bin/env python3
"""Full-size golden frame of the headline configuration from the CPU oracle (oracle/liboracle.so): c4_lucy, 1920x1080,
32 samples per pixel, depth of field on, semantics preset gpu_ref, seed 1 - what `bench.py` renders.  Every ray is traced
in reference order (several minutes on 8 cores), so the ray / node / triangle totals of the frame are the reference's.
Writes tests/golden/render/c4_lucy_1080p32.npz: the 8-bit image (R, G, B planes, row 0 = bottom) and the counters.
Usage: python tools/make_fullframe_golden.py [--threads N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as orc  # noqa: E402
from cuda_raytracing_amd import abi  # noqa: E402
from cuda_raytracing_amd.host import SCENE_DIR, HostScene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "render", "c4_lucy_1080p32.npz"))
    args = ap.parse_args()
    W, H, spp = 1920, 1080, 32
    hs = HostScene(os.path.join(SCENE_DIR, "c4_lucy.txt"))
    st = orc.settings(spp=spp, dof=True, focal=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    t0 = time.time()
    rgba, _, s = orc.render(hs, st, W, H, threads=args.threads)
    dt = time.time() - t0
    counters = {k: int(getattr(s, k)) for k in ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits",
                                                "texture_fetches", "pixels")}
    # planes compress better than interleaved pixels
    np.savez_compressed(args.out, r=rgba[..., 0], g=rgba[..., 1], b=rgba[..., 2], width=W, height=H, spp=spp, seed=1,
                        flags=int(abi.RT_MODE_GPU_REF), focal=1.35, aperture=1.0, oracle_seconds=dt,
                        **{"ctr_" + k: np.uint64(v) for k, v in counters.items()})
    print(f"{args.out}: {os.path.getsize(args.out) / 1e6:.2f} MB, oracle {dt:.0f} s, {counters}")


if __name__ == "__main__":
    main()
