#!/usr/bin/env python3
"""Full-size golden frames of bench configurations from the CPU oracle (oracle/liboracle.so), with the settings `bench.py`
renders them with (semantics preset gpu_ref, seed 1): headline = c4_lucy 1920x1080x32 spp, c2 = c2_textured_mirror
1280x720x128, c3 = c3_orange 1920x1080x64, c5x4 = c5_final 1920x1080 at 4 samples per pixel, c5 = the same at its full 128 (over an hour of oracle), c4bands = c4_lucy 3840x2160x32 on
34 bands of 8 rows spread evenly over the frame (rows 64 k + 24 .. 64 k + 31; the whole frame would take the oracle an hour).  Every ray is traced in reference order (minutes on 8 cores), so the ray / node /
triangle totals of a frame are the reference's.  Writes tests/golden/render/<scene>_<height>p<spp>.npz: the 8-bit image
(R, G, B planes, row 0 = bottom) and the counters.
c4 = the whole 3840x2160x32 frame (over an hour as well).
Usage: python tools/make_fullframe_golden.py [--config headline|c2|c3|c5x4|c5|c4bands|c4] [--threads N]"""
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


BAND_ROWS, BAND_PITCH, BAND_FIRST = 8, 64, 24


def bands(args):
    """C4 (3840x2160x32) on every eighth band of 8 rows: the bands stacked into one image, counters summed over the bands."""
    W, H, spp, aperture, focal = 3840, 2160, 32, 1.0, 1.35
    hs = HostScene(os.path.join(SCENE_DIR, "c4_lucy.txt"))
    st = orc.settings(spp=spp, dof=True, focal=focal, aperture=aperture, seed=1, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    keys = ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits", "texture_fetches", "pixels")
    total = dict.fromkeys(keys, 0)
    rows = []
    t0 = time.time()
    y0s = list(range(BAND_FIRST, H - BAND_ROWS + 1, BAND_PITCH))
    for y0 in y0s:
        rgba, _, s = orc.render(hs, st, W, H, rect=(0, y0, W, BAND_ROWS), threads=args.threads)
        rows.append(rgba[y0:y0 + BAND_ROWS, :, :3].copy())
        for k in keys:
            total[k] += int(getattr(s, k))
        print(f"band at row {y0}: {time.time() - t0:.0f} s", flush=True)
    dt = time.time() - t0
    img = np.concatenate(rows, 0)
    out = os.path.join(ROOT, "tests", "golden", "render", "c4_lucy_2160p32_bands.npz")
    np.savez_compressed(out, r=img[..., 0], g=img[..., 1], b=img[..., 2], band_y0=np.array(y0s), band_rows=BAND_ROWS, width=W, height=H, spp=spp,
                        seed=1, flags=int(abi.RT_MODE_GPU_REF), focal=focal, aperture=aperture, oracle_seconds=dt,
                        **{"ctr_" + k: np.uint64(v) for k, v in total.items()})
    print(f"{out}: {os.path.getsize(out) / 1e6:.2f} MB, oracle {dt:.0f} s, {total}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--config", default="headline", choices=["headline", "c2", "c3", "c5x4", "c5", "c4bands", "c4"])
    args = ap.parse_args()
    if args.config == "c4bands":
        return bands(args)
    name, W, H, spp, aperture, focal = {"headline": ("c4_lucy", 1920, 1080, 32, 1.0, 1.35), "c2": ("c2_textured_mirror", 1280, 720, 128, 10.0, 10.2),
                                        "c3": ("c3_orange", 1920, 1080, 64, 10.0, 14.5),
                                        "c5x4": ("c5_final", 1920, 1080, 4, 150.0, 168.0),  # C5's frame at 4 of its 128 samples
                                        "c5": ("c5_final", 1920, 1080, 128, 150.0, 168.0),  # the whole of C5: about 75 minutes on 8 cores
                                        "c4": ("c4_lucy", 3840, 2160, 32, 1.0, 1.35)}[args.config]  # the whole of C4: about 70 minutes on 8 cores
    args.out = os.path.join(ROOT, "tests", "golden", "render", f"{name}_{H}p{spp}.npz")
    hs = HostScene(os.path.join(SCENE_DIR, name + ".txt"))
    st = orc.settings(spp=spp, dof=True, focal=focal, aperture=aperture, seed=1, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    t0 = time.time()
    rgba, _, s = orc.render(hs, st, W, H, threads=args.threads)
    dt = time.time() - t0
    counters = {k: int(getattr(s, k)) for k in ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits",
                                                "texture_fetches", "pixels")}
    # planes compress better than interleaved pixels
    np.savez_compressed(args.out, r=rgba[..., 0], g=rgba[..., 1], b=rgba[..., 2], width=W, height=H, spp=spp, seed=1,
                        flags=int(abi.RT_MODE_GPU_REF), focal=focal, aperture=aperture, oracle_seconds=dt,
                        **{"ctr_" + k: np.uint64(v) for k, v in counters.items()})
    print(f"{args.out}: {os.path.getsize(args.out) / 1e6:.2f} MB, oracle {dt:.0f} s, {counters}")


if __name__ == "__main__":
    main()
