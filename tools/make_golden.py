#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's own CPU code.

Needs the reference tree (/root/reference) and is therefore run in the build container only;
its outputs are committed.  It builds oracle/_ref/ref_dump (the reference's CPU sources behind
oracle/ref_dump.cpp), runs it, and stores what the reference computed:

  parse/<scene>.json      parsed scene (primitive list order, every field, camera basis, lights)
                          + per-mesh BVH statistics and a SHA-256 of the canonical tree dump
  parse/ref_<scene>.json  the same for every scene file the reference ships
  bvh/<scene>.npz         full canonical tree for small meshes
  render/<scene>.npz      per-pixel nearest-hit primitive index, hit distance and pass-1 colour
                          of Raytracer::RayTracing for scenes its CPU renderer supports
  texture/<name>.npz      Bmp::GetSmoothColor samples

Only values are stored (fixtures); no reference source text is.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "assets")
SCENES = os.path.join(ROOT, "scenes")
REF_RES = "/root/reference/CUDA_RayTracing/resources"
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")

# mesh files of scenes/ -> the file the reference can read for the same model
MESH_SOURCE = {
    "lucy_34k_normalized.rtm": os.path.join(REF_RES, "lucy3_tris_normalized.obj"),
    "lucy_34k_smooth.rtm": os.path.join(REF_RES, "lucy3_tris_smooth.obj"),
    "bunny_69k.rtm": os.path.join(REF_RES, "stanford-bunny.obj"),
    "icosa.obj": os.path.join(ASSETS, "icosa.obj"),
}
# scenes the reference's CPU renderer can render deterministically (sphere/plane/square,
# point lights, drefl = 0): these get per-pixel fixtures
RENDERABLE = ["c1_purple_sphere", "t_cpu_features", "t_inside"]
FULL_BVH_MAX_NODES = 4096


def run(args, cwd):
    subprocess.run(args, cwd=cwd, check=True, stdout=subprocess.DEVNULL)


def scene_for_reference(path, tmpdir):
    """Copy of a scenes/ file whose mesh `file=` entries name what the reference can parse."""
    out = []
    for line in open(path):
        t = line.split()
        if len(t) == 2 and t[0] == "file=" and t[1] in MESH_SOURCE:
            line = line.replace(t[1], MESH_SOURCE[t[1]])
        out.append(line)
    dst = os.path.join(tmpdir, os.path.basename(path))
    with open(dst, "w") as f:
        f.writelines(out)
    return dst


def parse_scene(scene_path, cwd, name, tmpdir, out_name):
    js = os.path.join(tmpdir, name + ".json")
    bvh = os.path.join(tmpdir, name + ".bvh")
    run([REF_DUMP, "parse", scene_path, js, bvh], cwd)
    d = json.load(open(js))
    raw = np.fromfile(bvh, dtype="<f8").reshape(-1, 16)
    off = 0
    small = {}
    mi = 0
    for p in d["primitives"]:
        if p["type"] != 6:
            continue
        n = p["bvh_nodes"]
        tree = raw[off:off + n]
        off += n
        p["bvh_sha256"] = hashlib.sha256(np.ascontiguousarray(tree).tobytes()).hexdigest()
        if 0 < n <= FULL_BVH_MAX_NODES:
            small[f"mesh{mi}"] = tree
        mi += 1
    assert off == len(raw)
    with open(os.path.join(GOLD, "parse", out_name + ".json"), "w") as f:
        json.dump(d, f, indent=0, sort_keys=True)
    if small:
        np.savez_compressed(os.path.join(GOLD, "bvh", out_name + ".npz"), **small)
    return d


def render_scene(scene_path, cwd, name, tmpdir):
    out = os.path.join(tmpdir, name + ".render")
    run([REF_DUMP, "render", scene_path, out], cwd)
    with open(out, "rb") as f:
        w, h = np.frombuffer(f.read(8), dtype="<i4")
        hit = np.frombuffer(f.read(4 * w * h), dtype="<i4").reshape(h, w)
        dist = np.frombuffer(f.read(8 * w * h), dtype="<f8").reshape(h, w)
        rgb = np.frombuffer(f.read(24 * w * h), dtype="<f8").reshape(h, w, 3)
    rgb8 = (rgb * 255).astype(np.int64).astype(np.uint8)  # Bmp::SetColor truncation (CR/bmp.cpp:127-131)
    data = {"hit": hit.astype(np.int8), "rgb8": rgb8}
    if w * h <= 8192:
        data["dist"] = dist
        data["rgb"] = rgb
    else:  # large frames: full-precision values on a 8-pixel lattice only
        data["lattice"] = np.int32(8)
        data["dist"] = dist[::8, ::8].copy()
        data["rgb"] = rgb[::8, ::8].copy()
    np.savez_compressed(os.path.join(GOLD, "render", name + ".npz"), **data)


def texture_samples(bmp, name, tmpdir):
    out = os.path.join(tmpdir, name + ".tex")
    run([REF_DUMP, "texture", bmp, "24", out], ASSETS)
    with open(out, "rb") as f:
        w, h, n = np.frombuffer(f.read(12), dtype="<i4")
        rec = np.frombuffer(f.read(), dtype="<f8").reshape(n * n, 5)
    np.savez_compressed(os.path.join(GOLD, "texture", name + ".npz"), width=w, height=h, uv=rec[:, :2], rgb=rec[:, 2:])


def main():
    if not os.path.isdir(REF_RES):
        sys.exit("reference tree not present; fixtures are generated in the build container only")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    for sub in ("parse", "bvh", "render", "texture"):
        os.makedirs(os.path.join(GOLD, sub), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for fn in sorted(os.listdir(SCENES)):
            if not fn.endswith(".txt"):
                continue
            name = fn[:-4]
            src = scene_for_reference(os.path.join(SCENES, fn), tmp)
            parse_scene(src, ASSETS, name, tmp, name)
            if name in RENDERABLE:
                render_scene(src, ASSETS, name, tmp)
            print("scene", name)
        for fn in sorted(os.listdir(REF_RES)):
            if fn.startswith("scene") and fn.endswith(".txt"):
                parse_scene(os.path.join(REF_RES, fn), REF_RES, "ref_" + fn[:-4], tmp, "ref_" + fn[:-4])
                print("reference scene", fn)
        for bmp in ("floor.bmp", "marble.bmp", "stone.bmp"):
            texture_samples(os.path.join(ASSETS, bmp), bmp[:-4], tmp)


if __name__ == "__main__":
    main()
