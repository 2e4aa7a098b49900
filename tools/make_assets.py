#!/usr/bin/env python3
"""Generate the data files under assets/ (committed; regenerate only when this script changes).

* Procedural 24-bit BMP textures (checker / marble / stone look-alikes).  The reference's own
  texture files are not redistributed; the scenes under scenes/ use these instead.
* Packed meshes (*.rtm: "RTMESH1\\0", u32 n_vertices, u32 n_faces, vertices as f64 triples,
  faces as 0-based i32 triples) converted from the OBJ models the reference's scenes name
  (read from the reference tree when it is present).  Vertices are parsed with Python's
  float(), which rounds exactly like the C++ stream extraction the reference uses
  (CR/primitive.cpp:407-410), so the packed file yields the same doubles as the OBJ text.
* A 12-vertex icosahedron OBJ (own construction) for OBJ-parser and BVH tie-handling tests.
"""
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
REF_RES = "/root/reference/CUDA_RayTracing/resources"


def write_bmp(path, rgb):
    """rgb: uint8 [H, W, 3], row 0 = first row stored in the file."""
    h, w, _ = rgb.shape
    pad = (-(w * 3)) % 4
    rows = []
    for i in range(h):
        rows.append(rgb[i, :, ::-1].tobytes() + b"\0" * pad)
    data = b"".join(rows)
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54)
    hdr += struct.pack("<IIIHHIIIIII", 40, w, h, 1, 24, 0, len(data), 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(hdr + data)


def value_noise(rng, n, cells):
    g = rng.random((cells + 1, cells + 1))
    g[-1, :] = g[0, :]
    g[:, -1] = g[:, 0]
    x = np.linspace(0, cells, n, endpoint=False)
    xi = x.astype(int)
    xf = x - xi
    s = xf * xf * (3 - 2 * xf)
    a = g[xi][:, xi]
    b = g[xi + 1][:, xi]
    c = g[xi][:, xi + 1]
    d = g[xi + 1][:, xi + 1]
    sy, sx = s[:, None], s[None, :]
    return (a * (1 - sy) + b * sy) * (1 - sx) + (c * (1 - sy) + d * sy) * sx


def fbm(rng, n, octaves=5):
    out = np.zeros((n, n))
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        out += amp * value_noise(rng, n, 4 << o)
        tot += amp
        amp *= 0.5
    return out / tot


def textures():
    rng = np.random.default_rng(20240511)
    n = 256
    yy, xx = np.mgrid[0:n, 0:n]
    chk = (((yy // 32) + (xx // 32)) & 1).astype(float)
    f = fbm(rng, n)
    floor = np.stack([60 + 170 * chk + 20 * f, 50 + 150 * chk + 25 * f, 40 + 120 * chk + 30 * f], -1)
    write_bmp(os.path.join(ASSETS, "floor.bmp"), np.clip(floor, 0, 255).astype(np.uint8))
    f2 = fbm(rng, n)
    veins = 0.5 + 0.5 * np.sin((xx / n * 6 + 5 * f2) * math.pi)
    marble = np.stack([120 + 130 * veins, 115 + 125 * veins, 125 + 120 * veins], -1)
    write_bmp(os.path.join(ASSETS, "marble.bmp"), np.clip(marble, 0, 255).astype(np.uint8))
    f3 = fbm(rng, n, 6)
    # non-multiple-of-4 width exercises BMP row padding
    stone = np.stack([90 + 110 * f3, 85 + 100 * f3, 80 + 90 * f3], -1)[:, :250]
    write_bmp(os.path.join(ASSETS, "stone.bmp"), np.clip(stone, 0, 255).astype(np.uint8))


def convert_obj(src, dst):
    verts, faces = [], []
    with open(src, "r") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                faces.append((int(t[1]) - 1, int(t[2]) - 1, int(t[3]) - 1))
    v = np.asarray(verts, dtype="<f8")
    fa = np.asarray(faces, dtype="<i4")
    with open(dst, "wb") as f:
        f.write(b"RTMESH1\0" + struct.pack("<II", len(v), len(fa)))
        f.write(v.tobytes())
        f.write(fa.tobytes())
    print(f"{os.path.basename(dst)}: {len(v)} vertices, {len(fa)} faces")


def icosahedron():
    p = (1 + math.sqrt(5)) / 2
    v = []
    for a in (-1, 1):
        for b in (-p, p):
            v += [(0, a, b), (a, b, 0), (b, 0, a)]
    v = np.asarray(v, dtype=float)
    v /= np.linalg.norm(v[0])
    # faces = all triples of mutually adjacent vertices (edge length 2/|v|)
    edge = 2 / math.sqrt(1 + p * p)
    faces = []
    for i in range(12):
        for j in range(i + 1, 12):
            for k in range(j + 1, 12):
                if all(abs(np.linalg.norm(v[a] - v[b]) - edge) < 1e-9 for a, b in ((i, j), (j, k), (i, k))):
                    n = np.cross(v[j] - v[i], v[k] - v[i])
                    faces.append((i, j, k) if np.dot(n, v[i]) > 0 else (i, k, j))
    assert len(faces) == 20
    with open(os.path.join(ASSETS, "icosa.obj"), "w") as f:
        f.write("# unit icosahedron\n\n")
        for x in v:
            f.write("v %.6f %.6f %.6f\n" % tuple(x))
        f.write("\n")
        for a, b, c in faces:
            f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))


def main():
    os.makedirs(ASSETS, exist_ok=True)
    textures()
    icosahedron()
    if os.path.isdir(REF_RES):
        convert_obj(os.path.join(REF_RES, "lucy3_tris_normalized.obj"), os.path.join(ASSETS, "lucy_34k_normalized.rtm"))
        convert_obj(os.path.join(REF_RES, "lucy3_tris_smooth.obj"), os.path.join(ASSETS, "lucy_34k_smooth.rtm"))
        convert_obj(os.path.join(REF_RES, "stanford-bunny.obj"), os.path.join(ASSETS, "bunny_69k.rtm"))
    else:
        print("reference tree not present: packed meshes left as committed", file=sys.stderr)


if __name__ == "__main__":
    main()
