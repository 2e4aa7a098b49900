"""Pin the CPU oracle (oracle/cpu_ref.c) to the reference: in RT_MODE_CPU_REF it must reproduce
bit for bit what the reference's own CPU renderer computed (tests/golden/render/*.npz, produced
by tools/make_golden.py from Raytracer::RayTracing and Scene::FindNearestPrimitiveGetCollide,
CR/raytracer.cpp:136-178, CR/Scene.cpp:27-38)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_binding as orc
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.host import HostScene
from helpers import GOLD, SCENES, same_double

RENDERABLE = ["c1_purple_sphere", "t_cpu_features", "t_inside"]


@pytest.mark.parametrize("name", RENDERABLE)
def test_oracle_reproduces_reference_cpu_renderer(name):
    g = np.load(os.path.join(GOLD, "render", name + ".npz"))
    hs = HostScene(os.path.join(SCENES, name + ".txt"))
    W, H = hs.desc.camera.W, hs.desc.camera.H
    st = orc.settings(spp=1, flags=abi.RT_MODE_CPU_REF)
    hit, t, _, _, _ = orc.trace_primary(hs, st, W, H)
    rgba, rgb, stats = orc.render(hs, st, W, H)
    assert np.array_equal(hit, g["hit"].astype(np.int32)), "nearest-hit primitive differs from the reference"
    assert np.array_equal(rgba[..., :3], g["rgb8"]), "8-bit colour differs from the reference"
    step = int(g["lattice"]) if "lattice" in g else 1
    assert same_double(t[::step, ::step], g["dist"]), "hit distance differs from the reference"
    assert same_double(rgb[::step, ::step], g["rgb"]), "colour differs from the reference (must be bit-exact)"
    assert stats.rays_primary == W * H and stats.pixels == W * H


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors published with Random123 (kat_vectors)."""
    lib = orc.load()
    kats = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
            ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
            ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kats:
        c = (C.c_uint32 * 4)(*ctr)
        lib.orc_philox4x32_10(C.byref(c), key[0], key[1])
        assert tuple(c) == want


def test_frand_range_and_mapping():
    lib = orc.load()
    v = np.array([lib.orc_frand_at(7, 123, 4, d) for d in range(4096)], dtype=np.float32)
    assert v.min() > 0.0 and v.max() <= 1.0
    assert abs(float(v.mean()) - 0.5) < 0.03
    # draw d of (pixel, sample) is word d%4 of Philox block d/4 with counter (pixel, sample, d/4, 0)
    c = (C.c_uint32 * 4)(123, 4, 5, 0)
    lib.orc_philox4x32_10(C.byref(c), 7, 0)
    want = np.float32(c[2]) * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)
    assert lib.orc_frand_at(7, 123, 4, 22) == want


def test_oracle_image_independent_of_threads_and_tiles():
    hs = HostScene(os.path.join(SCENES, "t_icosa.txt"))
    st = orc.settings(spp=2, dof=True, focal=9.0, aperture=1.0, seed=5, flags=abi.RT_MODE_GPU_REF)
    full, _, s_full = orc.render(hs, st, 48, 36, threads=0)
    one, _, s_one = orc.render(hs, st, 48, 36, threads=1)
    assert np.array_equal(full, one) and s_full.rays == s_one.rays
    img = np.zeros_like(full)
    total = 0
    for (x0, y0, w, h) in [(0, 0, 20, 36), (20, 0, 28, 10), (20, 10, 28, 26)]:
        part, _, s = orc.render(hs, st, 48, 36, rect=(x0, y0, w, h))
        img[y0:y0 + h, x0:x0 + w] = part[y0:y0 + h, x0:x0 + w]
        total += s.rays
    assert np.array_equal(img, full) and total == s_full.rays


# ---- Bezier surface of revolution (no reference counterpart: parity unpinned, analytic checks only) -----

BEZ_HEAD = """background
 color= 0 0 0
end
camera
 O= 0.3 -10 1.2
 N= 0 1 -0.02
 image_W= 64
 image_H= 48
end
light point
 O= 0 -10 10
 color= 1 1 1
end
"""


def _scene_from_text(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return HostScene(str(p))


def test_bezier_newton_straight_profile_is_a_cylinder_body(tmp_path):
    """A constant-radius profile swept around the axis is an open cylinder: wherever the oracle's cylinder
    reports a body hit, the Newton intersector must report the same distance."""
    bez = _scene_from_text(tmp_path, "b.txt", BEZ_HEAD + "primitive bezier\n O1= 0 0 0\n O2= 0 0 2\n P= 0 0.5\n P= 0.7 0.5\n P= 2 0.5\n diff= 1\nend\n")
    cyl = _scene_from_text(tmp_path, "c.txt", BEZ_HEAD + "primitive cylinder\n O1= 0 0 0\n O2= 0 0 2\n R= 0.5\n diff= 1\nend\n")
    st = orc.settings(flags=abi.RT_F_NORMALIZE_PRIMARY)
    hb, tb, _, _, _ = orc.trace_primary(bez, st, 64, 48)
    hc, tc, _, _, _ = orc.trace_primary(cyl, st, 64, 48)
    assert (hc >= 0).sum() > 50
    assert np.array_equal(hb >= 0, hc >= 0)   # the camera looks from the side: no cap is visible
    m = hc >= 0
    assert np.abs(tb[m] - tc[m]).max() < 1e-10


def test_bezier_newton_cone_profile_matches_the_analytic_cone(tmp_path):
    """A linear profile r = 1 - z/4 (0 <= z <= 2) is a truncated cone; compare with the closed form."""
    hs = _scene_from_text(tmp_path, "k.txt", BEZ_HEAD + "primitive bezier\n O1= 0 0 0\n O2= 0 0 2\n P= 0 1\n P= 2 0.5\n diff= 1\nend\n")
    st = orc.settings(flags=abi.RT_F_NORMALIZE_PRIMARY)
    W, H = 64, 48
    hit, t, _, _, _ = orc.trace_primary(hs, st, W, H)
    cam = hs.desc.camera
    O = np.array([cam.O.x, cam.O.y, cam.O.z])
    N = np.array([cam.N.x, cam.N.y, cam.N.z])
    Dx = np.array([cam.Dx.x, cam.Dx.y, cam.Dx.z])
    Dy = np.array([cam.Dy.x, cam.Dy.y, cam.Dy.z])
    n_checked = 0
    for y in range(H):
        for x in range(W):
            # primary ray of pixel (x, y): CR/camera.cpp:57-59
            V = N + Dy * (2 * y / H - 1) + Dx * (2 * x / W - 1)
            V /= np.linalg.norm(V)
            # |(O + tV)_xy|^2 = (1 - (Oz + t Vz)/4)^2
            k = 1 - O[2] / 4
            a = V[0] ** 2 + V[1] ** 2 - V[2] ** 2 / 16
            b = 2 * (O[0] * V[0] + O[1] * V[1]) + 2 * k * V[2] / 4
            c = O[0] ** 2 + O[1] ** 2 - k * k
            disc = b * b - 4 * a * c
            want = None
            if disc >= 0:
                for r in sorted([(-b - disc ** 0.5) / (2 * a), (-b + disc ** 0.5) / (2 * a)]):
                    z = O[2] + r * V[2]
                    if r > 1e-6 and 0 <= z <= 2:
                        want = r
                        break
            if want is None:
                continue
            zc = O[2] + want * V[2]
            if min(zc, 2 - zc) < 1e-3:
                continue  # on the rim, either answer is legitimate
            assert hit[y, x] >= 0, (x, y)
            assert abs(t[y, x] - want) < 1e-9, (x, y, t[y, x], want)
            n_checked += 1
    assert n_checked > 100


def test_bezier_power_basis_equals_bernstein_form():
    from math import comb
    rng = np.random.default_rng(5)
    for n in range(1, 10):
        P = rng.uniform(-2, 3, n + 1)
        c = np.zeros(n + 1)
        orc.load().orc_bezier_power_basis(C.c_void_p(P.ctypes.data), n, C.c_void_p(c.ctypes.data))
        for u in np.linspace(0, 1, 11):
            bern = sum(comb(n, i) * u ** i * (1 - u) ** (n - i) * P[i] for i in range(n + 1))
            assert abs(np.polyval(c[::-1], u) - bern) < 1e-9 * (1 + abs(bern)) * 4 ** n


# ---- bilinear fetch: the CPU renderer's weights and the .cu files' weights, by hand ---------------------

def _fetch(tex, flags, u, v):
    tex = np.ascontiguousarray(tex, dtype=np.uint8)
    t = abi.Texture(width=tex.shape[1], height=tex.shape[0], rgb=tex.ctypes.data_as(C.POINTER(C.c_uint8)))
    out = (C.c_double * 3)()
    orc.load().orc_smooth_color(C.byref(t), C.c_uint32(flags), C.c_double(u), C.c_double(v), out)
    return list(out)


def test_bilinear_taps_of_both_renderers_on_one_texel_quad():
    """One fetch from an asymmetric 2x2 texture (rows = U, columns = V), expected values worked out by hand from the
    reference's two fetch routines, not from the oracle.

    u = 0.375, v = 0.125 on a 2x2 texture: U = 0.75, V = 0.25, U1 = V1 = 0, U2 = V2 = 1, rat_U = 0.25, rat_V = 0.75.
    Channel 0 of the texels: ima[0][0] = 40, ima[0][1] = 80, ima[1][0] = 120, ima[1][1] = 200.

    CR/bmp.cpp:146-149 (double taps; the / 256 of GetColor folded into the end):
        ima[U1][V1] = ima[0][0] =  40 * .25 * .75 =  7.5
        ima[U1][V2] = ima[0][1] =  80 * .25 * .25 =  5
        ima[U2][V1] = ima[1][0] = 120 * .75 * .75 = 67.5
        ima[U2][V2] = ima[1][1] = 200 * .75 * .25 = 37.5         -> 117.5 / 256

    CR/Primitive.cu:108-111 on the transposed copy texture[col * H + row] = ima[row][col] (CR/Renderer_CUDA.cpp:149-152),
    every product cut to 8 bits (helper_math.h:1842-1844):
        texture[V1 * H + U1] = ima[0][0]: uchar(uchar( 40 * .25) * .75) = uchar( 10 * .75) =  7
        texture[V1 * H + U2] = ima[1][0]: uchar(uchar(120 * .25) * .25) = uchar( 30 * .25) =  7
        texture[V2 * H + U1] = ima[0][1]: uchar(uchar( 80 * .75) * .75) = uchar( 60 * .75) = 45
        texture[V2 * H + U2] = ima[1][1]: uchar(uchar(200 * .75) * .25) = uchar(150 * .25) = 37   -> 96 / 256
    (the CPU renderer's weights on 8-bit taps would give 7 + 5 + 67 + 37 = 116: the order of the weights matters.)"""
    tex = np.zeros((2, 2, 3), dtype=np.uint8)
    tex[0, 0] = (40, 41, 0)
    tex[0, 1] = (80, 83, 255)
    tex[1, 0] = (120, 127, 255)
    tex[1, 1] = (200, 211, 0)
    u, v = 0.375, 0.125
    cpu = _fetch(tex, abi.RT_MODE_CPU_REF, u, v)
    assert cpu[0] == 117.5 / 256
    assert cpu[1] == (41 * .25 * .75 + 83 * .25 * .25 + 127 * .75 * .75 + 211 * .75 * .25) / 256
    assert cpu[2] == (255 * .25 * .25 + 255 * .75 * .75) / 256
    gpu = _fetch(tex, abi.RT_F_TEX_U8_TAPS, u, v)
    assert gpu[0] == 96 / 256
    # channel 1: uchar(uchar(10.25) * .75) = 7, uchar(uchar(31.75) * .25) = 7, uchar(uchar(62.25) * .75) = 46, uchar(uchar(158.25) * .25) = 39
    assert gpu[1] == (7 + 7 + 46 + 39) / 256
    # channel 2: 0, uchar(uchar(63.75) * .25) = 15, uchar(uchar(191.25) * .75) = 143, 0
    assert gpu[2] == (15 + 143) / 256
    assert gpu == _fetch(tex, abi.RT_MODE_GPU_REF, u, v)


def test_audit_cites_the_oracle_where_it_is():
    """oracle/AUDIT.md maps every statement group of the reference's .cu files to line ranges of cpu_ref.c: they must still
    lie inside the functions they name."""
    import subprocess
    import sys
    from helpers import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_audit_lines.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
