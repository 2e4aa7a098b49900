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
