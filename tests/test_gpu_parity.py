"""GPU parity: the HIP render path (through the C ABI) against the CPU oracle and the committed
golden fixtures from the reference's own CPU renderer.

Bars: primary-ray hit primitive, hit leaf, hit distance and BVH visit order bit-exact; ray / node /
triangle counts exact in strict traversal; colour within 1/255 per channel (the only inexact
operations on the colour path are libm calls - pow, exp, sin, cos, acos, atan2 - whose device and
host implementations may differ in the last ulp), with the number of pixels that differ at all
bounded."""
import os

import numpy as np
import pytest

import oracle_binding as orc
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.host import HostScene
from helpers import GOLD, SCENES, same_double

pytestmark = pytest.mark.gpu

COLOUR_TOL = 1          # max abs error, in 8-bit levels
MAX_DIFF_FRACTION = 0.002  # pixels allowed to differ by that one level
MIN_PSNR_DB = 70.0      # peak signal-to-noise ratio of the 8-bit image against the oracle's (identical images: infinite)


def scene(name):
    return HostScene(os.path.join(SCENES, name + ".txt"))


def device_scene(hs):
    from cuda_raytracing_amd.device import DeviceScene
    return DeviceScene(hs, device=0)


def assert_colour_close(img, ref, what=""):
    d = np.abs(img[..., :3].astype(np.int32) - ref[..., :3].astype(np.int32))
    assert d.max() <= COLOUR_TOL, f"{what}: max abs colour error {d.max()} levels at {np.unravel_index(d.argmax(), d.shape)}"
    frac = float((d.max(-1) > 0).mean())
    assert frac <= MAX_DIFF_FRACTION, f"{what}: {frac:.4%} of pixels differ"
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr >= MIN_PSNR_DB, f"{what}: PSNR {psnr:.1f} dB"
    assert np.array_equal(img[..., 3], ref[..., 3])


@pytest.mark.parametrize("name", ["c1_purple_sphere", "t_cpu_features", "t_inside"])
def test_cpu_ref_mode_against_reference_fixtures(name):
    """Config 1 family: what the reference's CPU renderer computed (golden) vs the GPU."""
    g = np.load(os.path.join(GOLD, "render", name + ".npz"))
    hs = scene(name)
    W, H = hs.desc.camera.W, hs.desc.camera.H
    st = orc.settings(spp=1, flags=abi.RT_MODE_CPU_REF)
    with device_scene(hs) as ds:
        hit, t, _, _, _ = ds.trace_primary(st, W, H)
        img, stats = ds.render_frame(st, W, H)
    assert np.array_equal(hit, g["hit"].astype(np.int32)), "primary hit ids differ from the reference"
    step = int(g["lattice"]) if "lattice" in g else 1
    assert same_double(t[::step, ::step], g["dist"]), "primary hit distances differ from the reference (must be bit-exact)"
    ref = np.concatenate([g["rgb8"], np.full(g["rgb8"].shape[:2] + (1,), 255, np.uint8)], -1)
    assert_colour_close(img, ref, name)
    assert stats.rays_primary == W * H


@pytest.mark.parametrize("name,flags,spp,dof", [
    ("t_icosa", abi.RT_MODE_GPU_REF, 4, True),
    ("t_icosa", abi.RT_MODE_CPU_REF, 2, False),
    ("t_cpu_features", abi.RT_MODE_GPU_REF, 3, True),
    ("t_grammar_quirks", abi.RT_MODE_GPU_REF, 1, False),
    ("t_bezier", abi.RT_MODE_GPU_REF, 2, True),                                  # stacked-cylinder Bezier
    ("t_bezier", abi.RT_MODE_GPU_REF & ~abi.RT_F_BEZIER_STACKED, 2, True),       # true surface of revolution
    ("c5_final", abi.RT_MODE_GPU_REF & ~abi.RT_F_BEZIER_STACKED, 1, False),
    ("c2_textured_mirror", abi.RT_MODE_GPU_REF, 8, True),                        # config 2: square light, textured floor, mirror
    ("c3_orange", abi.RT_MODE_GPU_REF, 2, True),                                 # config 3: glass / mirror balls, cylinders, panels
    ("c3_orange", abi.RT_MODE_CPU_REF, 1, False),
    ("t_boxgrid", abi.RT_MODE_GPU_REF, 1, False),   # axis-aligned mesh, rays with a zero direction component, hits on shared edges
    ("t_boxgrid", abi.RT_MODE_GPU_REF, 3, True),
    ("t_boxgrid", abi.RT_MODE_CPU_REF, 1, False),
])
def test_full_render_against_oracle(name, flags, spp, dof):
    hs = scene(name)
    W, H = 96, 72
    st = orc.settings(spp=spp, dof=dof, focal=9.0, aperture=1.5, seed=11, flags=flags | abi.RT_F_STRICT_TRAVERSAL)
    ref, _, ostats = orc.render(hs, st, W, H)
    with device_scene(hs) as ds:
        img, stats = ds.render_frame(st, W, H)
        st.flags = flags  # pruned traversal: same image, same rays
        fast, fstats = ds.render_frame(st, W, H)
    assert_colour_close(img, ref, name)
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "shaded_hits", "texture_fetches", "nodes_visited", "tris_tested", "pixels"):
        assert getattr(stats, k) == getattr(ostats, k), (k, getattr(stats, k), getattr(ostats, k))
    assert np.array_equal(fast, img), "pruned traversal changed the image"
    assert fstats.rays == stats.rays and fstats.nodes_visited <= stats.nodes_visited


@pytest.mark.parametrize("name,rect", [("t_icosa", (0, 0, 96, 72)), ("c4_lucy", (0, 0, 160, 96)), ("c5_final", (0, 0, 160, 96)),
                                       ("t_boxgrid", (0, 0, 160, 96))])
def test_primary_hits_and_visit_order_bit_exact(name, rect):
    """Hit ids and BVH visit order of primary rays, reference-order traversal."""
    hs = scene(name)
    W, H = 160, 96
    cap = 1024
    st = orc.settings(flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    o_hit, o_t, o_leaf, o_log, o_cnt = orc.trace_primary(hs, st, W, H, rect=rect, visit_cap=cap)
    with device_scene(hs) as ds:
        hit, t, leaf, log, cnt = ds.trace_primary(st, W, H, rect=rect, visit_cap=cap)
    assert np.array_equal(hit, o_hit) and np.array_equal(leaf, o_leaf)
    assert same_double(t, o_t)
    assert np.array_equal(cnt, o_cnt), "number of BVH items visited differs"
    assert cnt.max() > 0
    k = np.minimum(cnt, cap)
    valid = np.arange(cap)[None, None, :] < k[..., None]
    assert np.array_equal(np.where(valid, log, 0), np.where(valid, o_log, 0)), "BVH visit order differs"


@pytest.mark.parametrize("name,W,H,spp", [("c4_lucy", 160, 90, 2), ("c5_final", 128, 96, 1)])
def test_mesh_scenes_against_oracle_and_pruned_equals_strict(name, W, H, spp):
    hs = scene(name)
    base = abi.RT_MODE_GPU_REF
    st = orc.settings(spp=spp, dof=True, focal=1.35 if name == "c4_lucy" else 168.0, aperture=1.0 if name == "c4_lucy" else 150.0,
                      seed=2, flags=base | abi.RT_F_STRICT_TRAVERSAL)
    ref, _, ostats = orc.render(hs, st, W, H)
    with device_scene(hs) as ds:
        strict, sstats = ds.render_frame(st, W, H)
        st.flags = base
        fast, fstats = ds.render_frame(st, W, H)
    assert_colour_close(strict, ref, name)
    assert sstats.rays == ostats.rays and sstats.nodes_visited == ostats.nodes_visited and sstats.tris_tested == ostats.tris_tested
    assert np.array_equal(fast, strict), "pruned traversal changed the image"
    assert fstats.rays == sstats.rays
    assert fstats.nodes_visited < sstats.nodes_visited


@pytest.mark.parametrize("stacked", [True, False])
def test_bezier_primary_hits_bit_exact(stacked):
    """Both Bezier intersectors use only + - * / sqrt: hit ids and distances must equal the oracle's bit for bit."""
    hs = scene("t_bezier")
    W, H = 192, 144
    flags = abi.RT_MODE_GPU_REF if stacked else abi.RT_MODE_GPU_REF & ~abi.RT_F_BEZIER_STACKED
    st = orc.settings(flags=flags)
    o_hit, o_t, _, _, _ = orc.trace_primary(hs, st, W, H)
    with device_scene(hs) as ds:
        hit, t, _, _, _ = ds.trace_primary(st, W, H)
    bez = [i for i in range(hs.desc.n_primitives) if hs.desc.primitives[i].type == abi.RT_PRIM_BEZIER]
    assert all((o_hit == i).sum() > 100 for i in bez), "scene does not show both Bezier primitives"
    assert np.array_equal(hit, o_hit)
    assert same_double(t, o_t)


@pytest.mark.parametrize("spp", [1, 3, 32, 64, 100, 1500])
def test_sample_counts(spp):
    """1 lane per sample: groups that do not divide 64 and counts above 64 (several passes)."""
    hs = scene("t_icosa")
    W, H = (40, 24) if spp <= 100 else (9, 5)
    st = orc.settings(spp=spp, dof=True, focal=9.0, aperture=2.0, seed=spp, flags=abi.RT_MODE_GPU_REF)
    ref, _, ostats = orc.render(hs, st, W, H)
    with device_scene(hs) as ds:
        img, stats = ds.render_frame(st, W, H)
    assert_colour_close(img, ref, f"spp={spp}")
    assert stats.rays == ostats.rays


def test_call_cut_into_several_launches(monkeypatch):
    """A call with more paths than one launch may hold runs as several launches (here: at most 2 blocks each)."""
    hs = scene("t_icosa")
    W, H = 70, 45
    st = orc.settings(spp=5, dof=True, focal=9.0, aperture=1.0, seed=11, flags=abi.RT_MODE_GPU_REF)
    with device_scene(hs) as ds:
        whole, ws = ds.render_frame(st, W, H)
        monkeypatch.setenv("RT_AMD_LAUNCH_PATHS", str(2 * 64 * 5))
        cut, cs = ds.render_frame(st, W, H, tile=16)
        monkeypatch.setenv("RT_AMD_LAUNCH_PATHS", "64")  # less than one block: one block per launch
        rect = (3, 2, 30, 21)
        canvas = np.zeros_like(whole)
        ds.render_tile(st, W, H, rect, out=canvas)
    assert np.array_equal(cut, whole) and cs.rays == ws.rays
    x0, y0, w, h = rect
    assert np.array_equal(canvas[y0:y0 + h, x0:x0 + w], whole[y0:y0 + h, x0:x0 + w])


@pytest.mark.parametrize("builder", ["device", "device_sah"])
@pytest.mark.parametrize("name,W,H,spp", [("t_icosa", 96, 72, 3), ("t_boxgrid", 96, 72, 2), ("c4_lucy", 160, 90, 2), ("c5_final", 128, 96, 1)])
def test_trees_built_on_the_device_give_the_same_image(name, W, H, spp, builder):
    """The pruned walk keeps only hits the reference's own walk reaches, so any valid tree renders the same bits.  The trees
    themselves are read back and checked: every triangle once, depth within the walk's stack, boxes around their triangles;
    the surface-area builder's trees must be as good as the host's by the surface-area measure."""
    from cuda_raytracing_amd.device import DeviceScene
    hs = scene(name)
    st = orc.settings(spp=spp, dof=True, focal=9.0, aperture=1.0, seed=21, flags=abi.RT_MODE_GPU_REF)
    try:
        with DeviceScene(hs, device=0, tree_builder="host") as ds:
            img_h, st_h = ds.render_frame(st, W, H)
            hit_h = ds.trace_primary(st, W, H, (0, 0, W, H))[:3]
            assert ds.tree_build_ms == 0.0
            trees_h, cost_h = ds.check_trees()
        with DeviceScene(hs, device=0, tree_builder=builder) as ds:
            img_d, st_d = ds.render_frame(st, W, H)
            hit_d = ds.trace_primary(st, W, H, (0, 0, W, H))[:3]
            assert ds.tree_build_ms > 0.0
            trees_d, cost_d = ds.check_trees()
    finally:
        abi.load_library().rt_set_tree_builder(0)
    assert trees_d == trees_h == hs.desc.n_meshes and cost_h > 0
    if builder == "device_sah":
        assert cost_d <= 1.05 * cost_h, (cost_d, cost_h)
    assert np.array_equal(img_h, img_d)
    assert st_h.rays == st_d.rays and st_h.shaded_hits == st_d.shaded_hits
    for a, b in zip(hit_h, hit_d):
        assert np.array_equal(a, b)


def test_shadow_walk_variants_give_the_same_image(monkeypatch):
    """Plain and wave-shared shadow mesh walks, and the library's own choice between them, render the same bits."""
    hs = scene("c5_final")
    W, H = 400, 264
    st = orc.settings(spp=2, dof=True, focal=168.0, aperture=150.0, seed=5, flags=abi.RT_MODE_GPU_REF)
    images, rays = [], []
    for choice in ("0", "1", None):
        if choice is None:
            monkeypatch.delenv("RT_AMD_SHARED_WALK", raising=False)
        else:
            monkeypatch.setenv("RT_AMD_SHARED_WALK", choice)
        with device_scene(hs) as ds:
            img, s = ds.render_frame(st, W, H)
            picked, _, _ = ds.shared_walk
            assert picked == (1 if choice is None else int(choice))  # a scene with meshes gets the shared variant
        images.append(img)
        rays.append((s.rays, s.rays_shadow_traced))
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[0], images[2])
    # (the rays a render has to trace differ between the builds: the shared variant of this scene is the four-wave build,
    # which drops points no mesh can shadow before dealing their rays)
    assert rays[0][0] == rays[1][0] == rays[2][0] and rays[1] == rays[2]
    assert 0 < rays[1][1] <= rays[0][1] < s.rays_shadow


@pytest.mark.parametrize("name,W,H,focal,aperture", [("c4_lucy", 320, 180, 1.35, 1.0), ("c5_final", 256, 144, 168.0, 150.0)])
@pytest.mark.parametrize("preset", ["gpu_ref", "cpu_ref"])
def test_builds_for_three_and_four_waves_give_the_same_image(monkeypatch, name, W, H, focal, aperture, preset):
    """The kernel build for mesh-heavy scenes (four waves per SIMD) looks down the meshes' trees before it deals shadow rays
    and drops the points no mesh can shadow; the other build does not.  Both must render the reference-order image bit for
    bit, under both presets (they differ in how far a mesh's shadow reaches: RT_F_MESH_RAW_DIR), the first with fewer rays."""
    hs = scene(name)
    flags = abi.RT_MODE_GPU_REF if preset == "gpu_ref" else abi.RT_MODE_CPU_REF
    st_strict = orc.settings(spp=3, dof=True, focal=focal, aperture=aperture, seed=11, flags=flags | abi.RT_F_STRICT_TRAVERSAL)
    st_fast = orc.settings(spp=3, dof=True, focal=focal, aperture=aperture, seed=11, flags=flags)
    traced = {}
    strict = None
    for waves in ("3", "4"):
        monkeypatch.setenv("RT_AMD_WAVES", waves)
        with device_scene(hs) as ds:
            if strict is None:
                strict, sstats = ds.render_frame(st_strict, W, H)
            fast, fstats = ds.render_frame(st_fast, W, H)
        assert np.array_equal(fast, strict), f"{name} {preset}: the {waves}-wave build differs from the reference-order image"
        assert fstats.rays == sstats.rays and fstats.shaded_hits == sstats.shaded_hits
        traced[waves] = fstats.rays_shadow_traced
    assert 0 < traced["4"] <= traced["3"] < sstats.rays_shadow, traced
    if preset == "gpu_ref":  # (sampled lamps: the look down the trees has points to drop)
        assert traced["4"] < traced["3"], traced


def test_many_overlapping_meshes_shared_plain_and_strict_agree(tmp_path, monkeypatch):
    """Seven meshes whose boxes overlap between a floor and two lamps: a shadow ray has more candidate meshes than roots fit
    on the shared walk's stack, so some are walked by the lane on its own."""
    rng = np.random.default_rng(3)
    text = [EDGE_HEAD.replace("image_W= 64", "image_W= 96").replace("image_H= 48", "image_H= 72"),
            "light sphere\n O= 1 -2 6\n R= 0.5\n color= 1 1 1\nend\n", "light point\n O= -3 -3 5\n color= 0.5 0.5 0.4\nend\n",
            "primitive plane\n N= 0 0 1\n R= 0\n color= 1 1 1\n diff= 0.9\nend\n"]
    for k in range(7):
        n = 40
        verts = rng.normal(size=(n, 3)) * 0.6
        faces = [(1 + rng.integers(0, n), 1 + rng.integers(0, n), 1 + rng.integers(0, n)) for _ in range(60)]
        faces = [f for f in faces if len(set(f)) == 3]
        (tmp_path / f"m{k}.obj").write_text("".join(f"v {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}\n" for v in verts) + "".join(f"f {a} {b} {c}\n" for a, b, c in faces))
        text.append(f"primitive mesh\n O= {0.25 * k - 0.8:.2f} {0.1 * k:.2f} {1.2 + 0.15 * k:.2f}\n scale= 1 1 1\n rotation= 0.{k} 0.2 0\n file= m{k}.obj\n"
                    f" color= 0.{3 + k} 0.6 0.{9 - k}\n diff= 0.8\n refl= 0.1\nend\n")
    (tmp_path / "s.txt").write_text("".join(text))
    hs = HostScene(str(tmp_path / "s.txt"), asset_dir=str(tmp_path))
    W, H = 96, 72
    st = orc.settings(spp=2, dof=True, focal=6.0, aperture=1.0, seed=9, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    ref, _, ostats = orc.render(hs, st, W, H)
    images = []
    for choice in ("0", "1"):
        monkeypatch.setenv("RT_AMD_SHARED_WALK", choice)
        with device_scene(hs) as ds:
            if choice == "0":
                strict, sstats = ds.render_frame(st, W, H)
            fast, fstats = ds.render_frame(orc.settings(spp=2, dof=True, focal=6.0, aperture=1.0, seed=9, flags=abi.RT_MODE_GPU_REF), W, H)
            assert ds.shared_walk[0] == int(choice)
        images.append(fast)
    assert_colour_close(strict, ref, "seven meshes")
    assert sstats.rays == ostats.rays and sstats.nodes_visited == ostats.nodes_visited
    assert np.array_equal(images[0], strict) and np.array_equal(images[1], strict)
    assert fstats.rays == sstats.rays


@pytest.mark.parametrize("name,W,H,spp,flags", [
    ("c4_lucy", 320, 180, 4, abi.RT_MODE_GPU_REF), ("c5_final", 256, 144, 3, abi.RT_MODE_GPU_REF), ("c3_orange", 240, 136, 4, abi.RT_MODE_GPU_REF),
    ("t_icosa", 96, 72, 5, abi.RT_MODE_CPU_REF), ("t_bezier", 96, 72, 2, abi.RT_MODE_GPU_REF & ~abi.RT_F_BEZIER_STACKED),
])
def test_three_kernel_render_equals_one_kernel_render(monkeypatch, name, W, H, spp, flags):
    """RT_AMD_DEFER=1: paths without shadow rays -> all shadow rays -> colours from the records.  Same image, same ray counts;
    with room for too few records the call falls back to the one kernel and still gives the same image.  Only libraries built
    with -DRT_EXPERIMENTS have this render (tools/build_experiments.sh; run the suite on one with RT_AMD_LIB=build_ab/exp.so)."""
    if not abi.has_experiments():
        pytest.skip("the loaded library was built without -DRT_EXPERIMENTS")
    hs = scene(name)
    st = orc.settings(spp=spp, dof=True, focal=9.0, aperture=1.0, seed=13, flags=flags)
    monkeypatch.delenv("RT_AMD_DEFER", raising=False)
    with device_scene(hs) as ds:
        assert ds.deferred_shadows == (0, 0)
        one, os_ = ds.render_frame(st, W, H)
        cone_walk = ds.waves_per_simd == 4  # the kernel for mesh-heavy scenes looks down the trees before it deals shadow rays: fewer are traced
    monkeypatch.setenv("RT_AMD_DEFER", "1")
    with device_scene(hs) as ds:
        three, ts = ds.render_frame(st, W, H)
        rect = (W // 4 + 3, H // 4 + 1, W // 2, H // 2)
        part = np.zeros_like(three)
        ds.render_tile(st, W, H, rect, out=part)
        assert ds.deferred_shadows == (1, 0)
        monkeypatch.setenv("RT_AMD_DEFER_NODES", "1000")  # far too few node records: the one kernel takes over
        again, as_ = ds.render_frame(st, W, H)
        assert ds.deferred_shadows == (1, 1)
    assert np.array_equal(three, one) and np.array_equal(again, one)
    x0, y0, w, h = rect
    assert np.array_equal(part[y0:y0 + h, x0:x0 + w], one[y0:y0 + h, x0:x0 + w])
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "shaded_hits", "texture_fetches", "pixels"):
        assert getattr(ts, k) == getattr(os_, k) == getattr(as_, k), k
    assert as_.rays_shadow_traced == os_.rays_shadow_traced  # the fall-back is the one kernel
    assert ts.rays_shadow_traced >= os_.rays_shadow_traced if cone_walk else ts.rays_shadow_traced == os_.rays_shadow_traced


FLAGS = [abi.RT_F_NORMALIZE_PRIMARY, abi.RT_F_SKY_Y, abi.RT_F_TIE_LATER_WINS, abi.RT_F_SHADOW_SELF_SKIP, abi.RT_F_SQUARE_UV_NO_ORIGIN,
         abi.RT_F_TEX_U8_TAPS, abi.RT_F_PIXEL_MAP_TRUNC, abi.RT_F_MESH_RAW_DIR]


@pytest.mark.parametrize("flag", FLAGS)
def test_each_semantics_flag(flag):
    """Every CPU-vs-CUDA difference bit, toggled alone on top of both presets."""
    hs = scene("t_icosa")
    tex, cam = (90, 66), (96, 72)  # texture != viewport so the pixel map matters
    with device_scene(hs) as ds:
        for base in (abi.RT_MODE_CPU_REF, abi.RT_MODE_GPU_REF):
            st = orc.settings(spp=2, seed=9, flags=base ^ flag)
            ref, _, _ = orc.render(hs, st, *tex, cam=cam)
            img, _ = ds.render_frame(st, *tex, cam=cam)
            assert_colour_close(img, ref, f"flag {flag:#x} on base {base:#x}")


def test_partition_and_rectangles_do_not_change_the_image():
    hs = scene("t_icosa")
    W, H = 100, 70
    st = orc.settings(spp=4, dof=True, focal=9.0, aperture=1.0, seed=4, flags=abi.RT_MODE_GPU_REF)
    with device_scene(hs) as ds:
        full, fs = ds.render_frame(st, W, H)
        merged = np.zeros_like(full)
        rays = 0
        for part in range(3):
            _, s = ds.render_frame(st, W, H, tile=16, part=part, n_parts=3, out=merged)
            rays += s.rays
        assert np.array_equal(merged, full) and rays == fs.rays
        rect = (13, 9, 51, 37)
        canvas = np.full_like(full, 7)
        ds.render_tile(st, W, H, rect, out=canvas)
        x0, y0, w, h = rect
        assert np.array_equal(canvas[y0:y0 + h, x0:x0 + w], full[y0:y0 + h, x0:x0 + w])
        canvas[y0:y0 + h, x0:x0 + w] = 7
        assert (canvas == 7).all(), "pixels outside the rectangle were touched"
        again, _ = ds.render_frame(st, W, H)
        assert np.array_equal(again, full), "two renders of the same frame differ"


def test_device_destination_and_stream():
    import torch
    hs = scene("c1_purple_sphere")
    W, H = 160, 120
    st = orc.settings(spp=1, flags=abi.RT_MODE_CPU_REF)
    with device_scene(hs) as ds:
        host, _ = ds.render_frame(st, W, H)
        s = torch.cuda.Stream()
        ds.set_stream(s.cuda_stream)
        dev = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ds.render_frame(st, W, H, out=dev)
        ds.set_stream(0)
    assert np.array_equal(dev.cpu().numpy(), host)


def test_errors_are_codes_not_exits():
    hs = scene("c1_purple_sphere")
    st = orc.settings()
    with device_scene(hs) as ds:
        with pytest.raises(abi.RtError) as e:
            ds.render_tile(st, 64, 48, (60, 0, 10, 10))
        assert e.value.code == -1 and "rectangle" in str(e.value)
        st.spp = 0
        with pytest.raises(abi.RtError):
            ds.render_frame(st, 64, 48)


def test_python_ownership_mirror_matches_device_tiles():
    """cuda_raytracing_amd.distributed.owned_mask must pick exactly the pixels rt_render_frame renders."""
    from cuda_raytracing_amd.distributed import owned_mask
    hs = scene("c1_purple_sphere")
    W, H, tile, world = 100, 70, 16, 3
    st = orc.settings()
    with device_scene(hs) as ds:
        for part in range(world):
            canvas = np.zeros((H, W, 4), dtype=np.uint8)  # alpha 0 = untouched, the kernel writes alpha 255
            ds.render_frame(st, W, H, tile=tile, part=part, n_parts=world, out=canvas)
            assert np.array_equal(canvas[..., 3] == 255, owned_mask(W, H, tile, part, world))


def test_headless_cli_and_renderer_class(tmp_path):
    """The C++ front end (rt_render) and the Python Renderer mirror produce the same BMP bytes."""
    import subprocess
    from cuda_raytracing_amd import Renderer
    from helpers import ROOT
    out_cli, out_py = tmp_path / "cli.bmp", tmp_path / "py.bmp"
    exe = os.path.join(ROOT, "cuda_raytracing_amd", "rt_render")
    p = subprocess.run([exe, "--scene", os.path.join(SCENES, "t_icosa.txt"), "--assets", os.path.join(ROOT, "assets"), "--out", str(out_cli),
                        "--size", "96x72", "--spp", "4", "--dof", "1.5", "9", "--seed", "1"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    r = Renderer(96, 72, 96, 72, os.path.join(SCENES, "t_icosa.txt"))
    r.RenderInParallel(str(out_py), 4, 1.5, 9.0)
    a, b = out_cli.read_bytes(), out_py.read_bytes()
    assert len(a) == 54 + 96 * 72 * 3 and a == b
    # segment by segment (the reference's progressive mode): the same frame, plus one file per row of segments
    out_prog = tmp_path / "prog.bmp"
    p = subprocess.run([exe, "--scene", os.path.join(SCENES, "t_icosa.txt"), "--assets", os.path.join(ROOT, "assets"), "--out", str(out_prog),
                        "--size", "96x72", "--spp", "4", "--dof", "1.5", "9", "--seed", "1", "--progressive", "4", "--sequence", str(tmp_path / "seq")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert out_prog.read_bytes() == a and "progress 100.0%" in p.stderr
    assert sorted(f.name for f in tmp_path.glob("seq_*.bmp")) == [f"seq_{k:04d}.bmp" for k in range(1, 5)]
    assert (tmp_path / "seq_0004.bmp").read_bytes() == a and (tmp_path / "seq_0001.bmp").read_bytes() != a
    # a missing scene is an error message and a non-zero exit status, not a crash
    p = subprocess.run([exe, "--scene", "/nonexistent.txt"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "cannot open scene file" in p.stderr


def test_headline_frame_properties():
    """The bench configuration at full size (c4_lucy, 1920x1080, 32 spp, depth of field): properties that do not need
    a live oracle - the pruned walk gives the strict walk's image and ray counts, the frame does not depend on how it is
    cut into parts - and the whole frame, pixel for pixel, against the frame the CPU oracle rendered (committed fixture)."""
    hs = scene("c4_lucy")
    W, H = 1920, 1080
    st = orc.settings(spp=32, dof=True, focal=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
    with device_scene(hs) as ds:
        fast, fstats = ds.render_frame(st, W, H)
        parts = [ds.render_frame(st, W, H, tile=32, part=k, n_parts=3)[0] for k in range(3)]
        st.flags |= abi.RT_F_STRICT_TRAVERSAL
        strict, sstats = ds.render_frame(st, W, H)
    assert np.array_equal(fast, strict), "pruned traversal changed the full-size frame"
    assert fstats.rays == sstats.rays and fstats.rays_primary == W * H * 32
    from cuda_raytracing_amd.distributed import owned_mask
    whole = np.zeros_like(fast)
    for k in range(3):
        m = owned_mask(W, H, 32, k, 3)
        whole[m] = parts[k][m]
    assert np.array_equal(whole, fast), "the frame depends on its partition"
    # every pixel of the frame against the CPU oracle's frame (tests/golden/render/c4_lucy_1080p32.npz, made by
    # tools/make_fullframe_golden.py: every ray traced in reference order), and the reference-order counters against its totals
    g = np.load(os.path.join(GOLD, "render", "c4_lucy_1080p32.npz"))
    assert (int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])) == (W, H, 32, 1) and int(g["flags"]) == abi.RT_MODE_GPU_REF
    ref = np.stack([g["r"], g["g"], g["b"], np.full((H, W), 255, np.uint8)], -1)
    assert_colour_close(fast, ref, "headline frame against the oracle's frame")
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits", "texture_fetches", "pixels"):
        assert getattr(sstats, k) == int(g["ctr_" + k]), (k, getattr(sstats, k), int(g["ctr_" + k]))
    assert fstats.rays_shadow_traced < fstats.rays_shadow and sstats.rays_shadow_traced == sstats.rays_shadow


@pytest.mark.parametrize("name,W,H,spp,aperture,focal,blocks", [
    ("c2_textured_mirror", 1280, 720, 128, 10.0, 10.2, [(632, 352), (200, 100), (1000, 600)]),
    ("c3_orange", 1920, 1080, 64, 10.0, 14.5, [(952, 532), (400, 300), (1500, 800)]),
    ("c4_lucy", 3840, 2160, 32, 1.0, 1.35, [(1912, 1072), (1200, 600)]),
    ("c5_final", 1920, 1080, 128, 150.0, 168.0, [(952, 532), (700, 700)]),
])
def test_other_configs_at_full_size(name, W, H, spp, aperture, focal, blocks):
    """The remaining benchmark configurations at their full sizes: the frame does not depend on its partition, the ray
    count is what the oracle counts per path, and 8x8 blocks match the oracle at the full sample count."""
    hs = scene(name)
    st = orc.settings(spp=spp, dof=True, focal=focal, aperture=aperture, seed=1, flags=abi.RT_MODE_GPU_REF)
    meshes = name in ("c4_lucy", "c5_final")
    with device_scene(hs) as ds:
        whole, stats = ds.render_frame(st, W, H)
        parts = [ds.render_frame(st, W, H, tile=32, part=k, n_parts=2) for k in range(2)]
        if meshes:  # the whole frame once more in reference order (7-9 s): same image, and the node / triangle totals for the fixture below
            st.flags |= abi.RT_F_STRICT_TRAVERSAL
            strict, sstats = ds.render_frame(st, W, H)
            st.flags &= ~abi.RT_F_STRICT_TRAVERSAL
            assert np.array_equal(strict, whole), "pruned traversal changed the full-size frame"
            assert sstats.rays == stats.rays and sstats.rays_shadow_traced == sstats.rays_shadow
    assert stats.rays_primary == W * H * spp
    assert parts[0][1].rays + parts[1][1].rays == stats.rays
    from cuda_raytracing_amd.distributed import owned_mask
    glued = np.zeros_like(whole)
    for k in range(2):
        m = owned_mask(W, H, 32, k, 2)
        glued[m] = parts[k][0][m]
    assert np.array_equal(glued, whole), "the frame depends on its partition"
    for (x0, y0) in blocks:
        ref, _, _ = orc.render(hs, st, W, H, rect=(x0, y0, 8, 8))
        assert_colour_close(whole[y0:y0 + 8, x0:x0 + 8], ref[y0:y0 + 8, x0:x0 + 8], f"{name} block at {x0},{y0}")
    # every configuration has the CPU oracle's own full frame committed (tools/make_fullframe_golden.py; minutes of the oracle for
    # C2 / C3, over an hour each for C4 and C5): every pixel, the ray / shading totals of the frame and, for the mesh scenes, the
    # node and triangle totals of the reference-order traversal
    g = np.load(os.path.join(GOLD, "render", f"{name}_{H}p{spp}.npz"))
    assert (int(g["width"]), int(g["height"]), int(g["spp"]), int(g["seed"])) == (W, H, spp, 1)
    assert float(g["focal"]) == focal and float(g["aperture"]) == aperture and int(g["flags"]) == abi.RT_MODE_GPU_REF
    ref = np.stack([g["r"], g["g"], g["b"], np.full((H, W), 255, np.uint8)], -1)
    assert_colour_close(whole, ref, f"{name} full frame against the oracle's frame")
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "shaded_hits", "texture_fetches", "pixels"):
        assert getattr(stats, k) == int(g["ctr_" + k]), (k, getattr(stats, k), int(g["ctr_" + k]))
    if meshes:
        for k in ("nodes_visited", "tris_tested"):
            assert getattr(sstats, k) == int(g["ctr_" + k]), (k, getattr(sstats, k), int(g["ctr_" + k]))


def test_final_scene_full_frame_against_oracle_fixture():
    """C5's scene (Lucy + bunny, 102,897 triangles, three sphere lamps) at its full 1920x1080 with 4 of its 128 samples per pixel:
    every pixel against the frame the CPU oracle rendered (committed fixture), reference-order counters against its totals."""
    g = np.load(os.path.join(GOLD, "render", "c5_final_1080p4.npz"))
    W, H, spp = int(g["width"]), int(g["height"]), int(g["spp"])
    assert (W, H, spp) == (1920, 1080, 4)
    hs = scene("c5_final")
    st = orc.settings(spp=spp, dof=True, focal=float(g["focal"]), aperture=float(g["aperture"]), seed=int(g["seed"]), flags=int(g["flags"]))
    with device_scene(hs) as ds:
        assert ds.waves_per_simd == 4 and ds.shared_walk[0] == 1  # the mesh-heavy build
        fast, fstats = ds.render_frame(st, W, H)
        st.flags |= abi.RT_F_STRICT_TRAVERSAL
        strict, sstats = ds.render_frame(st, W, H)
    assert np.array_equal(fast, strict), "pruned traversal changed the frame"
    ref = np.stack([g["r"], g["g"], g["b"], np.full((H, W), 255, np.uint8)], -1)
    assert_colour_close(fast, ref, "c5_final frame against the oracle's frame")
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits", "texture_fetches", "pixels"):
        assert getattr(sstats, k) == int(g["ctr_" + k]), (k, getattr(sstats, k), int(g["ctr_" + k]))
    assert fstats.rays == sstats.rays


def test_4k_frame_bands_against_oracle_fixture():
    """C4 (c4_lucy at 3840x2160x32 with its lens): 34 bands of 8 rows spread over the whole height (1.04 M of the 8.3 M pixels)
    against what the CPU oracle rendered for them (committed fixture); reference-order counters of the bands against its totals."""
    g = np.load(os.path.join(GOLD, "render", "c4_lucy_2160p32_bands.npz"))
    W, H, spp, rows = int(g["width"]), int(g["height"]), int(g["spp"]), int(g["band_rows"])
    assert (W, H, spp) == (3840, 2160, 32)
    hs = scene("c4_lucy")
    st = orc.settings(spp=spp, dof=True, focal=float(g["focal"]), aperture=float(g["aperture"]), seed=int(g["seed"]), flags=int(g["flags"]))
    keys = ("rays_primary", "rays_secondary", "rays_shadow", "nodes_visited", "tris_tested", "shaded_hits", "texture_fetches", "pixels")
    total = dict.fromkeys(keys, 0)
    with device_scene(hs) as ds:
        fast, fstats = ds.render_frame(st, W, H)
        st.flags |= abi.RT_F_STRICT_TRAVERSAL
        strict = np.zeros((H, W, 4), np.uint8)
        for y0 in g["band_y0"]:
            _, s = ds.render_tile(st, W, H, (0, int(y0), W, rows), out=strict)
            for k in keys:
                total[k] += getattr(s, k)
    pick = np.concatenate([np.arange(int(y0), int(y0) + rows) for y0 in g["band_y0"]])
    assert np.array_equal(fast[pick], strict[pick]), "pruned traversal changed the bands"
    ref = np.stack([g["r"], g["g"], g["b"], np.full(g["r"].shape, 255, np.uint8)], -1)
    assert ref.shape[0] == len(pick) == 34 * 8
    assert_colour_close(fast[pick], ref, "c4_lucy 4K bands against the oracle's")
    for k in keys:
        assert total[k] == int(g["ctr_" + k]), (k, total[k], int(g["ctr_" + k]))
    assert fstats.pixels == W * H


def test_more_than_64_primitives(tmp_path):
    """Above 64 primitives the per-lamp candidate masks are off and the shading table stays in global memory."""
    rng = np.random.default_rng(7)
    text = ["background\n top= 0.4 0.6 0.9\n bottom= 0.9 0.9 0.8\nend\n",
            "camera\n O= 0 -14 5\n N= 0 1 -0.3\n lens_W= 1.0\n lens_H= 0.75\n image_W= 96\n image_H= 72\n shade_quality= 0.5\nend\n",
            "light sphere\n O= 4 -5 9\n R= 0.5\n color= 1 1 1\nend\n", "light point\n O= -6 -4 7\n color= 0.4 0.4 0.5\nend\n",
            "primitive plane\n N= 0 0 1\n R= 0\n Dx= 3 0 0\n Dy= 0 3 0\n color= 1 1 1\n texture= floor.bmp\n diff= 0.8\n refl= 0.1\nend\n"]
    for k in range(70):
        x, y = -7 + 14 * (k % 10) / 9, -3 + 8 * (k // 10) / 6
        r = 0.25 + 0.3 * rng.random()
        c = 0.3 + 0.7 * rng.random(3)
        kind = ["diff= 0.7\n spec= 0.3", "refl= 0.7\n diff= 0.2", "refr= 0.9\n rindex= 1.5\n refl= 0.1"][k % 3]
        text.append(f"primitive sphere\n O= {x:.3f} {y:.3f} {r:.3f}\n R= {r:.3f}\n color= {c[0]:.3f} {c[1]:.3f} {c[2]:.3f}\n {kind}\nend\n")
    path = tmp_path / "many.txt"
    path.write_text("\n".join(text))
    hs = HostScene(str(path))
    assert hs.desc.n_primitives > 64
    W, H = 96, 72
    st = orc.settings(spp=2, dof=True, focal=14.0, aperture=2.0, seed=3, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    ref, _, ostats = orc.render(hs, st, W, H)
    with device_scene(hs) as ds:
        img, stats = ds.render_frame(st, W, H)
        st.flags = abi.RT_MODE_GPU_REF
        fast, fstats = ds.render_frame(st, W, H)
    assert_colour_close(img, ref, "70 spheres")
    assert stats.rays == ostats.rays and fstats.rays == stats.rays
    assert np.array_equal(fast, img)


EDGE_HEAD = "background\n top= 0.2 0.5 0.9\n bottom= 0.9 0.8 0.7\nend\ncamera\n O= 0 -6 1.5\n N= 0 1 -0.1\n image_W= 64\n image_H= 48\nend\n"


@pytest.mark.parametrize("body", [
    "",                                                                                             # nothing but sky
    "primitive sphere\n O= 0 0 1\n R= 1\n color= 1 0.5 0.2\n diff= 0.8\n refl= 0.3\nend\n",       # geometry, no lamp
    "light point\n O= 2 -3 5\n color= 1 1 1\nend\n",                                               # a lamp, nothing to light
    "light sphere\n O= 2 -3 5\n R= 0.4\n color= 1 1 1\nend\nprimitive mesh\n O= 0 0 1\n scale= 1 1 1\n rotation= 0.2 0.1 0\n file= one.obj\n"
    " color= 0.8 0.8 0.3\n diff= 0.9\nend\nprimitive mesh\n O= 1.5 0.5 1\n scale= 1 1 1\n rotation= 0 0.3 0.2\n file= two.obj\n color= 0.3 0.8 0.8\n"
    " diff= 0.7\n refl= 0.2\nend\nprimitive plane\n N= 0 0 1\n R= 0\n color= 1 1 1\n diff= 0.8\nend\n",  # one- and two-triangle meshes: no tree to speak of
], ids=["empty", "no-lamp", "lamp-only", "tiny-meshes"])
def test_degenerate_scenes(tmp_path, body):
    (tmp_path / "one.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 0 0 1\nf 1 2 3\n")
    (tmp_path / "two.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nf 1 2 3\nf 1 3 4\n")
    (tmp_path / "s.txt").write_text(EDGE_HEAD + body)
    hs = HostScene(str(tmp_path / "s.txt"), asset_dir=str(tmp_path))
    W, H = 64, 48
    st = orc.settings(spp=2, dof=True, focal=6.0, aperture=1.0, seed=9, flags=abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL)
    ref, _, ostats = orc.render(hs, st, W, H)
    o_hit, o_t, o_leaf, _, _ = orc.trace_primary(hs, st, W, H)
    with device_scene(hs) as ds:
        img, stats = ds.render_frame(st, W, H)
        hit, t, leaf, _, _ = ds.trace_primary(st, W, H)
        st.flags = abi.RT_MODE_GPU_REF
        fast, _ = ds.render_frame(st, W, H)
    assert np.array_equal(hit, o_hit) and np.array_equal(leaf, o_leaf) and same_double(t, o_t)
    assert_colour_close(img, ref, "degenerate scene")
    assert stats.rays == ostats.rays
    assert np.array_equal(fast, img)
