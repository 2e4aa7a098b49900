"""Randomised scenes against the shortcuts of the pruned render: candidate masks per lamp, umbra tests, bounding capsules,
the candidate grid, the four-wide walk with its single-precision boxes over trees from all three builders, the shared shadow walk.  Every one of them must
be invisible: the pruned image equals the reference-order image bit for bit and the ray totals are the same.  A sample of
the scenes is also checked against the CPU oracle, so that "both GPU modes wrong in the same way" cannot pass.

The generator aims at what the shortcuts are made of: lamps of all three kinds, some inside or right behind objects;
spheres and cylinders large enough to put whole regions into their umbra; huge wall spheres around the scene;
axis-parallel squares and planes; thin long cylinders and Bezier stacks (capsule tests); small meshes, several at once."""
import os

import numpy as np
import pytest

import oracle_binding as orc
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.host import HostScene
from helpers import ROOT

pytestmark = pytest.mark.gpu

N_SCENES = int(os.environ.get("RT_FUZZ_SCENES", "400"))      # a soak run sets these two: other seeds, more of them
FIRST_SEED = int(os.environ.get("RT_FUZZ_FIRST_SEED", "0"))
BIG_MESHES = os.environ.get("RT_FUZZ_BIG_MESHES") == "1"  # half of the meshes are the benchmark scenes' large ones (deep trees, long walks)
ORACLE_EVERY = int(os.environ.get("RT_FUZZ_ORACLE_EVERY", "13"))  # every how many scenes the CPU oracle renders the scene as well
# (a soak run may also render larger frames with more samples: more and deeper paths per scene)
W, H, SPP = int(os.environ.get("RT_FUZZ_W", "64")), int(os.environ.get("RT_FUZZ_H", "48")), int(os.environ.get("RT_FUZZ_SPP", "2"))


def _v(a):
    return " ".join(f"{x:.4f}" for x in a)


def _material(rng, allow_glass=True):
    kind = rng.integers(0, 5 if allow_glass else 3)
    c = 0.2 + 0.8 * rng.random(3)
    lines = [f" color= {_v(c)}"]
    if kind == 0:
        lines += [f" diff= {0.5 + 0.5 * rng.random():.3f}"]
    elif kind == 1:
        lines += [f" diff= {0.3 + 0.5 * rng.random():.3f}", f" spec= {0.5 * rng.random():.3f}", f" refl= {0.4 * rng.random():.3f}"]
        if rng.random() < 0.3:
            lines += [f" drefl= {0.2 * rng.random():.3f}"]
    elif kind == 2:
        lines += [f" refl= {0.4 + 0.5 * rng.random():.3f}", f" diff= {0.3 * rng.random():.3f}"]
    elif kind == 3:
        lines += [f" refr= {0.6 + 0.4 * rng.random():.3f}", f" rindex= {1.2 + 0.6 * rng.random():.3f}", f" absor= {_v(rng.random(3))}"]
    else:
        lines += [f" refr= {0.5 + 0.4 * rng.random():.3f}", f" refl= {0.3 * rng.random():.3f}", f" rindex= {1.3 + 0.5 * rng.random():.3f}",
                  f" diff= {0.3 * rng.random():.3f}"]
    return "\n".join(lines) + "\n"


def make_scene(seed, tmp):
    rng = np.random.default_rng(seed)
    sq = [0.25, 0.5, 1.0][rng.integers(0, 3)]
    t = [f"background\n top= {_v(rng.random(3))}\n bottom= {_v(rng.random(3))}\nend\n",
         f"camera\n O= {_v([rng.normal(0, 1), -9 + rng.normal(0, 1), 2.5 + rng.normal(0, 0.7)])}\n N= {_v([rng.normal(0, 0.1), 1, -0.2 + rng.normal(0, 0.1)])}\n"
         f" lens_W= 1.0\n lens_H= 0.75\n image_W= {W}\n image_H= {H}\n shade_quality= {sq}\n drefl_quality= 1\nend\n"]
    centres = []  # where objects are, so that some lamps can go inside / right behind them
    n_obj = rng.integers(2, 9)
    floor = rng.integers(0, 3)
    if floor == 0:
        t.append(f"primitive plane\n N= 0 0 1\n R= 0\n Dx= 3 0 0\n Dy= 0 3 0\n{_material(rng, False)}" + (" texture= floor.bmp\n" if rng.random() < 0.5 else "") + "end\n")
    elif floor == 1:
        t.append(f"primitive square\n O= 0 0 0\n Dx= {6 + 4 * rng.random():.3f} 0 0\n Dy= 0 {6 + 4 * rng.random():.3f} 0\n{_material(rng, False)}end\n")
    if rng.random() < 0.3:  # walls: spheres so large that the scene sits inside them, or a huge one beside it
        r = 10.0 ** rng.uniform(2, 5)
        where = [0, 0, r - 0.5 - 8 * rng.random()] if rng.random() < 0.5 else [r + 6, 0, 0]
        t.append(f"primitive sphere\n O= {_v(where)}\n R= {r:.3f}\n{_material(rng, False)}end\n")
    if rng.random() < 0.25:
        axis = rng.integers(0, 3)
        o = [0.0, 0.0, 0.0]
        o[axis] = 5.0 + rng.random()
        dx, dy = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
        dx[(axis + 1) % 3], dy[(axis + 2) % 3] = 4.0, 4.0
        t.append(f"primitive square\n O= {_v(o)}\n Dx= {_v(dx)}\n Dy= {_v(dy)}\n{_material(rng)}end\n")
    for _ in range(n_obj):
        c = np.array([rng.normal(0, 2.2), rng.normal(0, 2.2), abs(rng.normal(1.5, 1.2))])
        centres.append(c)
        k = rng.integers(0, 8)
        if k <= 1:
            r = [0.3, 0.8, 1.6, 3.0][rng.integers(0, 4)] * (0.7 + 0.6 * rng.random())
            t.append(f"primitive sphere\n O= {_v(c)}\n R= {r:.4f}\n{_material(rng)}end\n")
        elif k == 2:
            ax = rng.normal(size=3) if rng.random() < 0.6 else np.eye(3)[rng.integers(0, 3)]
            ax = ax / np.linalg.norm(ax) * (0.5 + 4 * rng.random())
            r = [0.08, 0.4, 1.5][rng.integers(0, 3)] * (0.7 + 0.6 * rng.random())
            t.append(f"primitive cylinder\n O1= {_v(c)}\n O2= {_v(c + ax)}\n R= {r:.4f}\n{_material(rng)}" + (" texture= marble.bmp\n" if rng.random() < 0.3 else "") + "end\n")
        elif k == 3:
            ax = np.array([0, 0, 1.0]) if rng.random() < 0.5 else rng.normal(size=3)
            ax = ax / np.linalg.norm(ax) * (1 + 2 * rng.random())
            pts = "".join(f" P= {z:.3f} {r:.3f}\n" for z, r in zip(np.sort(rng.random(rng.integers(3, 7))) * 2.5, 0.2 + rng.random(7)))
            t.append(f"primitive bezier\n O1= {_v(c)}\n O2= {_v(c + ax)}\n{pts}{_material(rng, False)}end\n")
        elif k == 4:
            a, b = rng.normal(0, 1.2, 3), rng.normal(0, 1.2, 3)
            t.append(f"primitive triangle\n O1= {_v(c)}\n O2= {_v(c + a)}\n O3= {_v(c + b)}\n{_material(rng, False)}end\n")
        elif k == 5:
            a = rng.normal(size=3)
            b = np.cross(a, rng.normal(size=3))
            a, b = a / np.linalg.norm(a) * (0.5 + 2 * rng.random()), b / np.linalg.norm(b) * (0.5 + 2 * rng.random())
            t.append(f"primitive square\n O= {_v(c)}\n Dx= {_v(a)}\n Dy= {_v(b)}\n{_material(rng)}end\n")
        else:
            f = ["icosa.obj", "boxgrid.obj"][rng.integers(0, 2)]
            rot = rng.normal(0, 0.8, 3) if rng.random() < 0.7 else np.zeros(3)
            s = 0.4 + 1.4 * rng.random()
            if BIG_MESHES and rng.random() < 0.5:  # the statue (33 k triangles, unit size) or the bunny (69 k, size 0.17)
                f, s = [("lucy_34k_normalized.rtm", 0.8 + 2.2 * rng.random()), ("bunny_69k.rtm", 8.0 + 20.0 * rng.random())][rng.integers(0, 2)]
            t.append(f"primitive mesh\n O= {_v(c)}\n scale= {s:.3f} {s:.3f} {s:.3f}\n rotation= {_v(rot)}\n file= {f}\n{_material(rng)}end\n")
    for _ in range(rng.integers(1, 4)):
        mode = rng.integers(0, 4)
        if mode == 0 and centres:      # inside or at an object
            o = centres[rng.integers(0, len(centres))] + rng.normal(0, 0.2, 3)
        elif mode == 1 and centres:    # right behind an object, seen from the origin
            c = centres[rng.integers(0, len(centres))]
            o = c * (1.3 + rng.random()) + np.array([0, 0, 0.5])
        else:
            o = np.array([rng.normal(0, 4), rng.normal(0, 4), 3 + 6 * rng.random()])
        kind = rng.integers(0, 3)
        col = 0.3 + 1.2 * rng.random(3)
        if kind == 0:
            t.append(f"light point\n O= {_v(o)}\n color= {_v(col)}\nend\n")
        elif kind == 1:
            t.append(f"light sphere\n O= {_v(o)}\n R= {[0.05, 0.5, 1.5][rng.integers(0, 3)]:.3f}\n color= {_v(col)}\nend\n")
        else:
            e = 0.2 + 1.5 * rng.random()
            t.append(f"light square\n O= {_v(o)}\n Dx= {e:.3f} 0 0\n Dy= 0 {e * (0.5 + rng.random()):.3f} 0\n color= {_v(col)}\nend\n")
    path = os.path.join(tmp, f"fuzz_{seed}.txt")
    with open(path, "w") as f:
        f.write("".join(t))
    return path


def _keep(path, seed, why):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(path) as f, open(os.path.join(out, f"fuzz_fail_{seed}.txt"), "w") as g:
        g.write(f"# {why}\n" + f.read())


def check_scene(path, seed, with_oracle, waves4=None):
    """One scene, with the settings that go with its seed: None if the pruned render equals the reference-order render (and, on
    request, the oracle's), else what differs."""
    from cuda_raytracing_amd.device import DeviceScene
    hs = HostScene(path, asset_dir=os.path.join(ROOT, "assets"))
    flags = abi.RT_MODE_GPU_REF if seed % 3 else abi.RT_MODE_CPU_REF
    if seed % 5 == 0:
        flags &= ~abi.RT_F_BEZIER_STACKED
    if seed % 7 == 0:
        flags ^= abi.RT_F_MESH_RAW_DIR
    st = orc.settings(spp=SPP, dof=bool(seed % 2), focal=9.0, aperture=1.0, seed=seed + 1, flags=flags | abi.RT_F_STRICT_TRAVERSAL)
    # the pruned walk's trees from each of the three builders in turn (scenes without meshes have none)
    # (the builder turns over with seed // 3, the preset with seed % 3, the oracle sample every 13th scene: every preset meets
    # every builder, and the oracle sees all of those pairs)
    # every other pair of scenes gets the kernel build for mesh-heavy scenes where the library has one for it (four waves
    # per SIMD; it alone looks down the meshes' trees before dealing shadow rays): the rule would pick it for none of these
    if (seed // 2) % 2 if waves4 is None else waves4:
        os.environ["RT_AMD_WAVES"] = "4"
    try:
        ds_cm = DeviceScene(hs, device=0, tree_builder=("host", "device_sah", "device")[(seed // 3) % 3])
    finally:
        os.environ.pop("RT_AMD_WAVES", None)
    with ds_cm as ds:
        ds.check_trees()
        strict, ss = ds.render_frame(st, W, H)
        st.flags = flags
        fast, fs = ds.render_frame(st, W, H)
    why = None
    if not np.array_equal(fast, strict):
        bad = np.argwhere((fast != strict).any(-1))
        why = f"pruned image differs from the reference-order image at {len(bad)} pixels, first {bad[0].tolist()}"
    else:
        for k in ("rays_primary", "rays_secondary", "rays_shadow", "shaded_hits", "texture_fetches", "pixels"):
            if getattr(fs, k) != getattr(ss, k):
                why = f"{k}: pruned {getattr(fs, k)} != reference-order {getattr(ss, k)}"
        if fs.rays_shadow_traced > fs.rays_shadow or ss.rays_shadow_traced != ss.rays_shadow:
            why = "traced shadow rays inconsistent"
    if why is None and with_oracle:  # both GPU modes could be wrong together
        st.flags = flags | abi.RT_F_STRICT_TRAVERSAL
        ref, _, os_ = orc.render(hs, st, W, H)
        d = np.abs(strict[..., :3].astype(int) - ref[..., :3].astype(int))
        if d.max() > 1 or (d.max(-1) > 0).mean() > 0.01:
            why = f"reference-order image differs from the oracle: max {d.max()} levels, {(d.max(-1) > 0).mean():.3%} of pixels"
        elif ss.rays != os_.rays or ss.nodes_visited != os_.nodes_visited or ss.tris_tested != os_.tris_tested:
            why = f"counters differ from the oracle: rays {ss.rays} vs {os_.rays}, nodes {ss.nodes_visited} vs {os_.nodes_visited}"
    return why


def test_pruned_render_equals_reference_order_render_on_random_scenes(tmp_path):
    failures = []
    checked_oracle = 0
    for seed in range(FIRST_SEED, FIRST_SEED + N_SCENES):
        if seed % 100 == 0:
            print(f"fuzz: seed {seed}, {len(failures)} failures so far", flush=True)  # visible with -s: a soak run must not look hung
        path = make_scene(seed, str(tmp_path))
        why = check_scene(path, seed, with_oracle=seed % ORACLE_EVERY == 0)
        checked_oracle += seed % ORACLE_EVERY == 0
        if why:
            _keep(path, seed, why)
            failures.append((seed, why))
    abi.load_library().rt_set_tree_builder(0)
    assert not failures, f"{len(failures)} of {N_SCENES} scenes failed (scene files kept in gpurun_out/fuzz_fail_<seed>.txt): {failures[:5]}"
    assert checked_oracle >= N_SCENES // ORACLE_EVERY


REGRESS = sorted(f for f in os.listdir(os.path.join(ROOT, "tests", "golden", "fuzz_regress")) if f.startswith("seed_") and f.endswith(".txt"))


@pytest.mark.parametrize("name", REGRESS)
def test_scenes_a_soak_run_once_failed_on(name):
    """Scenes of soak runs (RT_FUZZ_SCENES / RT_FUZZ_FIRST_SEED) on which the pruned render once differed, kept as written by the
    generator, with both kernel builds and against the oracle.  seed_33674: a ray leaves a wall sphere of radius 1e5 towards a point
    lamp behind a thin cylinder and grazes the cylinder 2e-5 inside its surface; at that distance the reference's quadratic
    (CR/Collision.cu:3-107) carries more rounding error than that and reports a miss, while the umbra test of the shadow candidates,
    which reasons about exact geometry with 1e-6 margins, declared the lamp hidden.  The tests that predict an intersector's answer
    now keep clear of its rounding noise (isect_noise)."""
    seed = int(name[len("seed_"):-len(".txt")])
    path = os.path.join(ROOT, "tests", "golden", "fuzz_regress", name)
    for waves4 in (False, True):
        why = check_scene(path, seed, with_oracle=not waves4, waves4=waves4)
        assert why is None, (name, waves4, why)
