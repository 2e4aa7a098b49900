#!/usr/bin/env python3
"""Benchmark of the render path through the C ABI, on N GPUs of one node.

  python bench.py [--config headline|c1|c2|c3|c4|c5] [--gpus N] [--steps K] [--warmup W]

--gpus N > 1 without a torch.distributed environment starts the N ranks itself (one process per GPU, before anything
touches a GPU); it can also be launched as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full frame of the configuration: every rank renders its tiles (tile t -> rank t % N) into device memory,
rank 0 gathers them.  The scene is resident in HBM before the timed region; rank 0 prints one JSON line.

What the line reports
  value                    rays of the frame per second of the slowest rank, counting every ray the reference traces for
                           this frame (primary + reflect/refract + one shadow ray per light sample per shaded point).
                           Most shadow samples are decided here without a ray (lamp behind the surface, nothing that can
                           stand in between, point inside an umbra - the image is bit-identical to tracing them all), so
                           this is a rate of rays *resolved*;
  mrays_per_s_traced       the same with only the shadow rays that were actually intersected with the scene;
  mrays_per_s_primary_only pixels x samples per second, the one figure that can be derived for the reference's own timings;
  roofline                 the render kernel against the roof that binds it: vector-ALU issue.  `achieved` = vector
                           instructions of one launch (SQ_INSTS_VALU of the same command under rocprofv3, profiles/current/)
                           x 64 lanes / the kernel time measured live with HIP events; `peak` prices every instruction at
                           4 cycles, so `frac` is issue-slot occupancy, not efficiency.  What is under it is printed beside
                           it: `rate_weighted_issue_frac` (4 cycles for double precision, 2 for the rest), `lane_utilisation`
                           (active lanes per vector instruction), `fp64_share`, `useful_frac` (double-precision operations
                           of active lanes against the 78.6 TFLOP/s FMA peak).  The counters are used only if the file
                           carries the build id of the library that is being timed (rt_build_id); otherwise `frac` is null
                           and `counters_stale` true.  `flops_model` prices the launch with per-operation flop counts;
                           `hbm` puts the HBM counter traffic next to the bytes the kernel has to move;
                           `reference_order_pass` is the same frame traversed as the reference does, with the bytes of the
                           64 / 72 / 48 / 128 / 16 / 4-byte model (node, triangle, ray, shaded hit, texture fetch, pixel)
                           against the HBM peak; `survey_model` holds the same bytes against the pruned kernel's time.
  cpu_baseline             the CPU oracle (a port of the reference algorithm, OpenMP, every ray traced in reference order)
                           on a bounded sample of the same frame; for c1 also the reference's own CPU renderer.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E
FP64_VECTOR_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz
SIMDS = 1024
CLOCK_GHZ = 2.4
VALU_LANE_OPS_PEAK = SIMDS * 16 * CLOCK_GHZ / 1e3  # 39.3 T lane-instructions/s: one 64-lane instruction per SIMD per 4 cycles

# name -> (scene, width, height, spp, dof, aperture, focal, semantics preset): BASELINE.json:configs, CR/main.cpp:5-16
CONFIGS = {
    "c1": ("c1_purple_sphere", 640, 480, 1, False, 10.0, 10.0, "cpu_ref"),
    "c2": ("c2_textured_mirror", 1280, 720, 128, True, 10.0, 10.2, "gpu_ref"),
    "c3": ("c3_orange", 1920, 1080, 64, True, 10.0, 14.5, "gpu_ref"),
    "headline": ("c4_lucy", 1920, 1080, 32, True, 1.0, 1.35, "gpu_ref"),
    "c4": ("c4_lucy", 3840, 2160, 32, True, 1.0, 1.35, "gpu_ref"),
    "c5": ("c5_final", 1920, 1080, 128, True, 150.0, 168.0, "gpu_ref"),
}


def describe(hs, abi):
    """Workload string from the scene itself."""
    d = hs.desc
    names = {abi.RT_PRIM_SPHERE: "sphere", abi.RT_PRIM_PLANE: "plane", abi.RT_PRIM_SQUARE: "square", abi.RT_PRIM_CYLINDER: "cylinder",
             abi.RT_PRIM_BEZIER: "Bezier", abi.RT_PRIM_TRIANGLE: "triangle", abi.RT_PRIM_MESH: "mesh"}
    counts = {}
    for i in range(d.n_primitives):
        if d.primitives[i].is_light:
            continue
        n = names[d.primitives[i].type]
        counts[n] = counts.get(n, 0) + 1
    lamps = {}
    for i in range(d.n_lights):
        n = {abi.RT_LIGHT_POINT: "point", abi.RT_LIGHT_SQUARE: "square", abi.RT_LIGHT_SPHERE: "sphere"}[d.lights[i].type]
        lamps[n] = lamps.get(n, 0) + 1
    prims = ", ".join(f"{v} {k}{'s' if v > 1 and not k.endswith('s') else ''}" for k, v in sorted(counts.items()))
    lights = ", ".join(f"{v} {k} lamp{'s' if v > 1 else ''}" for k, v in sorted(lamps.items()))
    tris = f" ({d.n_leaves} triangles)" if d.n_leaves else ""
    return f"{prims}{tris}; {lights}"


def flops_model(s):
    """Double-precision operations the launch performs, from its own counters: per four-wide node visited 4 slab tests
    of 30 flop (six planes, compare and select; they run in single precision here), 60 per triangle test, 40 per analytic
    primitive test (every ray meets every top-level primitive of the scene), 50 per shaded hit and light, 25 per texture
    fetch.  A model, not a measurement: it prices useful arithmetic, the instruction count prices everything."""
    rays = s["rays_traced"]
    return (120.0 * s["nodes_visited"] + 60.0 * s["tris_tested"] + 40.0 * rays * s["n_prims"] + 50.0 * s["shaded_hits"] * max(1, s["n_lights"])
            + 25.0 * s["texture_fetches"])


def cpu_baseline(hs, st, width, height, budget_s=25.0):
    """Oracle (port of the reference algorithm) on a lattice of 32x32 blocks of the same frame, about budget_s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as orc

    lib = orc.load()
    cores = lib.orc_max_threads()
    b = 32 if min(width, height) >= 256 else 16
    nx, ny = 16, 12
    rects = [(min(max(int((i + 0.5) * width / nx) - b // 2, 0), width - b), min(max(int((j + 0.5) * height / ny) - b // 2, 0), height - b), b, b)
             for j in range(ny) for i in range(nx)]
    # blocks in an order that covers the frame evenly however early the budget ends: every 8th, then every 4th, ...
    order, seen = [], set()
    for stride in (8, 4, 2, 1):
        for k in range(0, len(rects), stride):
            if k not in seen:
                seen.add(k)
                order.append(rects[k])
    rays, done = 0, 0
    t0 = time.perf_counter()
    for r in order:
        if done >= 12 and time.perf_counter() - t0 > budget_s:
            break
        _, _, s = orc.render(hs, st, width, height, rect=r)
        rays += s.rays
        done += 1
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": int(cores), "kind": "port",
            "sample": f"{done} blocks of {b}x{b} pixels spread evenly over the same {width}x{height}x{st.spp}spp frame "
                      f"({rays} rays in {dt:.1f} s; every ray traced, reference-order traversal, OpenMP over rows)"}


def profile_counters(config, build_id):
    """Counters of one render-kernel launch of this configuration (one frame), from the rocprofv3 PMC passes of the same
    command that tools/profile_round.sh summarised under profiles/current/ (rocprofv3 cannot run inside this process).
    Returns (counters, path, stale): counters collected with another build of the library are not returned."""
    path = os.path.join(ROOT, "profiles", "current", f"pmc_{config}.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
    except Exception:
        return None, None, False
    if pmc.get("build_id") != build_id:
        return None, os.path.relpath(path, ROOT), True
    return pmc, os.path.relpath(path, ROOT), False


def algorithmic_bytes(s):
    """Bytes the reference's own traversal of the frame touches: 64 per node visited, 72 per triangle tested, 48 per ray (origin
    and direction), 128 per shaded hit (primitive / material record), 16 per texture fetch, 4 per pixel written.  The counts
    must come from a reference-order render (every ray traced, no culling)."""
    return (64.0 * s.nodes_visited + 72.0 * s.tris_tested + 48.0 * s.rays + 128.0 * s.shaded_hits + 16.0 * s.texture_fetches
            + 4.0 * s.pixels)


def reference_c1():
    """The reference's own CPU renderer (oracle/_ref/ref_dump, both passes of Raytracer::Run) on config 1."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.run([exe, "run", os.path.join(ROOT, "scenes", "c1_purple_sphere.txt"), "/tmp/_c1_ref.bmp"], cwd=os.path.join(ROOT, "assets"),
                           capture_output=True, text=True, timeout=120)
        line = [l for l in p.stderr.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--scene", default=None, help="override the configuration's scene")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the reference-order pass (headline only) that counts the reference's node visits")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks ourselves; nothing in this process has touched a GPU yet (torch is not even imported)
        from cuda_raytracing_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(__file__, args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from cuda_raytracing_amd import abi
    from cuda_raytracing_amd.device import DeviceScene
    from cuda_raytracing_amd.distributed import FrameGather
    from cuda_raytracing_amd.host import SCENE_DIR, HostScene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # RT_BENCH_REHEARSE=1: a dress rehearsal of the multi-rank path on a box with fewer GPUs than ranks - all ranks render on
    # GPU 0 and the exchange runs over gloo on host copies of the frames.  The partition, the launcher, the packing order
    # and the assembled frame are the real ones (rank 0 checks the frame against a single-rank render); the timings are not.
    rehearse = world > 1 and os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    comm = "cpu" if rehearse else "cuda"

    scene_name, W, H, spp, dof, aperture, focal, preset = CONFIGS[args.config]
    scene_name, W, H, spp = args.scene or scene_name, args.width or W, args.height or H, args.spp or spp
    flags = abi.RT_MODE_GPU_REF if preset == "gpu_ref" else abi.RT_MODE_CPU_REF
    hs = HostScene(os.path.join(SCENE_DIR, scene_name + ".txt"))
    st = abi.Settings(dof_enabled=int(dof), spp=spp, focal_distance=focal, aperture=aperture, seed=1, flags=flags)
    ds = DeviceScene(hs, device=local)
    stream = torch.cuda.current_stream()
    ds.set_stream(stream.cuda_stream)
    image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    gather = FrameGather(W, H, args.tile, rank, world, torch.device(comm))
    frame = [None]

    exchange_ms = [0.0]
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def step():
        _, s = ds.render_frame(st, W, H, tile=args.tile, part=rank, n_parts=world, out=image)
        if world == 1:
            frame[0] = image
        elif rehearse:
            t = time.perf_counter()
            frame[0] = gather.gather(image.cpu())
            exchange_ms[0] += (time.perf_counter() - t) * 1e3
        else:  # pack kernel, gather over RCCL, unpack kernel: all on this stream (rank 0 also waits here for the slowest peer's pack)
            ev[0].record(stream)
            frame[0] = gather.gather(image)
            ev[1].record(stream)
            ev[1].synchronize()
            exchange_ms[0] += ev[0].elapsed_time(ev[1])
        return s

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # reference-order traversal of this rank's tiles: how many nodes / triangles the reference's walk visits for the same frame
    strict_stats = None
    if rank == 0 and args.config == "headline" and not args.no_strict:
        st.flags = flags | abi.RT_F_STRICT_TRAVERSAL
        _, strict_stats = ds.render_frame(st, W, H, tile=args.tile, part=rank, n_parts=world, out=image)
        st.flags = flags

    for _ in range(args.warmup):
        step()
    sync()
    exchange_ms[0] = 0.0
    t0 = time.perf_counter()
    kernel_ms = 0.0
    last = None
    for _ in range(args.steps):
        last = step()
        kernel_ms += last.kernel_ms
    sync()
    elapsed = time.perf_counter() - t0
    rays = torch.tensor([float(last.rays), float(last.rays_primary), float(last.rays_traced)], dtype=torch.float64, device=comm)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    frame_rays, frame_primary, frame_traced = (float(x) for x in rays.tolist())
    # every rank's own kernel time and time in the exchange, so that a multi-GPU line explains its own efficiency
    per_rank = torch.zeros((world, 2), dtype=torch.float64, device=comm)
    per_rank[rank, 0] = kernel_ms / args.steps
    per_rank[rank, 1] = exchange_ms[0] / args.steps
    if world > 1:
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    per_rank = per_rank.tolist()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        dof_txt = f"depth of field on (aperture {aperture:g}, focal {focal:g})" if dof else "depth of field off"
        mesh_txt = "Lucy 34k-tri " if scene_name == "c4_lucy" else ""
        out = {
            "metric": f"Mrays/s (primary + reflect/refract + shadow rays the reference traces for the frame, over the frame time; seconds_per_frame beside it), {mesh_txt}{scene_name} {W}x{H} {spp}spp",
            "value": frame_rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "seconds_per_frame": ms_per_step / 1e3,
            "mrays_per_s_traced": frame_traced * args.steps / elapsed / 1e6,
            "mrays_per_s_primary_only": frame_primary * args.steps / elapsed / 1e6,
            "rays_resolved_per_frame": frame_rays,
            "rays_traced_per_frame": frame_traced,
            "value_counts": "rays resolved: every shadow sample the reference traces counts, whether a ray was needed here or the sample was "
                            "decided without one (same image bit for bit); mrays_per_s_traced counts only rays intersected with the scene",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {scene_name} ({describe(hs, abi)}); {W}x{H}, {spp} spp, {dof_txt}, semantics preset {preset}, pruned traversal",
                       "tile": args.tile, "partition": f"tile t -> rank t % {world}, gather to rank 0"},
        }
        if rehearse:
            whole, ws = ds.render_frame(st, W, H)  # the same frame on one rank
            out["rehearsal"] = {"backend": "gloo, all ranks on GPU 0 (timings are not those of a multi-GPU run)",
                                "frame_matches_single_rank": bool((frame[0].numpy() == whole).all()),
                                "rays_match_single_rank": bool(ws.rays == int(frame_rays))}
        picked, _, _ = ds.shared_walk
        n_prims = hs.desc.n_primitives
        cap = 16 if n_prims <= 16 else 64 if n_prims <= 64 else 0
        kname = "render_kernel<false, %d, %s%s>" % (cap, "true" if picked == 1 else "false", ", 4" if ds.waves_per_simd == 4 else "")
        out["config"]["shadow_walk_variant"] = {0: "plain", 1: "shared"}[picked]
        kavg = kernel_ms / args.steps
        build_id = abi.load_library().rt_build_id().decode()
        pmc, pmc_path, stale = profile_counters(args.config, build_id)
        counters = {"rays_traced": last.rays_traced, "nodes_visited": last.nodes_visited, "tris_tested": last.tris_tested, "shaded_hits": last.shaded_hits,
                    "texture_fetches": last.texture_fetches, "n_prims": n_prims, "n_lights": hs.desc.n_lights}
        flops = flops_model(counters)
        needed = 24.0 * last.rays_primary * 2 + 4.0 * last.pixels + 7.0e6  # path colours written, then read by the resolve kernel; the frame; the scene once
        roof = {"bound": "valu_issue", "unit": "T lane-instr/s", "peak": VALU_LANE_OPS_PEAK, "achieved": None, "frac": None, "traffic": None,
                "kernel": kname, "kernel_ms_avg": kavg, "launches_per_frame": 1, "library_build_id": build_id,
                "frac_is": "issue-slot occupancy (every vector instruction priced at 4 cycles, all 64 lanes counted): see rate_weighted_issue_frac, "
                           "lane_utilisation and useful_frac for what is under it",
                "peak_note": f"{SIMDS} SIMDs x 16 lanes x {CLOCK_GHZ} GHz: one 64-lane vector instruction per SIMD per 4 cycles (the FP64 rate; "
                             f"no FMA contraction, so the {FP64_VECTOR_TFLOPS} TFLOP/s FMA figure is out of reach by construction)",
                "flops_model": {"flops_per_launch": flops, "achieved_tflops": flops / (kavg * 1e-3) / 1e12, "peak_tflops": FP64_VECTOR_TFLOPS,
                                "frac": flops / (kavg * 1e-3) / 1e12 / FP64_VECTOR_TFLOPS},
                "hbm": {"bytes_needed_per_launch": needed, "peak_gbps": HBM_PEAK_GBPS}}
        if stale:
            roof["counters_stale"] = True
            roof["counters_from"] = pmc_path
            roof["counters_note"] = "the counter file was collected with another build of the library: run tools/profile_round.sh with this one"
        if world > 1:
            roof["kernel_ms_per_rank"] = {"min": min(r[0] for r in per_rank), "max": max(r[0] for r in per_rank), "all": [r[0] for r in per_rank]}
            roof["exchange_ms"] = per_rank[0][1]
            roof["exchange_ms_per_rank"] = [r[1] for r in per_rank]
            roof["exchange_note"] = ("pack kernel + gather + unpack kernel per frame, timed on each rank's stream" if not rehearse else
                                     "rehearsal: host copies over gloo, wall clock") + "; rank 0's figure includes waiting for the slowest peer"
        if world == 1 and pmc:
            try:
                sq = pmc["pmc_sq"]
                valu = sq["SQ_INSTS_VALU"]
                roof["achieved"] = valu * 64 / (kavg * 1e-3) / 1e12
                roof["frac"] = roof["achieved"] / VALU_LANE_OPS_PEAK
                roof["valu_insts_per_launch"] = valu
                roof["waves_per_simd"] = sq["SQ_WAVES"] / SIMDS
                roof["wave_wait_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
                roof["counters_from"] = pmc_path
            except Exception:
                pass
            try:
                a, b = pmc["pmc_mix_a"], pmc["pmc_mix_b"]
                n_valu = a["SQ_INSTS_VALU"]
                n_f64 = a["SQ_INSTS_VALU_ADD_F64"] + a["SQ_INSTS_VALU_MUL_F64"] + a["SQ_INSTS_VALU_FMA_F64"] + a["SQ_INSTS_VALU_TRANS_F64"]
                lane_util = b["SQ_THREAD_CYCLES_VALU"] / (64.0 * b["SQ_ACTIVE_INST_VALU"])
                f64_ops = (n_f64 + a["SQ_INSTS_VALU_FMA_F64"]) * 64.0 * lane_util  # an FMA is two operations; lanes as active as the average instruction's
                simd_cycles = SIMDS * CLOCK_GHZ * 1e9 * (kavg * 1e-3)
                roof["lane_utilisation"] = lane_util
                roof["fp64_share"] = n_f64 / n_valu
                roof["useful_tflops"] = f64_ops / (kavg * 1e-3) / 1e12
                roof["useful_frac"] = roof["useful_tflops"] / FP64_VECTOR_TFLOPS
                roof["rate_weighted_issue_frac"] = (4.0 * n_f64 + 2.0 * (n_valu - n_f64)) / simd_cycles
                roof["instruction_mix"] = {"fp64": n_f64 / n_valu,
                                           "fp32": (b["SQ_INSTS_VALU_ADD_F32"] + b["SQ_INSTS_VALU_MUL_F32"] + b["SQ_INSTS_VALU_FMA_F32"] + b["SQ_INSTS_VALU_TRANS_F32"]) / n_valu,
                                           "int_and_convert": (a["SQ_INSTS_VALU_INT32"] + a["SQ_INSTS_VALU_INT64"] + a["SQ_INSTS_VALU_CVT"]) / n_valu}
                roof["instruction_mix"]["moves_selects_compares_exchanges"] = 1.0 - sum(roof["instruction_mix"].values())
            except Exception:
                pass
            try:
                traffic = (pmc["pmc_fetch"]["FETCH_SIZE"] + pmc["pmc_write"]["WRITE_SIZE"]) * 1024.0
                roof["traffic"] = traffic
                roof["hbm"].update({"traffic_bytes_per_launch": traffic, "achieved_gbps": traffic / (kavg * 1e-3) / 1e9,
                                    "hbm_traffic_frac": traffic / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic_over_needed": traffic / needed,
                                    "note": "FETCH_SIZE + WRITE_SIZE (KiB -> bytes) of one launch, FETCH_SIZE as reported (reads here are 4- to 16-byte "
                                            "per-lane gathers and scratch reloads; wide streaming reads would be under-counted up to 2x on gfx950)"})
            except Exception:
                pass
        if strict_stats is not None:
            ab = algorithmic_bytes(strict_stats)
            gbps = ab / (strict_stats.kernel_ms * 1e-3) / 1e9
            roof["reference_order_pass"] = {"kernel_ms": strict_stats.kernel_ms, "nodes_visited": strict_stats.nodes_visited,
                                            "tris_tested": strict_stats.tris_tested, "rays": strict_stats.rays,
                                            "algorithmic_bytes": ab, "algorithmic_gbps": gbps, "frac_of_hbm": gbps / HBM_PEAK_GBPS,
                                            "note": "the same frame with the reference's own traversal (every ray traced, no culling): the kernel that does the "
                                                    "work the byte model prices (64 B per node visit, 72 per triangle test, 48 per ray, 128 per shaded hit, 16 per "
                                                    "texture fetch, 4 per pixel); the tree stays in L2 / Infinity Cache, so these bytes are cache traffic, not HBM"}
            roof["survey_model"] = {"algorithmic_bytes": ab, "gbps_at_pruned_kernel_time": ab / (kavg * 1e-3) / 1e9,
                                    "frac_of_hbm": ab / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "note": "above 1: the pruned kernel does not do this work (it walks its own four-wide tree and decides most shadow "
                                            "samples without a ray; same image bit for bit), so the byte model cannot grade it"}
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, st, W, H, budget_s=25.0 if args.config != "c1" else 5.0)
            if args.config == "c1":
                ref = reference_c1()
                if ref is not None:
                    out["cpu_baseline"]["reference_cpu_renderer"] = {"seconds_per_frame": ref["seconds"], "kind": "reference",
                                                                      "what": "oracle/_ref/ref_dump: the reference's own CPU sources, Raytracer::Run (both passes) on this scene, all host cores",
                                                                      "gpu_seconds_per_frame": ms_per_step / 1e3}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
