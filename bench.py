#!/usr/bin/env python3
"""Headline benchmark: Lucy 34k-triangle scene, 1920x1080, 32 samples per pixel, depth of field on,
rendered through the C ABI on N GPUs of one node (tiles interleaved over ranks, one gather).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one full frame (every rank renders its tiles into device memory, rank 0 gathers).  The
scene is resident in HBM before the timed region; rank 0 prints one JSON line.  `value` is all rays
of the frame (primary + reflect/refract + shadow, counted by the kernel) per second of the slowest
rank.  `roofline` prices the render kernel of rank 0 with the algorithmic-bytes model
64*nodes + 72*tris + 48/ray + 4/pixel + 128/shaded hit + 16/texture fetch, where nodes and tris are
the visit counts of the reference-order (strict) traversal of the same tiles; `traffic` is read from the
PMC summary under profiles/current/ (rocprofv3 cannot run inside this process).  `cpu_baseline` times
the CPU oracle (a port of the reference algorithm, OpenMP) on a bounded sample of the same frame.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (spec)


def algorithmic_bytes(s):
    return (64 * s.nodes_visited + 72 * s.tris_tested + 48 * (s.rays_primary + s.rays_secondary + s.rays_shadow)
            + 4 * s.pixels + 128 * s.shaded_hits + 16 * s.texture_fetches)


def cpu_baseline(hs, st, width, height, budget_s=25.0):
    """Oracle (port of the reference algorithm) on a lattice of 32x32 blocks of the same frame, about budget_s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as orc

    lib = orc.load()
    cores = lib.orc_max_threads()
    nx, ny, b = 16, 12, 32
    rects = [(int((i + 0.5) * width / nx) - b // 2, int((j + 0.5) * height / ny) - b // 2, b, b) for j in range(ny) for i in range(nx)]
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
                      f"({rays} rays in {dt:.1f} s, reference-order traversal, OpenMP over rows)"}


def measured_traffic():
    """HBM-side bytes of one render_kernel<false, 16, ...> launch of this same command, from the rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE, each in a process of its own) that tools/profile_round.sh summarised last."""
    path = os.path.join(ROOT, "profiles", "current", "pmc_render_kernel.json")
    try:
        with open(path) as f:
            j = json.load(f)
        fetch_kib, write_kib = j["pmc_fetch"]["FETCH_SIZE"], j["pmc_write"]["WRITE_SIZE"]
    except Exception:
        return None, None
    note = ("FETCH_SIZE + WRITE_SIZE (KiB -> bytes) of one launch, profiles/current/. FETCH_SIZE is taken as reported: the reads are "
            "4-byte-per-lane scratch reloads and 16-byte-per-lane gathers, widths for which the gfx950 half-count of wide streaming "
            "reads is not calibrated; at most 2x the fetch part. Nearly all of it is register-spill scratch and the path colours handed to the resolve kernel (3.2 GB), not scene data.")
    extra = {}
    try:
        sq = j["pmc_sq"]
        extra = {"valu_busy": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] * (sq["SQ_WAVES"] / (256 * 4)),
                 "wave_wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], "valu_insts": sq["SQ_INSTS_VALU"],
                 "waves_per_simd": sq["SQ_WAVES"] / (256 * 4), "max_waves_per_simd": 8}
    except Exception:
        pass
    return (fetch_kib + write_kib) * 1024.0, {"traffic_note": note, **extra}


def reference_c1(lib_dir):
    """The reference's own CPU renderer (oracle/_ref/ref_dump, both passes of Raytracer::Run) on config 1."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if not os.path.exists(exe):
        return None
    try:
        p = subprocess.run([exe, "run", os.path.join(ROOT, "scenes", "c1_purple_sphere.txt"), "/tmp/_c1_ref.bmp"], cwd=os.path.join(ROOT, "assets"),
                           capture_output=True, text=True, timeout=120)
        line = [l for l in p.stderr.splitlines() if l.startswith("{")][-1]
        return json.loads(line)["seconds"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="c4_lucy")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict", action="store_true", help="skip the reference-order pass that prices the roofline")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    W, H = args.width, args.height
    hs = HostScene(os.path.join(SCENE_DIR, args.scene + ".txt"))
    st = abi.Settings(dof_enabled=1, spp=args.spp, focal_distance=1.35, aperture=1.0, seed=1, flags=abi.RT_MODE_GPU_REF)
    ds = DeviceScene(hs, device=local)
    stream = torch.cuda.current_stream()
    ds.set_stream(stream.cuda_stream)
    image = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    gather = FrameGather(W, H, args.tile, rank, world, image.device)

    def step():
        _, s = ds.render_frame(st, W, H, tile=args.tile, part=rank, n_parts=world, out=image)
        gather.gather(image)
        return s

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # reference-order traversal of this rank's tiles: the visit counts behind the algorithmic bytes
    strict_stats = None
    if rank == 0 and not args.no_strict:
        st.flags = abi.RT_MODE_GPU_REF | abi.RT_F_STRICT_TRAVERSAL
        _, strict_stats = ds.render_frame(st, W, H, tile=args.tile, part=rank, n_parts=world, out=image)
        st.flags = abi.RT_MODE_GPU_REF

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    last = None
    for _ in range(args.steps):
        last = step()
        kernel_ms += last.kernel_ms
    sync()
    elapsed = time.perf_counter() - t0
    rays = torch.tensor([float(last.rays), float(last.rays_primary)], dtype=torch.float64, device="cuda")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    frame_rays, frame_primary = float(rays[0].item()), float(rays[1].item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary + reflect/refract + shadow rays), Lucy 34k-tri 1920x1080 32spp",
            "value": frame_rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "seconds_per_frame": ms_per_step / 1e3,
            "mrays_per_s_primary_only": frame_primary * args.steps / elapsed / 1e6,
            "rays_per_frame": frame_rays,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.scene}: Lucy 33,446-triangle mesh + 4 spheres, plane, cylinder, Bezier vase, 2 sphere lamps; "
                                   f"{W}x{H}, {args.spp} spp, depth of field on (aperture 1, focal 1.35), semantics preset gpu_ref, pruned traversal",
                       "tile": args.tile, "partition": f"tile t -> rank t % {world}, gather to rank 0"},
        }
        picked, plain_ms, shared_ms = ds.shared_walk
        out["config"]["shadow_walk_variant"] = {"picked": {0: "plain", 1: "shared"}.get(picked, "plain (not timed)"),
                                                "sample_ms_plain": plain_ms, "sample_ms_shared": shared_ms}
        kavg = kernel_ms / args.steps
        if strict_stats is not None:
            ab = algorithmic_bytes(strict_stats)
            achieved = ab / (kavg * 1e-3) / 1e9
            touched = 128 * last.nodes_visited + 96 * last.tris_tested  # four-wide nodes and leaf records the pruned walk loaded
            traffic, tnote = measured_traffic()
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                               "traffic": traffic, "kernel": "render_kernel<false, 16, %s>" % ("true" if picked == 1 else "false"), "kernel_ms_avg": kavg,
                               "algorithmic_bytes_per_launch": ab,
                               "bvh_bytes_requested_per_launch": touched,
                               "strict_kernel_ms": strict_stats.kernel_ms,
                               "strict_achieved": ab / (strict_stats.kernel_ms * 1e-3) / 1e9,
                               "note": "achieved = algorithmic bytes of the reference-order traversal / pruned kernel time (rank 0's launch). "
                                       "The pruned walk visits a small fraction of those nodes and the scene (a few MB) stays in L2, so the "
                                       "figure exceeds the HBM peak: the kernel is bound by double-precision VALU issue and dependent-load "
                                       "latency, not by HBM (see valu_busy / wave_wait_frac)"}
            if tnote:
                out["roofline"].update(tnote)
        else:
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                               "kernel": "render_kernel<false, 16, %s>" % ("true" if picked == 1 else "false"), "kernel_ms_avg": kavg}
        if world == 1 and not args.no_cpu_baseline:
            st.flags = abi.RT_MODE_GPU_REF
            out["cpu_baseline"] = cpu_baseline(hs, st, W, H)
            ref_s = reference_c1(ROOT)
            if ref_s is not None:
                out["cpu_baseline"]["reference_cpu_renderer_config1_seconds"] = ref_s
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
