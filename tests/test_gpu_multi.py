"""GPU tests of the exchange step and of the several-GPUs-in-one-process frame call (rt_render_frame_multi).
The box has one GPU: the N > 1 code path is exercised by naming that GPU several times (members then share it), which
runs the same partition, pack, peer copy and unpack code as N distinct GPUs would."""
import os

import numpy as np
import pytest

import oracle_binding as orc
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.distributed import owned_mask, packed_index
from cuda_raytracing_amd.host import HostScene
from helpers import SCENES

pytestmark = pytest.mark.gpu


def scene(name):
    return HostScene(os.path.join(SCENES, name + ".txt"))


@pytest.mark.parametrize("W,H,tile,world", [(100, 70, 16, 3), (64, 48, 32, 2), (33, 17, 8, 2), (1920, 1080, 32, 8), (40, 24, 32, 5)])
def test_pack_and_unpack_kernels_match_the_packed_order(W, H, tile, world):
    import torch
    lib = abi.load_library()
    abi.check(lib.rt_init(0))
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 2 ** 31, (H, W), dtype=np.int64).astype(np.int32)
    dev = torch.from_numpy(frame).cuda()
    rebuilt = torch.zeros_like(dev)
    total = 0
    for part in range(world):
        idx = packed_index(W, H, tile, part, world)
        n = int(lib.rt_member_pixels(W, H, tile, part, world))
        assert n == idx.size == int(owned_mask(W, H, tile, part, world).sum())
        total += n
        packed = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
        abi.check(lib.rt_pack_tiles(dev.data_ptr(), W, H, tile, part, world, packed.data_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(packed.cpu().numpy()[:n], frame.reshape(-1)[idx])
        abi.check(lib.rt_unpack_tiles(packed.data_ptr(), W, H, tile, part, world, rebuilt.data_ptr()))
    torch.cuda.synchronize()
    assert total == W * H and torch.equal(rebuilt, dev)
    # all packs but the first part's in the rows of one buffer, written into the frame by one launch
    most = max(int(lib.rt_member_pixels(W, H, tile, part, world)) for part in range(world))
    landing = torch.full((world, most + 3), -1, dtype=torch.int32, device="cuda")
    for part in range(1, world):
        abi.check(lib.rt_pack_tiles(dev.data_ptr(), W, H, tile, part, world, landing[part].data_ptr()))
    again = torch.zeros_like(dev)
    abi.check(lib.rt_pack_tiles(dev.data_ptr(), W, H, tile, 0, world, landing[0].data_ptr()))
    abi.check(lib.rt_unpack_tiles(landing[0].data_ptr(), W, H, tile, 0, world, again.data_ptr()))
    abi.check(lib.rt_unpack_tiles_all(landing[1].data_ptr(), (most + 3) * 4, W, H, tile, 1, world, again.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(again, dev)
    assert lib.rt_unpack_tiles_all(landing[1].data_ptr(), 8, W, H, tile, 1, world, again.data_ptr()) != 0  # stride shorter than a pack


@pytest.mark.parametrize("members", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_frame_of_a_scene_group_equals_the_single_scene_frame(members):
    from cuda_raytracing_amd.device import DeviceScene, SceneGroup
    hs = scene("t_icosa")
    W, H = 200, 136
    st = orc.settings(spp=3, dof=True, focal=9.0, aperture=1.0, seed=4, flags=abi.RT_MODE_GPU_REF)
    with DeviceScene(hs, device=0) as ds:
        ref, rstats = ds.render_frame(st, W, H)
    with SceneGroup(hs, devices=members) as grp:
        assert grp.size == len(members)
        img, stats = grp.render_frame(st, W, H, tile=32)
        again, _ = grp.render_frame(st, W, H, tile=16)
        assert all(ms > 0 for ms in grp.member_ms)
        import torch
        dev = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        grp.render_frame(st, W, H, out=dev)
    assert np.array_equal(img, ref) and np.array_equal(again, ref) and np.array_equal(dev.cpu().numpy(), ref)
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "rays_shadow_traced", "shaded_hits", "texture_fetches", "pixels"):
        assert getattr(stats, k) == getattr(rstats, k), k


def test_group_rejects_gpus_that_do_not_exist():
    from cuda_raytracing_amd.device import SceneGroup
    hs = scene("c1_purple_sphere")
    n = abi.load_library().rt_device_count()
    with pytest.raises(abi.RtError) as e:
        SceneGroup(hs, devices=[0, n])
    assert "does not exist" in str(e.value)


def test_renderer_cli_with_a_gpu_count(tmp_path):
    """rt_render --gpus 1 goes through the plain path; the frame must not depend on it."""
    import subprocess
    from helpers import ROOT
    exe = os.path.join(ROOT, "cuda_raytracing_amd", "rt_render")
    outs = []
    for extra in ([], ["--gpus", "1"]):
        out = tmp_path / f"o{len(outs)}.bmp"
        p = subprocess.run([exe, "--scene", os.path.join(SCENES, "t_icosa.txt"), "--assets", os.path.join(ROOT, "assets"), "--out", str(out),
                            "--size", "96x72", "--spp", "2", "--seed", "1"] + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1]


def test_bench_two_three_and_eight_ranks_rehearsal():
    """`bench.py --gpus N` end to end on the one-GPU box: it starts its own ranks, every rank renders its tiles (on GPU 0),
    the packs are gathered (over gloo, host copies) and rank 0's assembled frame must equal a single-rank render."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = dict(os.environ, RT_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    for n, cfg in ((2, ["--config", "c1"]), (3, ["--config", "headline", "--width", "200", "--height", "136", "--spp", "2"]),
                   (8, ["--config", "headline", "--width", "480", "--height", "272", "--spp", "2"])):  # the rank count of a full node
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                            "--no-strict"] + cfg, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == n and line["value"] > 0
        assert line["rehearsal"]["frame_matches_single_rank"] and line["rehearsal"]["rays_match_single_rank"], line["rehearsal"]
        # a multi-rank line explains its own efficiency: every rank's kernel time and its time in the exchange
        roof = line["roofline"]
        assert len(roof["kernel_ms_per_rank"]["all"]) == n and roof["kernel_ms_per_rank"]["min"] > 0
        assert len(roof["exchange_ms_per_rank"]) == n and roof["exchange_ms"] > 0 and roof["frac"] is None


def test_exchange_calls_over_rccl_with_one_rank(tmp_path):
    """The collectives `bench.py --gpus N` issues, on the real backend ("nccl" = RCCL) with the tensors it passes them: the
    gather of packed int32 tiles on the device, the float64 sum / max reductions, the barrier.  One rank is all a one-GPU box
    can hold (RCCL refuses two ranks on one GPU); it shows that the backend comes up and accepts these calls here."""
    import subprocess
    import sys
    from helpers import ROOT
    script = tmp_path / "one_rank.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from cuda_raytracing_amd.distributed import FrameGather\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "g = FrameGather(100, 70, 16, 0, 2, torch.device('cuda'))  # the buffers of rank 0 of a two-rank job\n"
        "g.pack.copy_(torch.arange(g.max_count, dtype=torch.int32, device='cuda'))\n"
        "dist.gather(g.pack, g.recv[:1], dst=0)\n"
        "t = torch.tensor([3.0, 5.0], dtype=torch.float64, device='cuda')\n"
        "dist.all_reduce(t, op=dist.ReduceOp.SUM); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()\n"
        "assert torch.equal(g.recv[0], g.pack) and t.tolist() == [3.0, 5.0]\n"
        "dist.destroy_process_group(); print('rccl ok')\n")
    from cuda_raytracing_amd.launch import free_port
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert p.returncode == 0 and "rccl ok" in p.stdout, p.stderr[-3000:]
