"""rt_render_tile_noalloc (the entry point that allocates no device memory) against rt_render_tile, through the C ABI:
two calls in a row, the second with a smaller input, then the limits."""
import os

import numpy as np
import pytest

import oracle_binding as orc
from cuda_raytracing_amd import abi
from cuda_raytracing_amd.host import HostScene
from helpers import SCENES

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_matches_render_tile():
    import torch
    from cuda_raytracing_amd.device import DeviceScene
    hs = HostScene(os.path.join(SCENES, "c4_lucy.txt"))
    with DeviceScene(hs, device=0) as ds:
        for (W, H, rect, spp) in [(320, 200, (8, 16, 300, 170), 5), (96, 64, (3, 2, 61, 45), 2)]:
            st = orc.settings(spp=spp, dof=True, focal=1.35, aperture=1.0, seed=6, flags=abi.RT_MODE_GPU_REF)
            want = np.full((H, W, 4), 9, dtype=np.uint8)
            _, wstats = ds.render_tile(st, W, H, rect, out=want)
            got = torch.full((H, W, 4), 9, dtype=torch.uint8, device="cuda")
            free_before = torch.cuda.mem_get_info()[0]
            _, gstats = ds.render_tile_noalloc(st, W, H, rect, got)
            free_after = torch.cuda.mem_get_info()[0]
            assert free_after >= free_before, "device memory was allocated"
            assert np.array_equal(got.cpu().numpy(), want)  # bit for bit, pixels outside the rectangle untouched
            # (node / triangle counts are left out: in the shared shadow walk they depend on which lane finds an occluder first)
            for k in ("rays_primary", "rays_secondary", "rays_shadow", "rays_shadow_traced", "shaded_hits", "texture_fetches", "pixels"):
                assert getattr(gstats, k) == getattr(wstats, k), k
        # a call with more paths than the workspace holds at once runs as several launches: 2^19 paths = 8192 pixels at 64 spp
        W, H, rect = 160, 120, (0, 0, 160, 120)
        st = orc.settings(spp=64, dof=True, focal=1.35, aperture=1.0, seed=2, flags=abi.RT_MODE_GPU_REF)
        want, _ = ds.render_tile(st, W, H, rect)
        got = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        ds.render_tile_noalloc(st, W, H, rect, got)
        assert np.array_equal(got.cpu().numpy(), want)
        # limits are errors, not allocations
        big = torch.zeros((1100, 1100, 4), dtype=torch.uint8, device="cuda")
        with pytest.raises(abi.RtError) as e:
            ds.render_tile_noalloc(st, 1100, 1100, (0, 0, 1100, 1100), big)
        assert e.value.code == -5 and "blocks" in str(e.value)
        st.spp = 8193
        with pytest.raises(abi.RtError) as e:
            ds.render_tile_noalloc(st, 96, 64, (0, 0, 8, 8), got)
        assert e.value.code == -5
        st.spp = 1
        with pytest.raises(abi.RtError) as e:
            ds.render_tile_noalloc(st, 96, 64, (0, 0, 8, 8), np.zeros((64, 96, 4), dtype=np.uint8))
        assert "device memory" in str(e.value)


def test_noalloc_calls_from_two_threads_on_one_gpu_take_turns():
    """Two host threads, each with a scene and a stream of its own, call rt_render_tile_noalloc at the same time on one GPU.
    Both work in that GPU's one fixed array; the library's per-device lock makes them take turns, so every image must equal
    what rt_render_tile gives for the same call."""
    import threading

    import torch
    from cuda_raytracing_amd.device import DeviceScene
    jobs = [("c4_lucy", 1.35, [(256, 160, (0, 0, 256, 160), 3), (128, 96, (5, 7, 100, 80), 6), (256, 160, (16, 8, 200, 150), 2)]),
            ("c3_orange", 14.5, [(200, 120, (0, 0, 200, 120), 4), (256, 144, (9, 3, 240, 130), 2), (96, 64, (0, 0, 96, 64), 8)])]
    scenes = [DeviceScene(HostScene(os.path.join(SCENES, name + ".txt")), device=0) for name, _, _ in jobs]
    try:
        want = []
        for ds, (_, focal, calls) in zip(scenes, jobs):
            want.append([ds.render_tile(orc.settings(spp=spp, dof=True, focal=focal, aperture=1.0, seed=4, flags=abi.RT_MODE_GPU_REF), W, H, rect)[0]
                         for (W, H, rect, spp) in calls])
        got = [[None] * len(calls) for _, _, calls in jobs]
        errors = []
        start = threading.Barrier(len(jobs))

        def worker(k):
            try:
                ds, (_, focal, calls) = scenes[k], jobs[k]
                stream = torch.cuda.Stream(device=0)
                ds.set_stream(stream.cuda_stream)  # the render stream belongs to the calling thread
                start.wait()
                for rounds in range(3):  # the same calls three times over, to give the two threads time to collide
                    for i, (W, H, rect, spp) in enumerate(calls):
                        st = orc.settings(spp=spp, dof=True, focal=focal, aperture=1.0, seed=4, flags=abi.RT_MODE_GPU_REF)
                        out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
                        ds.render_tile_noalloc(st, W, H, rect, out)
                        got[k][i] = out.cpu().numpy()
                ds.set_stream(0)
            except Exception as e:  # noqa: BLE001  (reported by the main thread)
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k in range(len(jobs)):
            for i in range(len(jobs[k][2])):
                assert np.array_equal(got[k][i], want[k][i]), (jobs[k][0], i)
    finally:
        for ds in scenes:
            ds.close()
