"""Multi-rank frame assembly on CPU (gloo, world_size 2 and 3): tile ownership + the one gather.
Rendering itself needs a GPU; here each rank fills its owned tiles from a known frame, and rank 0
must end up with exactly that frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cuda_raytracing_amd.distributed import FrameGather, owned_mask


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, tile, result):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        frame = torch.from_numpy(rng.integers(0, 256, (h, w, 4), dtype=np.uint8))
        mine = torch.from_numpy(owned_mask(w, h, tile, rank, world))
        image = torch.where(mine[..., None], frame, torch.zeros_like(frame)).contiguous()
        g = FrameGather(w, h, tile, rank, world, torch.device("cpu"))
        out = g.gather(image)
        if rank == 0:
            result.put(bool(torch.equal(out, frame)))
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile", [(2, 100, 70, 16), (3, 64, 48, 32), (2, 33, 17, 8)])
def test_gather_reassembles_the_frame(world, w, h, tile):
    ctx = mp.get_context("spawn")
    result = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, tile, result)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get(timeout=5) is True


def test_ownership_partitions_the_image():
    w, h, tile = 100, 70, 16
    for world in (1, 2, 4, 8):
        total = sum(owned_mask(w, h, tile, r, world).astype(int) for r in range(world))
        assert (total == 1).all()
        counts = [owned_mask(w, h, tile, r, world).sum() for r in range(world)]
        assert max(counts) - min(counts) <= 3 * tile * tile  # interleaving keeps the shares close


def test_packed_order_covers_exactly_the_owned_pixels():
    """A rank's packed buffer lists each of its pixels once, tile after tile; the library's count agrees (no GPU needed)."""
    from cuda_raytracing_amd import abi
    from cuda_raytracing_amd.distributed import packed_index
    lib = abi.load_library()
    for (w, h, tile, world) in [(100, 70, 16, 3), (64, 48, 32, 2), (33, 17, 8, 4), (1920, 1080, 32, 8)]:
        seen = np.zeros(w * h, dtype=np.int32)
        for r in range(world):
            idx = packed_index(w, h, tile, r, world)
            assert idx.size == int(lib.rt_member_pixels(w, h, tile, r, world))
            assert np.array_equal(np.sort(idx), np.flatnonzero(owned_mask(w, h, tile, r, world).reshape(-1)))
            seen[idx] += 1
            # inside a tile the order is row-major: the first tile's first row is contiguous in the image
            first_row = idx[: min(tile, w)]
            assert np.array_equal(first_row, first_row[0] + np.arange(first_row.size))
        assert (seen == 1).all()
    assert lib.rt_member_pixels(64, 48, 32, 2, 2) == -1
