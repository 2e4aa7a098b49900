"""Multi-GPU frame rendering: tile partition + the one exchange step (a gather to rank 0).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU
tests).  The image is cut into tile x tile squares; tile t (row-major) belongs to rank
t % world_size - the same rule `rt_render_frame` applies on the device side.  Interleaving matters:
cost per tile is very uneven (statue vs. empty background), contiguous strips would not balance.
Pixels are independent and the RNG is keyed by (pixel, sample), so the assembled frame is
bit-identical for every world size.  No other collective is used.
"""
import numpy as np
import torch
import torch.distributed as dist


def owned_mask(width, height, tile, part, n_parts):
    """bool [H, W]: pixels of the tiles owned by `part` (mirrors rt_render_frame)."""
    tx = (width + tile - 1) // tile
    ys, xs = np.mgrid[0:height, 0:width]
    t = (ys // tile) * tx + (xs // tile)
    return (t % n_parts) == part


class FrameGather:
    """Packs a rank's owned pixels, gathers the packs on rank 0 and scatters them into the frame."""

    def __init__(self, width, height, tile, rank, world, device):
        self.width, self.height, self.rank, self.world = width, height, rank, world
        masks = [owned_mask(width, height, tile, r, world).reshape(-1) for r in range(world)]
        self.counts = [int(m.sum()) for m in masks]
        self.max_count = max(self.counts)
        self.my_index = torch.from_numpy(np.flatnonzero(masks[rank])).to(device)
        self.all_index = [torch.from_numpy(np.flatnonzero(m)).to(device) for m in masks] if rank == 0 else None
        # RGBA8 pixels travel as one int32 each
        self.pack = torch.zeros(self.max_count, dtype=torch.int32, device=device)
        self.recv = [torch.zeros(self.max_count, dtype=torch.int32, device=device) for _ in range(world)] if rank == 0 else None

    def gather(self, image):
        """image: uint8 [H, W, 4] tensor holding this rank's tiles.  On rank 0 the other ranks'
        tiles are written into it and it is returned complete; other ranks return None."""
        flat = image.view(-1).view(torch.int32)  # [H*W]
        if self.world == 1:
            return image
        self.pack[: self.counts[self.rank]] = flat[self.my_index]
        dist.gather(self.pack, self.recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        for r in range(1, self.world):
            flat[self.all_index[r]] = self.recv[r][: self.counts[r]]
        return image
