"""Multi-GPU frame rendering, one process per GPU: tile partition + the one exchange step (a gather to rank 0).

torch.distributed carries the exchange (backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).  The image
is cut into tile x tile squares; tile t (row-major) belongs to rank t % world_size - the same rule `rt_render_frame`
applies on the device side.  Interleaving matters: cost per tile is very uneven (statue vs. empty background),
contiguous strips would not balance.  Pixels are independent and the RNG is keyed by (pixel, sample), so the assembled
frame is bit-identical for every world size.  No other collective is used.

A rank's tiles travel as one packed buffer: its tiles in row-major tile order, each tile row by row.  On the GPU the
library's own copy kernels pack and unpack (rt_pack_tiles / rt_unpack_tiles); `packed_index` is the same order spelled
out in numpy, used for CPU tensors (the gloo tests) and to check the kernels.  The single-process form of the same
exchange (one host program driving all GPUs, peer-to-peer copies, no torch) is rt_render_frame_multi in the C ABI.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import abi


def owned_mask(width, height, tile, part, n_parts):
    """bool [H, W]: pixels of the tiles owned by `part` (mirrors rt_render_frame)."""
    tx = (width + tile - 1) // tile
    ys, xs = np.mgrid[0:height, 0:width]
    t = (ys // tile) * tx + (xs // tile)
    return (t % n_parts) == part


def packed_index(width, height, tile, part, n_parts):
    """int64 [n]: flat pixel index (y * width + x) of every entry of `part`'s packed buffer, in packed order."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    out = []
    for t in range(part, tx * ty, n_parts):
        j, i = divmod(t, tx)
        y0, x0 = j * tile, i * tile
        ys = np.arange(y0, min(y0 + tile, height))[:, None]
        xs = np.arange(x0, min(x0 + tile, width))[None, :]
        out.append((ys * width + xs).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


class FrameGather:
    """Packs a rank's owned tiles, gathers the packs on rank 0 and writes them into the frame."""

    def __init__(self, width, height, tile, rank, world, device):
        self.width, self.height, self.tile, self.rank, self.world = width, height, tile, rank, world
        self.on_gpu = torch.device(device).type == "cuda"
        if self.on_gpu:
            self._lib = abi.load_library()
            self.counts = [int(self._lib.rt_member_pixels(width, height, tile, r, world)) for r in range(world)]
            self._index = None
        else:
            self._lib = None
            self._index = [torch.from_numpy(packed_index(width, height, tile, r, world)) for r in range(world)]
            self.counts = [int(ix.numel()) for ix in self._index]
        self.max_count = max(self.counts)
        # RGBA8 pixels travel as one int32 each; dist.gather wants equal sizes, so every pack has the largest part's size
        self.pack = torch.zeros(self.max_count, dtype=torch.int32, device=device)
        # rank 0 receives into the rows of one buffer, so that one kernel can write all peers' tiles into the frame
        self._landing = torch.zeros((world, self.max_count), dtype=torch.int32, device=device) if rank == 0 else None
        self.recv = [self._landing[r] for r in range(world)] if rank == 0 else None

    def _pack(self, image):
        if self.on_gpu:
            abi.check(self._lib.rt_pack_tiles(image.data_ptr(), self.width, self.height, self.tile, self.rank, self.world, self.pack.data_ptr()))
        else:
            self.pack[: self.counts[self.rank]] = image.view(-1).view(torch.int32)[self._index[self.rank]]

    def _unpack(self, r, image):
        if self.on_gpu:
            abi.check(self._lib.rt_unpack_tiles(self.recv[r].data_ptr(), self.width, self.height, self.tile, r, self.world, image.data_ptr()))
        else:
            image.view(-1).view(torch.int32)[self._index[r]] = self.recv[r][: self.counts[r]]

    def gather(self, image):
        """image: contiguous uint8 [H, W, 4] tensor holding this rank's tiles.  On rank 0 the other ranks' tiles are
        written into it and it is returned complete; other ranks return None.  On the GPU the copy kernels run on the
        stream the library was given (rt_set_stream): give it torch's current stream, which the collective uses too."""
        if self.world == 1:
            return image
        self._pack(image)
        dist.gather(self.pack, self.recv if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        if self.on_gpu:
            abi.check(self._lib.rt_unpack_tiles_all(self.recv[1].data_ptr(), self.max_count * 4, self.width, self.height, self.tile, 1, self.world,
                                                    image.data_ptr()))
        else:
            for r in range(1, self.world):
                self._unpack(r, image)
        return image
