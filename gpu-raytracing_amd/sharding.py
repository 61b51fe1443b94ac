"""Row-band sharding of a frame across ranks and the gather of the bands on rank 0.

The tracer path shards by image rows: rank r of P renders rows [r*H/P, (r+1)*H/P) with rt_trace's (y0, y1) and the
bands are collected on rank 0 with ONE collective (torch.distributed gather: RCCL on GPUs, gloo in the CPU tests).
The LBVH build does not shard (one global sort + one tree): it is replicated per rank ("replicas only").
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def band_bounds(height: int, world: int) -> List[int]:
    """Row boundaries: rank r owns [b[r], b[r+1]).  Bands differ by at most one row."""
    return [(r * height) // world for r in range(world + 1)]


def my_band(height: int, world: int, rank: int) -> Tuple[int, int]:
    b = band_bounds(height, world)
    return b[rank], b[rank + 1]


def gather_bands(frame, width: int, height: int, world: int, rank: int, dist=None, bytes_per_pixel: int = 4,
                 async_op: bool = False):
    """Collect every rank's rows into rank 0's `frame` (a flat uint8 tensor of width*height*bytes_per_pixel bytes,
    each rank having rendered its own rows in place).  Equal bands use one gather into views of rank 0's frame;
    ragged bands (height % world != 0) fall back to point-to-point into place.
    async_op=True (equal bands only) returns the collective's work handle instead of waiting, so the caller can
    trace the next frame into another buffer while this one is in flight."""
    if world == 1:
        return None
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    b = band_bounds(height, world)
    row = width * bytes_per_pixel
    band = frame[b[rank] * row:b[rank + 1] * row]
    views: Optional[list] = None
    if rank == 0:
        views = [frame[b[r] * row:b[r + 1] * row] for r in range(world)]
    if len({b[r + 1] - b[r] for r in range(world)}) == 1:
        return dist.gather(band, views, dst=0, async_op=async_op)
    if rank == 0:
        reqs = [dist.irecv(views[r], src=r) for r in range(1, world)]
        for q in reqs:
            q.wait()
    else:
        dist.send(band, dst=0)
    return None
