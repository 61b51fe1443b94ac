"""Partition of a frame across ranks and the gather of the pieces on rank 0.

The tracer path shards by image rows; the pieces are collected on rank 0 with ONE collective per frame
(torch.distributed gather: RCCL on GPUs, gloo in the CPU tests and in the one-GPU rehearsal).  Two partitions:

* ``bands``  -- rank r of P renders the contiguous rows [r*H/P, (r+1)*H/P) with rt_trace's (y0, y1), in place in its
  own full-frame buffer; the gather receives straight into views of rank 0's frame (no staging copy).
* ``strips`` -- interleaved strips of STRIP_ROWS rows: strip s belongs to rank s mod P (rt_trace_strips renders all of a
  rank's strips in ONE launch, written compactly); rank 0 gathers the compact buffers and de-interleaves them with one
  strided copy.  For cameras whose cost per row is uneven (camera B: the horizon is cheap, the foreground is not) the
  strips balance where bands do not; `choose_partition` picks it when the measured band costs have max/mean > 1.15.

The LBVH build does not shard (one global sort + one tree): it is replicated per rank ("replicas only").
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

STRIP_ROWS = 8          # one tile row of the tracer (a wave is an 8x8 pixel tile)
IMBALANCE_LIMIT = 1.15  # switch to interleaved strips when the bands' max/mean cost exceeds this


# ------------------------------------------------------------------------------------------------ bands
def band_bounds(height: int, world: int) -> List[int]:
    """Row boundaries: rank r owns [b[r], b[r+1]).  Bands differ by at most one row."""
    return [(r * height) // world for r in range(world + 1)]


def my_band(height: int, world: int, rank: int) -> Tuple[int, int]:
    b = band_bounds(height, world)
    return b[rank], b[rank + 1]


# ------------------------------------------------------------------------------------------------ strips
def num_strips(height: int, strip_rows: int = STRIP_ROWS) -> int:
    return -(-height // strip_rows)


def strips_per_rank(height: int, world: int, strip_rows: int = STRIP_ROWS) -> int:
    """Strips in every rank's compact buffer (the same for all ranks so that one gather of equal pieces does it;
    the last ranks' final strip may lie outside the frame and is then left untouched)."""
    return -(-num_strips(height, strip_rows) // world)


def my_strips(height: int, world: int, rank: int, strip_rows: int = STRIP_ROWS) -> List[int]:
    """Global strip indices of `rank`, in the order they are laid out in its compact buffer."""
    return list(range(rank, num_strips(height, strip_rows), world))


def compact_rows(height: int, world: int, strip_rows: int = STRIP_ROWS) -> int:
    return strips_per_rank(height, world, strip_rows) * strip_rows


def deinterleave(staging, frame, width: int, height: int, world: int, strip_rows: int = STRIP_ROWS,
                 bytes_per_pixel: int = 4) -> None:
    """staging = the P compact buffers back to back (rank-major) -> frame rows in image order.  Strip s = j*P + r is
    local strip j of rank r, so viewing staging as [P][J][strip bytes] and swapping the first two axes gives the strips in
    global order; rows beyond `height` (padding of the last strip / of the last ranks) are dropped."""
    J = strips_per_rank(height, world, strip_rows)
    strip_bytes = strip_rows * width * bytes_per_pixel
    ordered = staging[:world * J * strip_bytes].view(world, J, strip_bytes).permute(1, 0, 2).reshape(-1)
    nbytes = height * width * bytes_per_pixel
    frame[:nbytes].copy_(ordered[:nbytes])


def choose_partition(band_costs: Sequence[float], limit: float = IMBALANCE_LIMIT) -> str:
    """'strips' when the per-band costs (one number per rank, any unit) are uneven: max / mean > limit."""
    costs = [float(c) for c in band_costs]
    mean = sum(costs) / max(len(costs), 1)
    if len(costs) < 2 or mean <= 0.0:
        return "bands"
    return "strips" if max(costs) / mean > limit else "bands"


# ------------------------------------------------------------------------------------------------ gather
def _backend_needs_host(frame, dist) -> bool:
    """gloo has no device gather: the one-GPU rehearsal (several ranks sharing a card) stages through the host."""
    return bool(getattr(frame, "is_cuda", False)) and dist.get_backend() == "gloo"


def gather_bands(frame, width: int, height: int, world: int, rank: int, dist=None, bytes_per_pixel: int = 4,
                 async_op: bool = False, force: bool = False):
    """Collect every rank's rows into rank 0's `frame` (a flat uint8 tensor of width*height*bytes_per_pixel bytes,
    each rank having rendered its own rows in place).  Equal bands use one gather into views of rank 0's frame;
    ragged bands (height % world != 0) fall back to point-to-point into place.
    async_op=True (equal bands only) returns the collective's work handle instead of waiting, so the caller can
    trace the next frame into another buffer while this one is in flight.  force=True runs the collective even
    for a single rank (an API check of the backend)."""
    if world == 1 and not force:
        return None
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    b = band_bounds(height, world)
    row = width * bytes_per_pixel
    if _backend_needs_host(frame, dist):
        import torch  # noqa: PLC0415
        torch.cuda.current_stream().synchronize()
        host = frame[b[rank] * row:b[rank + 1] * row].cpu()
        sizes = [(b[r + 1] - b[r]) * row for r in range(world)]
        pad = max(sizes)
        mine = torch.zeros(pad, dtype=torch.uint8)
        mine[:host.numel()] = host
        parts = [torch.empty(pad, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        if rank == 0:
            for r in range(1, world):
                frame[b[r] * row:b[r + 1] * row].copy_(parts[r][:sizes[r]])
        return None
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


def gather_strips(compact, staging, world: int, rank: int, dist=None, async_op: bool = False, force: bool = False):
    """Collect every rank's compact strip buffer (equal sizes) into rank 0's `staging` (world * compact.numel() bytes,
    rank-major).  The caller then runs `deinterleave(staging, frame, ...)` on rank 0 (after waiting on the returned
    work handle when async_op=True)."""
    if world == 1 and not force:
        return None
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    n = compact.numel()
    if _backend_needs_host(compact, dist):
        import torch  # noqa: PLC0415
        torch.cuda.current_stream().synchronize()
        mine = compact.cpu()
        parts = [torch.empty(n, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, parts, dst=0)
        if rank == 0:
            for r in range(world):
                staging[r * n:(r + 1) * n].copy_(parts[r])
        return None
    views = [staging[r * n:(r + 1) * n] for r in range(world)] if rank == 0 else None
    return dist.gather(compact, views, dst=0, async_op=async_op)
