"""CPU test of the N > 1 path: two processes over gloo render their row bands (with the oracle standing in for the
GPU tracer) and gather them on rank 0 with the same gpu-raytracing_amd/sharding.py code bench.py uses."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_path):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    scenes = importlib.import_module("gpu-raytracing_amd.scenes")
    from oracle import oracle_py as ora
    ora.set_threads(2)
    tris = scenes.grid_mesh(12, 2)
    b = ora.build_bvh(tris)                      # the build is replicated: every rank builds the same tree
    digest = torch.tensor(np.frombuffer(b["nodes"].tobytes(), np.uint8).astype(np.int64).sum())
    other = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(other, digest)
    assert all(int(o) == int(digest) for o in other), "replicated builds differ"
    cam = scenes.camera_b(12)
    y0, y1 = sharding.my_band(h, world, rank)
    img, cnt = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, rows=(y0, y1))
    frame = torch.from_numpy(img.reshape(-1).copy())
    work = sharding.gather_bands(frame, w, h, world, rank, dist, async_op=True)   # bench.py's double-buffered form
    if work is not None:
        work.wait()
    c = torch.tensor([int(cnt[0]), int(cnt[1])])
    dist.all_reduce(c)
    if rank == 0:
        full, fc = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h)
        ok = bool((frame.numpy().reshape(h, w, 4) == full).all()) and [int(c[0]), int(c[1])] == [int(fc[0]), int(fc[1])]
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(64, 48), (50, 37)])   # equal bands (one gather) and ragged bands (point-to-point)
def test_two_rank_band_gather(tmp_path, w, h):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + h
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_band_bounds():
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    for h in (1080, 2160, 37, 7):
        for n in (1, 2, 3, 4, 8):
            b = sharding.band_bounds(h, n)
            assert b[0] == 0 and b[-1] == h and all(b[i] <= b[i + 1] for i in range(n))
            assert max(b[i + 1] - b[i] for i in range(n)) - min(b[i + 1] - b[i] for i in range(n)) <= 1
    assert sharding.band_bounds(1080, 8) == [0, 135, 270, 405, 540, 675, 810, 945, 1080]
