"""CPU test of the N > 1 path: two processes over gloo render their row bands (with the oracle standing in for the
GPU tracer) and gather them on rank 0 with the same gpu-raytracing_amd/sharding.py code bench.py uses."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_path, partition="bands"):
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
    if partition == "bands":
        y0, y1 = sharding.my_band(h, world, rank)
        img, cnt = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, rows=(y0, y1))
        frame = torch.from_numpy(img.reshape(-1).copy())
        work = sharding.gather_bands(frame, w, h, world, rank, dist, async_op=True)   # bench.py's double-buffered form
        if work is not None:
            work.wait()
    else:
        # what rt_trace_strips does: this rank's strips rank, rank + P, ... stored compactly, then one gather of equal
        # pieces and the de-interleave on rank 0
        SR = sharding.STRIP_ROWS
        compact = np.zeros((sharding.compact_rows(h, world), w, 4), np.uint8)
        cnt = np.zeros(2, np.int64)
        for j, s_ in enumerate(sharding.my_strips(h, world, rank)):
            r0, r1 = s_ * SR, min(h, (s_ + 1) * SR)
            img, c1 = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, rows=(r0, r1))
            compact[j * SR:j * SR + (r1 - r0)] = img[r0:r1]
            cnt += np.array([int(c1[0]), int(c1[1])])
        ct = torch.from_numpy(compact.reshape(-1))
        staging = torch.zeros(ct.numel() * world, dtype=torch.uint8) if rank == 0 else None
        work = sharding.gather_strips(ct, staging, world, rank, dist, async_op=True)
        if work is not None:
            work.wait()
        frame = torch.zeros(w * h * 4, dtype=torch.uint8)
        if rank == 0:
            sharding.deinterleave(staging, frame, w, h, world)
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


@pytest.mark.parametrize("w,h", [(64, 48), (50, 37)])   # 6 strips -> 3 + 3;  5 strips (the last one partial) -> 3 + 2
def test_two_rank_strip_gather(tmp_path, w, h):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000) + h
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, w, h, out, "strips"), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_strip_arithmetic_and_deinterleave():
    """Every row of the frame belongs to exactly one (rank, local strip) and deinterleave() puts it back in place."""
    import torch
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    for h in (1080, 2160, 37, 8, 7, 135):
        for world in (1, 2, 3, 4, 8):
            w, SR = 5, sharding.STRIP_ROWS
            owner = np.full(h, -1)
            J = sharding.strips_per_rank(h, world)
            staging = torch.zeros(world * J * SR * w * 4, dtype=torch.uint8)
            sv = staging.view(world, J * SR, w * 4)
            for r in range(world):
                strips = sharding.my_strips(h, world, r)
                assert len(strips) <= J
                for j, s in enumerate(strips):
                    for y in range(s * SR, min(h, (s + 1) * SR)):
                        assert owner[y] == -1
                        owner[y] = r
                        sv[r, j * SR + (y - s * SR)] = y % 251
            assert (owner >= 0).all()
            frame = torch.full((h * w * 4,), 255, dtype=torch.uint8)
            sharding.deinterleave(staging, frame, w, h, world)
            assert (frame.view(h, w * 4)[:, 0].numpy() == np.arange(h) % 251).all(), (h, world)


def test_choose_partition():
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    assert sharding.choose_partition([1.0]) == "bands"
    assert sharding.choose_partition([1.0, 1.05, 0.98, 1.02]) == "bands"          # camera A: max/mean 1.04
    assert sharding.choose_partition([0.3, 0.6, 1.2, 1.9]) == "strips"            # camera B: foreground bands cost more
    assert sharding.choose_partition([1.0, 1.0, 1.0, 1.61]) == "strips"           # 1.61 / 1.1525 = 1.40
    assert sharding.choose_partition([0.0, 0.0]) == "bands"


def test_band_bounds():
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    for h in (1080, 2160, 37, 7):
        for n in (1, 2, 3, 4, 8):
            b = sharding.band_bounds(h, n)
            assert b[0] == 0 and b[-1] == h and all(b[i] <= b[i + 1] for i in range(n))
            assert max(b[i + 1] - b[i] for i in range(n)) - min(b[i + 1] - b[i] for i in range(n)) <= 1
    assert sharding.band_bounds(1080, 8) == [0, 135, 270, 405, 540, 675, 810, 945, 1080]


def test_bench_self_launcher_without_gpu_fails_fast():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes before touching any GPU; on a
    machine without one every rank reports it and the parent returns non-zero promptly (no hang, no orphan ranks)."""
    import subprocess
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("this check is for the GPU-less container")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=180)
    assert p.returncode != 0
    assert "no GPU visible" in p.stderr and "stopping the other ranks" in p.stderr
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
