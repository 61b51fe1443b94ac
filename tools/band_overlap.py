#!/usr/bin/env python3
"""One rank's band of an N-GPU run (rows [0, H/N)) traced repeatedly on ONE GPU, with S streams in flight: does
overlapping consecutive frames hide the per-wave latency tail of a small band?"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
G, W, H = 708, 1920, 1080
inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
rt.RunBottomUpBuild(inp)
cam_d = rt.to_device(scenes.camera_a(G))
torch.cuda.synchronize()
for P in (1, 2, 4, 8):
    rows = (0, H // P)
    for S in (1, 2, 4, 6, 8):
        streams = [torch.cuda.Stream() for _ in range(S)]
        frames = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(S)]
        K = 60
        def run(k):
            for i in range(k):
                s = streams[i % S]
                with torch.cuda.stream(s):
                    rt.Trace(inp.triangles_out, inp.nodes_out, frames[i % S], (W, H), cam_d, 0, 2, rows=rows)
        run(S * 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        print(f"P={P} streams={S}: {dt:.3f} ms per band frame  -> {W * (H // P) / dt / 1e3:.0f} Mrays/s per GPU, x{P} = {W * (H // P) * P / dt / 1e3:.0f}")
