#!/usr/bin/env python3
"""Per-tile divergence statistics of the bench frame: how many box steps the slowest lane of each 8x8 tile needs
versus the tile's mean (bounds what lane refill could gain).  Uses the internal render type 100 (raw counts)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
G, W, H = 708, 1920, 1080
inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
rt.RunBottomUpBuild(inp)
for name, cam in (("A", scenes.camera_a(G)), ("B", scenes.camera_b(G))):
    frame = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), rt.to_device(cam), 0, 2, render_type=100)
    torch.cuda.synchronize()
    c = frame.cpu().numpy().view(np.uint32).reshape(H, W).astype(np.float64) / 2.0   # pair steps per pixel
    t = c[:H // 8 * 8].reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    print(f"camera {name}: mean steps/lane {t.mean():.1f}  mean of tile max {t.max(1).mean():.1f}  "
          f"=> best-case utilisation without refill {t.mean() / t.max(1).mean():.3f};  p50/p90/p99 tile max "
          f"{np.percentile(t.max(1), 50):.0f}/{np.percentile(t.max(1), 90):.0f}/{np.percentile(t.max(1), 99):.0f}")
