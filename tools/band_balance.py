#!/usr/bin/env python3
"""Per-band trace time of the bench frame on ONE GPU for P = 2, 4, 8 row bands: predicts multi-GPU load balance."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
sharding = importlib.import_module("gpu-raytracing_amd.sharding")
G, W, H = 708, 1920, 1080
inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
rt.RunBottomUpBuild(inp)
frame = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
for cname, cam in (("A", scenes.camera_a(G)), ("B", scenes.camera_b(G))):
    cam_d = rt.to_device(cam)
    for P in (1, 2, 4, 8):
        ts = []
        for r in range(P):
            y0, y1 = sharding.my_band(H, P, r)
            for _ in range(3):
                rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), cam_d, 0, 2, rows=(y0, y1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), cam_d, 0, 2, rows=(y0, y1))
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print(f"camera {cname} P={P}: band ms {[round(t, 3) for t in ts]}  max/mean {max(ts) / (sum(ts) / P):.2f}  "
              f"speed-up bound {sum(ts[:1]) and (ts and (sum(ts) / max(ts))):.2f} (vs sum of bands)")
