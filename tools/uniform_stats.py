#!/usr/bin/env python3
"""How coherent are the 64 rays of a wave?  Per vote of the box phase: on how many DISTINCT node pairs do the stepping lanes sit
(1 / 2 / 3-4 / more)?  Experiment arms -DRT_EXP_UNIFORM_STATS=1|2 of the tracer (csrc/Makefile librt_amd_exp.so) report the
histogram through the wave-step counters.  Bounds what a wave-uniform (scalar) fetch of shared pairs could save."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
arm = sys.argv[1]
rt.LIB_PATH = os.path.abspath(f"gpu-raytracing_amd/csrc/librt_amd_exp_ustat{arm}.so")
G, W, H = 708, 1920, 1080
tris = scenes.grid_mesh(G, 1)
for name in ("LBVH", "SAH"):
    inp = rt.BuildInput.allocate(tris, sah=name == "SAH")
    (rt.RunSahBuild if name == "SAH" else rt.RunBottomUpBuild)(inp)
    for cname, cam in (("A", scenes.camera_a(G)), ("B", scenes.camera_b(G))):
        frame = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
        c = torch.zeros(4, dtype=torch.int64, device="cuda")
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), rt.to_device(cam), 0, 1 if name == "SAH" else 2, counters=c)
        torch.cuda.synchronize()
        v = c.cpu().numpy()
        print(f"arm {arm} {name} camera {cname}: box tests {v[0]}  buckets {int(v[2])} {int(v[3])}")
