#!/usr/bin/env python3
"""Runs N LBVH builds of the bench scene (for profiling): python3 tools/build_loop.py [N] [G]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
G = int(sys.argv[2]) if len(sys.argv) > 2 else 708
inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * N)]
for i in range(N):
    ev[2 * i].record(); rt.RunBottomUpBuild(inp); ev[2 * i + 1].record()
torch.cuda.synchronize()
print("build ms:", [round(ev[2 * i].elapsed_time(ev[2 * i + 1]), 4) for i in range(N)], "n =", inp.num_triangles)
