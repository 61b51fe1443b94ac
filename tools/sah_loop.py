#!/usr/bin/env python3
"""Runs the SAH build of the 1M (or --10m) grid mesh a few times: the workload for `rocprofv3 --kernel-trace --stats`."""
import importlib, sys
import torch
sys.path.insert(0, ".")
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
G = 2237 if "--10m" in sys.argv else 708
tris = scenes.grid_mesh(G, 1)
inp = rt.BuildInput.allocate(tris, sah=True)
args = rt.Arguments(build_type=rt.kSAH, enable_pairs="--pairs" in sys.argv)
for _ in range(6):
    rt.RunSahBuild(inp, args)
torch.cuda.synchronize()
