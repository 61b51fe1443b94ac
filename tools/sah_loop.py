#!/usr/bin/env python3
"""Runs the SAH build of the 1M (or --10m) grid mesh a few times: the workload for `rocprofv3 --kernel-trace --stats`, and
(event-timed, printed) for sweeps of an experiment / tuning variant of the library: RT_LIB=<path> (csrc/Makefile)."""
import importlib, os, sys
import torch
sys.path.insert(0, ".")
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
if os.environ.get("RT_LIB"):
    rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB"])
G = 2237 if "--10m" in sys.argv else 708
tris = scenes.grid_mesh(G, 1)
inp = rt.BuildInput.allocate(tris, sah=True)
args = rt.Arguments(build_type=rt.kSAH, enable_pairs="--pairs" in sys.argv)
N = 8
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * N)]
for i in range(N):
    ev[2 * i].record(); rt.RunSahBuild(inp, args); ev[2 * i + 1].record()
torch.cuda.synchronize()
ms = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(2, N))
import numpy as np
status = rt.to_host(inp.scratch, np.uint32, 8, rt.sah_scratch_layout(tris.shape[0]).status)
print(f"sah build G={G}: median {ms[len(ms) // 2]:.4f} ms  min {ms[0]:.4f} ms  status {status[0]:#x}  delta {os.environ.get('RT_SAH_BATCH_DELTA', '0')}")
