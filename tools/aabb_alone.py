#!/usr/bin/env python3
"""scene_aabb_kernel alone, back to back, against the same kernel inside back-to-back builds (tools/build_loop.py):
does its 67 -> 100 us spread at 10M triangles come from the kernel or from what ran before it?
    python3 tools/aabb_alone.py [G]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2237
tris = scenes.grid_mesh(G, 1)
n = tris.shape[0]
inp = rt.BuildInput.allocate(tris)
box = torch.zeros(8, dtype=torch.int32, device="cuda")


def timed(fn, reps):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        out.append(round(e0.elapsed_time(e1) * 1e3, 1))
    return out


rt.CalculateSceneAabb(inp.triangles_in, n, box); torch.cuda.synchronize()
print("n =", n)
print("rt_calculate_scene_aabb alone, back to back (us, incl. its reset launch):", timed(lambda: rt.CalculateSceneAabb(inp.triangles_in, n, box), 10))
# after a kernel that leaves ~1.3 GB of freshly written lines behind (a device fill of the build's output buffers)
def after_writes():
    inp.nodes_out.fill_(1); inp.triangles_out.fill_(1)
vals = []
for _ in range(6):
    after_writes()
    vals += timed(lambda: rt.CalculateSceneAabb(inp.triangles_in, n, box), 1)
print("the same right after 1.9 GB of device fills (nodes_out, triangles_out):", vals)
vals = []
for _ in range(6):
    rt.RunBottomUpBuild(inp)
    vals += timed(lambda: rt.CalculateSceneAabb(inp.triangles_in, n, box), 1)
print("the same right after a full build:", vals)
