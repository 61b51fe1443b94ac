#!/usr/bin/env python3
"""What could a cost-ranked dispatch order of the tile rows buy a SERIAL frame?  

The tuning build's render type 100 gives the raw box-test count of every pixel; a wave (8x8 tile) lasts about as long as
its slowest lane, so cost(tile) = max over its 64 pixels of the pair steps + a constant.  The frame is then list-scheduled
onto the machine's wave slots (256 CUs x 32 waves of this kernel) in dispatch order -- the order the kernel uses today
(tile rows top-down, XCD chunks of 8 workgroups), tile rows sorted by their summed cost (heavy first), and tiles sorted
individually (LPT, the bound of any ordering) -- and the makespans are compared with the two trivial lower bounds (the
longest tile; total work / slots).   RT_LIB=gpu-raytracing_amd/csrc/librt_amd_tuning.so python3 tools/row_order_model.py"""
import heapq, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
rt.LIB_PATH = os.path.abspath(os.environ.get("RT_LIB", "gpu-raytracing_amd/csrc/librt_amd_tuning.so"))
G, W, H, SLOTS, FIXED = 708, 1920, 1080, 256 * 32, 12.0   # FIXED: launch + ray setup + store of a wave, in pair-step units


def makespan(costs):
    slots = [0.0] * SLOTS
    heapq.heapify(slots)
    end = 0.0
    for c in costs:
        t = heapq.heappop(slots) + c
        end = max(end, t)
        heapq.heappush(slots, t)
    return end


inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
rt.RunBottomUpBuild(inp)
for name, cam in (("A", scenes.camera_a(G)), ("B", scenes.camera_b(G))):
    frame = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), rt.to_device(cam), 0, 2, render_type=100)
    torch.cuda.synchronize()
    c = frame.cpu().numpy().view(np.uint32).reshape(H, W).astype(np.float64) / 2.0
    tile = c.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64).max(2) + FIXED   # [row][col]
    rows = tile.sum(1)
    top_down = tile.reshape(-1)
    heavy_rows = tile[np.argsort(-rows)].reshape(-1)
    lpt = np.sort(tile.reshape(-1))[::-1]
    lb = max(tile.max(), tile.sum() / SLOTS)
    m = {k: makespan(v) for k, v in (("top-down (today)", top_down), ("rows heavy-first", heavy_rows), ("tiles LPT (bound)", lpt))}
    print(f"camera {name}: tiles {tile.size}  longest tile {tile.max():.0f}  work/slots {tile.sum() / SLOTS:.0f}  lower bound {lb:.0f} pair steps")
    for k, v in m.items():
        print(f"   {k:20s} makespan {v:8.0f}  = {v / lb:.3f} x lower bound   vs today {m['top-down (today)'] / v:.3f} x")
    print(f"   heaviest tile rows (index: share of the frame's cost): " + ", ".join(f"{i}: {rows[i] / rows.sum():.3f}" for i in np.argsort(-rows)[:6]))
