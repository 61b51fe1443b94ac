#!/usr/bin/env python3
"""Where the LBVH hierarchy kernels spend their time: per-phase timestamps of every workgroup.

Uses the measurement variant of the library (cd gpu-raytracing_amd/csrc && make librt_amd_timing.so), never the shipped
one.  python3 tools/lbvh_phases.py [G]   (G = grid size of the bench mesh, 708 -> 1,002,528 triangles)"""
import ctypes, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("gpu-raytracing_amd")
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 708
rt.LIB_PATH = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_amd_timing.so")
L = rt.lib()
BLOCKS, SLOTS = 32768, 24
inp = rt.BuildInput.allocate(scenes.grid_mesh(G, 1))
for _ in range(3):
    rt.RunBottomUpBuild(inp)
torch.cuda.synchronize()
assert L.rt_debug_lbvh_stamps_clear() == 0
rt.RunBottomUpBuild(inp)
torch.cuda.synchronize()


def stamps(arr):
    buf = np.zeros(BLOCKS * SLOTS, dtype=np.uint64)
    assert L.rt_debug_lbvh_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), arr) == 0
    return buf.reshape(BLOCKS, SLOTS).astype(np.int64)


def us(x):
    return x / 100.0   # 100 MHz


leaf = stamps(0)
used = leaf[:, 0] > 0
t = leaf[used]
t0 = t[:, 0].min()
print(f"n = {inp.num_triangles}; leaf kernel: {used.sum()} workgroups, first start -> last end {us(t[:, 4].max() - t0):.2f} us")
# stamps: 0 start, 1 deltas + locks ready, 5 every gather landed (leaves in LDS), 6 leaf lines issued, 2 climb over,
# 3 node sweep issued, 4 open roots out
names = [("init (deltas, locks)", 0, 1), ("gather -> leaves in LDS", 1, 5), ("leaf lines out", 5, 6), ("climb", 6, 2),
         ("node sweep", 2, 3), ("open roots out", 3, 4)]
for nm, k0, k1 in names:
    d = us(t[:, k1] - t[:, k0])
    print(f"  {nm:24s} avg {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
d = us(t[:, 4] - t[:, 0])
print(f"  {'workgroup total':24s} avg {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
st = np.sort(us(t[:, 0] - t0))
print("  start times (us) percentiles 0/25/50/75/100:", [round(float(np.percentile(st, q)), 2) for q in (0, 25, 50, 75, 100)])

up = stamps(1)
used = up[:, 0] > 0
t = up[used]
t = t[:, :21]
t0 = t[:, 0].min()
print(f"upper kernel: {used.sum()} workgroups; first start -> last stamp {us(t.max() - t0):.2f} us")
lab = ["entry", "L1 prefix", "L1 pass entry", "L1 init | tables + ranges", "L1 climb | nodes", "L1 -", "L1 records out",
       "L2 ticket", "L2 prefix", "L2 pass entry", "L2 init | tables + ranges", "L2 climb | nodes", "L2 -", "L2 records out",
       "L3 ticket", "L3 prefix", "L3 pass entry", "L3 init | tables + ranges", "L3 climb | nodes", "L3 -", "L3 records out"]
for k, nm in enumerate(lab):
    col = t[:, k]
    ok = col > 0
    if ok.any():
        v = us(col[ok] - t0)
        print(f"  {nm:26s} reached by {ok.sum():4d}: at avg {v.mean():7.2f}  min {v.min():7.2f}  max {v.max():7.2f} us after the first workgroup's start")
