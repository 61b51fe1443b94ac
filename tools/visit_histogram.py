#!/usr/bin/env python3
"""How concentrated are node-pair visits?  (CPU oracle; decides whether an LDS cache of hot pairs could pay.)"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
from oracle import oracle_py as ora
ora.set_threads(8)
ora.lib().ora_set_visit_counts.argtypes = [ctypes.c_void_p]
G = 708
tris = scenes.grid_mesh(G, 1)
b = ora.build_bvh(tris)
for name, cam in (("A", scenes.camera_a(G)), ("B", scenes.camera_b(G))):
    counts = np.zeros(b["nodes"].shape[0], np.uint32)
    ora.lib().ora_set_visit_counts(counts.ctypes.data_as(ctypes.c_void_p))
    ora.trace(b["leaves"], b["nodes"], 0, 2, cam, 1920, 1080)
    ora.lib().ora_set_visit_counts(None)
    c = np.sort(counts[::2].astype(np.int64))[::-1]
    tot = c.sum()
    cs = np.cumsum(c)
    print(f"camera {name}: {tot} pair visits over {int((c > 0).sum())} distinct pairs; share of the K hottest pairs:",
          {k: round(float(cs[k - 1] / tot), 3) for k in (64, 256, 512, 1024, 2048, 4096, 16384)})
