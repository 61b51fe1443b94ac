#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle (oracle/liboracle.so).

    python tests/golden/make_golden.py

What they pin: the reference ships no fixtures and its CUDA kernels cannot be built here, so
these vectors are OUTPUTS OF THE ORACLE, frozen.  They catch drift of the oracle itself between machines / compilers
/ libm versions and give the GPU tests fixed expected bytes that do not depend on rebuilding the oracle.  They are
data only (inputs are regenerated from seeds or read from cornell34.obj; expected outputs are arrays and hashes).
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
scenes = importlib.import_module("gpu-raytracing_amd.scenes")
host = importlib.import_module("gpu-raytracing_amd.host_py")
from oracle import oracle_py as ora  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_scenes():
    """name -> (triangles, camera, w, h, attributes, materials, light)"""
    c = host.LoadOBJFromFile(os.path.join(HERE, "cornell34.obj"))
    cam_c = scenes.make_camera((5.0, 5.0, -5.25), 0.0, 0.0, 40.0)       # outside the open side, looking +z
    out = {"cornell34": (c["triangles"], cam_c, 256, 256, c["attributes"], c["materials"], tuple(c["light"]))}
    for name, tris in (("grid24", scenes.grid_mesh(24, 1)), ("soup2048", scenes.soup(2048, 7)),
                       ("flat12", scenes.flat_mesh(12, 3))):
        b = ora.scene_aabb(tris)
        lo, hi = ora.ordered_to_float(b[:3]), ora.ordered_to_float(b[3:])
        cam = scenes.camera_for_box(lo, hi)
        at = scenes.flat_attributes(tris, np.arange(tris.shape[0], dtype=np.int32) % 3)
        out[name] = (tris, cam, 160, 120, at, scenes.default_materials(3), tuple(float(x) for x in hi + (hi - lo) * 0.5))
    return out


def main():
    ora.set_threads(4)
    summary = {}
    for name, (tris, cam, w, h, at, mats, light) in fixture_scenes().items():
        b = ora.build_bvh(tris)
        n = b["n"]
        entry = {"n": n, "aabb_ordered": [int(x) for x in b["aabb"]], "codes_sha256": sha(b["codes"]),
                 "indices_sha256": sha(b["indices"]), "nodes_sha256": sha(b["nodes"]), "leaves_sha256": sha(b["leaves"]),
                 "count_nodes": list(ora.count_nodes(b["nodes"], 0, 2)), "camera_hex": cam.tobytes().hex(),
                 "w": w, "h": h, "light": [float(x) for x in light], "frames": {}}
        for rtype in (0, 1, 2, 3, 5):
            img, cnt = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=rtype, attributes=at,
                                 materials=mats, light=light)
            entry["frames"][str(rtype)] = {"sha256": sha(img), "box_tests": int(cnt[0]), "tri_tests": int(cnt[1]),
                                           "max_stack": int(cnt[2]), "nonblack": int((img[..., :3].max(axis=2) > 0).sum())}
            if name == "cornell34" and rtype in (0, 5):
                np.savez_compressed(os.path.join(HERE, f"cornell34_frame_r{rtype}.npz"), rgba=img)
        if name == "cornell34":   # small enough to keep whole
            np.savez_compressed(os.path.join(HERE, "cornell34_bvh.npz"), codes=b["codes"], indices=b["indices"],
                                nodes=b["nodes"].view(np.uint32).reshape(-1, 8), leaves=b["leaves"].view(np.uint32).reshape(-1, 16))
        summary[name] = entry
    json.dump(summary, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
