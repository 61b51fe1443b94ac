"""Shared helpers for the parity tests: run the HIP path through the C ABI and bring results to numpy."""
import importlib

import numpy as np

rt = importlib.import_module("gpu-raytracing_amd")


def gpu_build(tris: np.ndarray):
    """RunBottomUpBuild on the GPU; returns dict of numpy arrays shaped like oracle_py.build_bvh's plus the
    device buffers (for tracing)."""
    import torch
    tri = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    n = tri.shape[0]
    inp = rt.BuildInput.allocate(tri)
    # poison outputs so that any word the build forgets to write shows up in the comparison
    inp.nodes_out.fill_(0xCD)
    inp.triangles_out.fill_(0xCD)
    rt.RunBottomUpBuild(inp)
    torch.cuda.synchronize()
    lay = rt.scratch_layout(n)
    slots = 2 * max(n - 1, 1)
    out = dict(n=n, inp=inp,
               nodes=rt.to_host(inp.nodes_out, rt.NODE, slots),
               leaves=rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, n) if n else np.zeros(0, rt.TRIANGLE_PAIR),
               aabb=rt.to_host(inp.scratch, np.int32, 6, lay.p_aabb),
               status=rt.to_host(inp.scratch, np.uint32, 8, lay.status),
               codes=rt.to_host(inp.scratch, np.uint32, n, lay.morton) if n else np.zeros(0, np.uint32),
               indices=rt.to_host(inp.scratch, np.uint32, n, lay.sorted_indices) if n else np.zeros(0, np.uint32))
    assert out["status"][0] == 0, f"build reported error flags {out['status'][0]:#x}"
    return out


def gpu_trace(build, camera, w, h, render_type=0, attributes=None, materials=None, light=(0, 0, 0), rows=None, spp=1,
              root=0, count=2, textures=None, num_primitives=0):
    import torch
    inp = build["inp"]
    cam_d = rt.to_device(camera)
    rgba = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    at_d = rt.to_device(attributes) if attributes is not None else None
    mt_d = rt.to_device(materials) if materials is not None else None
    rt.Trace(inp.triangles_out, inp.nodes_out, rgba, (w, h), cam_d, root, count, render_type=render_type,
             attributes=at_d, materials=mt_d, num_materials=0 if materials is None else materials.shape[0],
             light=light, counters=counters, rows=rows, spp=spp,
             textures=rt.DeviceTextures(textures) if textures is not None else None, num_primitives=num_primitives)
    torch.cuda.synchronize()
    return rgba.cpu().numpy().reshape(h, w, 4), counters.cpu().numpy().astype(np.uint64)[:2]


def assert_nodes_equal(got: np.ndarray, exp: np.ndarray, what=""):
    """Field-by-field equality of Node arrays: integer words bit-exact, boxes by float value (exact)."""
    assert got.shape == exp.shape, what
    for f in ("w28", "w12"):
        bad = np.nonzero(got[f] != exp[f])[0]
        assert bad.size == 0, f"{what}: {f} differs at {bad[:8]} got {got[f][bad[:4]]} exp {exp[f][bad[:4]]} ({bad.size} slots)"
    for f in ("min", "max"):
        bad = np.nonzero((got[f] != exp[f]).any(axis=1))[0]
        assert bad.size == 0, (f"{what}: {f} differs at {bad[:8]} ({bad.size} slots); got {got[f][bad[:4]].tolist()} "
                               f"exp {exp[f][bad[:4]].tolist()} w28 {[hex(int(x)) for x in got['w28'][bad[:4]]]}")
