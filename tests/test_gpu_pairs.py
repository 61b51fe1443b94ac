"""GPU tests of --pairs: GenerateMortonCodesPairs + Pairing.cuh + the pair branches of
GenerateTriangles / GenerateAABBs / IntersectRayTrianglePair / RotateAttributes, with deterministic leaf slots
(prefix sum in input order instead of the reference's atomicAdd).  Bit-exact against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build_pairs(rt, tris, hybrid=False):
    import torch
    n = tris.shape[0]
    inp = rt.BuildInput.allocate(tris)
    inp.nodes_out.fill_(0xCD if not hybrid else 0)
    inp.triangles_out.fill_(0xCD)
    rt.RunBottomUpBuild(inp, rt.Arguments(build_type=rt.kHybrid if hybrid else rt.kBottomUp, enable_pairs=True), hybrid=hybrid)
    torch.cuda.synchronize()
    lay = rt.scratch_layout(n)
    L = int(rt.to_host(inp.scratch, np.uint32, 1, lay.num_leaves)[0])
    assert int(rt.to_host(inp.scratch, np.uint32, 1, lay.status)[0]) == 0
    return inp, L, lay


@pytest.mark.parametrize("name", ["grid24", "grid37", "grid300", "soup3000", "odd", "mixed"])
def test_pairs_build_bit_exact(name, rt, scenes, ora):
    from helpers import assert_nodes_equal, gpu_trace
    mixed = np.concatenate([scenes.grid_mesh(9, 2), scenes.soup(101, 3), scenes.grid_mesh(7, 4)[::-1]])
    tris = {"grid24": scenes.grid_mesh(24, 1), "grid37": scenes.grid_mesh(37, 5), "grid300": scenes.grid_mesh(300, 1),
            "soup3000": scenes.soup(3000, 7), "odd": scenes.grid_mesh(5, 1)[:33], "mixed": mixed}[name]
    n = tris.shape[0]
    inp, L, lay = _build_pairs(rt, tris)
    o = ora.build_pairs(tris)
    assert L == o["L"]
    if name.startswith("grid") :
        assert L == n // 2, "every grid cell's two triangles share an edge"
    assert (rt.to_host(inp.scratch, np.uint32, L, lay.morton) == o["codes"]).all()
    assert (rt.to_host(inp.scratch, np.uint32, L, lay.sorted_indices) == o["indices"]).all()
    got = rt.to_host(inp.nodes_out, rt.NODE, 2 * max(L - 1, 1))
    assert_nodes_equal(got, o["nodes"], name + " pairs")
    assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, L).tobytes() == o["leaves"].tobytes()
    assert ora.verify_hierarchy(got, 0, 2) == 0 and ora.count_nodes(got, 0, 2)[1] == L
    if ora.ref_available():
        assert ora.ref_verify_hierarchy(got, 0, 2) == ""
    # tracing a tree with quad leaves: second triangle (v2, v1, v3), ids, rotated attributes
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    mats = scenes.default_materials(3)
    at = scenes.flat_attributes(tris, np.arange(n, dtype=np.int32) % 3)
    light = tuple(float(x) for x in (hi + (hi - lo) * 0.5))
    plain = ora.build_bvh(tris)
    for rtype in (0, 1, 2, 3, 5):
        gi, gc = gpu_trace(dict(inp=inp), cam, 200, 150, rtype, attributes=at, materials=mats, light=light)
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 200, 150, render_type=rtype, attributes=at,
                           materials=mats, light=light)
        d = np.abs(gi.astype(int) - oi.astype(int))
        assert d.max() <= (1 if rtype == 5 else 0), f"render {rtype}"
        assert (gc == oc[:2]).all()
        if rtype in (0, 3):   # same surfaces as the unpaired tree
            pi, _ = ora.trace(plain["leaves"], plain["nodes"], 0, 2, cam, 200, 150, render_type=rtype, attributes=at,
                              materials=mats, light=light)
            assert (gi == pi).all(), "pairing must not change what is hit"


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_pairs_tiny(rt, scenes, ora, n):
    from helpers import gpu_trace
    tris = scenes.grid_mesh(2, 1)[:n]
    inp, L, lay = _build_pairs(rt, tris)
    o = ora.build_pairs(tris)
    assert L == o["L"]
    got = rt.to_host(inp.nodes_out, rt.NODE, 2 * max(L - 1, 1))
    assert got.tobytes() == o["nodes"].tobytes()
    cam = scenes.camera_for_box([0, 0, 0], [2, 2, 2])
    gi, _ = gpu_trace(dict(inp=inp), cam, 64, 48, 0)
    oi, _ = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 64, 48)
    assert (gi == oi).all()


def test_pairs_with_hybrid_top_tree(rt, scenes, ora):
    """hybrid + pairs: the reference roots the trace at 2n+1 although the top tree sits at 2L; here the
    caller reads L from the scratch layout and roots at 2L+1."""
    from helpers import gpu_trace
    tris = scenes.grid_mesh(40, 3)
    inp, L, lay = _build_pairs(rt, tris, hybrid=True)
    assert L == tris.shape[0] // 2
    got = rt.to_host(inp.nodes_out, rt.NODE, 2 * L + 520)
    assert ora.verify_hierarchy(got, 2 * L, 1) == 0 and ora.count_nodes(got, 2 * L, 1)[1] == L
    o = ora.build_pairs(tris)
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    gi, _ = gpu_trace(dict(inp=inp), cam, 160, 120, 0, root=2 * L + 1, count=2)
    oi, _ = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 160, 120)
    assert (gi == oi).all()
