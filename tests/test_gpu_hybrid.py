"""GPU tests of the hybrid build: RunBottomUpBuild(hybrid=true) = LBVH + SAH top tree over the <= 256 sub-roots
(ExtractDepth + SharedTaskBuild, reference BuildWrapper.cu:350-361).

The reference's node numbering for this stage depends on atomic arrival order, so parity with IT is
structural: VerifyHierarchy from the top root (the compiled reference checker too), every leaf reachable, and the
same kDepth frame as the pure bottom-up tree.  Against the oracle's deterministic restatement: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build_hybrid(rt, tris):
    import torch
    inp = rt.BuildInput.allocate(tris)
    inp.nodes_out.fill_(0)
    rt.RunBottomUpBuild(inp, hybrid=True)
    torch.cuda.synchronize()
    return inp


@pytest.mark.parametrize("name", ["grid24", "soup65536", "flat20", "grid37", "grid200"])
def test_hybrid_top_tree(name, rt, scenes, ora):
    from helpers import assert_nodes_equal, gpu_build, gpu_trace
    tris = {"grid24": lambda: scenes.grid_mesh(24, 1), "soup65536": lambda: scenes.soup(65536, 7),
            "flat20": lambda: scenes.flat_mesh(20, 3), "grid37": lambda: scenes.grid_mesh(37, 5),
            "grid200": lambda: scenes.grid_mesh(200, 1)}[name]()
    n = tris.shape[0]
    inp = _build_hybrid(rt, tris)
    o = ora.build_hybrid(tris)
    got = rt.to_host(inp.nodes_out, rt.NODE, o["nodes"].shape[0])
    assert_nodes_equal(got, o["nodes"], name + " hybrid")
    root = 2 * n + 1                                   # main.cu:222
    assert root == o["root"]
    assert ora.verify_hierarchy(got, root - 1, 1) == 0
    cn = ora.count_nodes(got, root - 1, 1)
    assert cn[1] == n, "every leaf reachable exactly once through the top tree"
    if ora.ref_available():
        assert ora.ref_verify_hierarchy(got, root - 1, 1) == ""
        assert ora.ref_count_nodes(got, root - 1, 1) == cn
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    hi_img, hc = gpu_trace(dict(inp=inp), cam, 200, 150, 0, root=root, count=2)
    oi, oc = ora.trace(o["leaves"], o["nodes"], root, 2, cam, 200, 150)
    assert (hi_img == oi).all() and (hc == oc[:2]).all()
    bu_img, _ = gpu_trace(gpu_build(tris), cam, 200, 150, 0)
    assert (hi_img == bu_img).all(), "hybrid and bottom-up trees must render the same depth frame"
    assert (hi_img[..., 0] > 0).sum() > 100


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7])
def test_hybrid_tiny(rt, scenes, ora, n):
    from helpers import gpu_trace
    tris = scenes.soup(max(n, 1), 2, dup_fraction=0.0, size=0.3)[:n]
    inp = _build_hybrid(rt, tris)
    o = ora.build_hybrid(tris)
    cam = scenes.camera_for_box([0, 0, 0], [1, 1, 1])
    gi, gc = gpu_trace(dict(inp=inp), cam, 64, 48, 0, root=2 * n + 1, count=2)
    oi, oc = ora.trace(o["leaves"], o["nodes"], o["root"], 2, cam, 64, 48)
    assert (gi == oi).all()
    if n >= 1:
        got = rt.to_host(inp.nodes_out, rt.NODE, o["nodes"].shape[0])
        assert got.tobytes() == o["nodes"].tobytes()
