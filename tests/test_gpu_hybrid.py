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


@pytest.mark.parametrize("pairs", [False, True])
def test_hybrid_and_pairs_build_stress_back_to_back(pairs, rt, scenes, ora):
    """300 rebuilds (hybrid top tree; with and without triangle pairs) of two alternating scenes into the same buffers
    without a host synchronisation in between, two out of three with trace launches in flight on side streams; every
    node byte the oracle defines is compared on the GPU (see tests/test_gpu_parity.py::test_lbvh_build_stress_back_to_back)."""
    import torch
    n = 200000
    load = rt.BuildInput.allocate(scenes.grid_mesh(200, 3))
    rt.RunBottomUpBuild(load)
    cam_d = rt.to_device(scenes.camera_a(200))
    frames = [torch.zeros(1280 * 720 * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    side = [torch.cuda.Stream() for _ in range(4)]
    sets = [scenes.soup(n, 51, dup_fraction=0.2), scenes.grid_mesh(317, 4)[:n]]
    args = rt.Arguments(build_type=rt.kHybrid, enable_pairs=pairs)
    dsets = [rt.to_device(s) for s in sets]
    inp = rt.BuildInput.allocate(sets[0])
    main = torch.cuda.current_stream()
    # the first build of each scene is checked against the oracle elsewhere (test_hybrid_top_tree, test_pairs_*); here
    # every later build must reproduce the first one byte for byte (buffers zeroed before each build: slots the build
    # leaves undefined stay zero)
    ref = [None, None]
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for it in range(300):
        k = it & 1
        inp.triangles_in.copy_(dsets[k])
        inp.nodes_out.zero_()
        inp.triangles_out.zero_()
        if it % 3 != 2 and it >= 2:
            for s_, fr in zip(side, frames):
                s_.wait_stream(main)
                with torch.cuda.stream(s_):
                    for _ in range(3):
                        rt.Trace(load.triangles_out, load.nodes_out, fr, (1280, 720), cam_d, 0, 2)
        rt.RunBottomUpBuild(inp, args, hybrid=True)
        if ref[k] is None:
            torch.cuda.synchronize()
            ref[k] = (inp.nodes_out.clone(), inp.triangles_out.clone())
        else:
            bad += (inp.nodes_out != ref[k][0]).any().to(torch.int64) + (inp.triangles_out != ref[k][1]).any().to(torch.int64)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} of 298 builds differ from the first build of their scene"
    # and the first builds are the oracle's trees
    for k in range(2):
        if not pairs:
            o = ora.build_hybrid(sets[k])
            got = np.frombuffer(ref[k][0].cpu().numpy().tobytes(), dtype=rt.NODE)[: o["nodes"].shape[0]]
            from helpers import assert_nodes_equal
            assert_nodes_equal(got, o["nodes"], f"scene {k}")
