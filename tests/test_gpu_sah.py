"""GPU parity of the SAH build path (rt_run_sah_build; RunSahBuild, BuildWrapper.cu:140-251, no spatial splits) against
the CPU oracle's deterministic restatement: Node[] (top tree slots [0, 128) + cell trees), TrianglePair[], scene bounds
and the grid cell counts are bit-exact; the tree passes the reference's own compiled VerifyHierarchy / CountNodes; and
kDepth frames traced through it equal the frames of the bottom-up tree of the same triangles (same nearest hits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes(scenes):
    return {
        "grid24": scenes.grid_mesh(24, 1),                      # 1152 tris: every root task is small (<= 32 per cell)
        "grid100": scenes.grid_mesh(100, 2),                    # 20000 tris: level loop + small tasks
        "soup65536": scenes.soup(65536, 7),                     # 25 % duplicated boxes: median splits of coincident centroids
        "flat20": scenes.flat_mesh(20, 3),                      # flat axis: NaN cell index path, 16 cells
        "dups4096": np.repeat(scenes.soup(8, 3, dup_fraction=0.0), 512, axis=0),   # 8 distinct boxes x 512: deep median chains
        "one": scenes.grid_mesh(4, 1)[:1],
        "two": scenes.grid_mesh(4, 1)[:2],
        "three": scenes.grid_mesh(4, 1)[:3],
        "odd": scenes.grid_mesh(9, 4)[:161],
        "bigsoup": scenes.soup(20000, 9, size=0.6),             # large triangles: most want a split, the budget runs out
        "line": np.concatenate([scenes.grid_mesh(1, 1)[:1] + np.float32(i) * np.array([1, 0, 0] * 3, np.float32) for i in range(500)]),
        # 59 octaves: trees 45+ levels deep against two level launches -- almost everything is built by sah_finish_kernel
        "fractal": scenes.fractal_corner(3000, 5),
    }


def _gpu_sah(rt, tris, pairs, splits=False):
    import torch
    tri = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = tri.shape[0]
    inp = rt.BuildInput.allocate(tri, sah=True)
    inp.nodes_out.fill_(0xCD)
    inp.triangles_out.fill_(0xCD)
    rt.RunSahBuild(inp, rt.Arguments(build_type=rt.kSAH, enable_pairs=pairs, enable_splits=splits))
    torch.cuda.synchronize()
    lay = rt.sah_scratch_layout(n)
    status = rt.to_host(inp.scratch, np.uint32, 8, lay.status)
    assert status[0] == 0, f"build reported error flags {status[0]:#x}"
    L, R = int(status[1]), int(status[2])
    return dict(inp=inp, n=n, L=L, R=R, nodes=rt.to_host(inp.nodes_out, rt.NODE, 128 + 2 * L),
                leaves=rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, R),
                cells=rt.to_host(inp.scratch, np.uint32, 64, lay.cell_counts))


@pytest.mark.parametrize("splits", [False, True])
@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("name", ["grid24", "grid100", "soup65536", "flat20", "dups4096", "one", "two", "three", "odd", "line", "bigsoup", "fractal"])
def test_sah_build_bit_exact(name, pairs, splits, rt, scenes, ora):
    from helpers import assert_nodes_equal
    tris = _scenes(scenes)[name]
    g = _gpu_sah(rt, tris, pairs, splits)
    o = ora.build_sah(tris, pairs, splits)
    assert g["L"] == o["L"] and g["R"] == o["R"]
    if splits:
        assert g["L"] < g["n"] + g["n"] // 5 + 1
    assert (g["cells"] == o["cell_counts"]).all(), "leaves per grid cell"
    assert g["leaves"].tobytes() == o["leaves"].tobytes(), "TrianglePair[] bytes"
    assert_nodes_equal(g["nodes"], o["nodes"], name)
    L = g["L"]
    if L > 1:   # (numLeafNodes counts references: a split leaf appears once per cell)
        assert ora.count_nodes(g["nodes"], 0, 1) == (2 * L - 1, L, L - 1)
        assert ora.verify_hierarchy(g["nodes"], 0, 1) == 0
        if ora.ref_available():
            assert ora.ref_count_nodes(g["nodes"], 0, 1) == (2 * L - 1, L, L - 1)
            assert ora.ref_verify_hierarchy(g["nodes"], 0, 1) == ""


@pytest.mark.parametrize("pairs", [False, True])
def test_sah_frames_equal_bottom_up_frames(pairs, rt, scenes, ora):
    """Same triangles -> same nearest hit: kDepth through the SAH tree == kDepth through the LBVH, with fewer box tests."""
    from helpers import gpu_build, gpu_trace
    G = 100
    tris = scenes.grid_mesh(G, 2)
    bu = gpu_build(tris)
    sah = _gpu_sah(rt, tris, pairs)
    for cam in (scenes.camera_a(G), scenes.camera_b(G)):
        f0, c0 = gpu_trace(bu, cam, 640, 360, 0)
        f1, c1 = gpu_trace(sah, cam, 640, 360, 0, root=0, count=1)
        if pairs:   # a quad leaf tests its second triangle as (v2, v1, v3): t rounds differently, edge-on rays may flip
            assert (np.abs(f0.astype(int) - f1.astype(int)).max(axis=-1) <= 1).mean() > 0.999
        else:
            assert (f0 == f1).all()
        assert c1[0] < c0[0], "the SAH tree needs fewer box tests"
        o = ora.build_sah(tris, pairs)
        e1, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 640, 360, render_type=0)
        assert (e1 == f1).all() and int(oc[0]) == int(c1[0]) and int(oc[1]) == int(c1[1])


def test_sah_large_structure(rt, scenes, ora):
    """1M triangles: too slow for the oracle's Python round trip to be worth it per test run, so checked through
    size-independent properties -- the reference's compiled checker over the whole tree and every leaf reachable once."""
    tris = scenes.grid_mesh(708, 1)
    g = _gpu_sah(rt, tris, False)
    L = g["L"]
    assert L == tris.shape[0]
    assert ora.count_nodes(g["nodes"], 0, 1) == (2 * L - 1, L, L - 1)
    assert ora.verify_hierarchy(g["nodes"], 0, 1) == 0
    w28 = g["nodes"]["w28"]
    tri_slots = (w28 >> 29) == 2
    ids = np.sort(w28[tri_slots] & 0x1FFFFFFF)
    assert ids.shape[0] == L and (ids == np.arange(L)).all(), "every leaf referenced exactly once"


@pytest.mark.parametrize("pairs", [False, True])
def test_sah_splits_frames(pairs, rt, scenes, ora):
    """--splits: leaves referenced from several cells; same nearest hits as the unsplit trees, GPU == oracle."""
    from helpers import gpu_build, gpu_trace
    tris = scenes.soup(20000, 9, size=0.4)
    bu = gpu_build(tris)
    sah = _gpu_sah(rt, tris, pairs, True)
    assert sah["L"] > sah["R"], "some leaves are referenced more than once"
    cam = scenes.camera_for_box([0, 0, 0], [1, 1, 1])
    f0, _ = gpu_trace(bu, cam, 640, 360, 0)
    f1, c1 = gpu_trace(sah, cam, 640, 360, 0, root=0, count=1)
    assert (f0 == f1).all()
    o = ora.build_sah(tris, pairs, True)
    e1, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 640, 360, render_type=0)
    assert (e1 == f1).all() and int(oc[0]) == int(c1[0]) and int(oc[1]) == int(c1[1])


@pytest.mark.parametrize("count", [1, 2, 3, 5])
def test_sah_tiny_trees_trace(count, rt, scenes, ora):
    """1..5 triangles: the top tree degenerates (a Tri leaf copied into slot 0 for one triangle); frames == oracle == LBVH."""
    from helpers import gpu_build, gpu_trace
    tris = scenes.grid_mesh(3, 2)[:count]
    sah = _gpu_sah(rt, tris, False)
    lo, hi = tris.reshape(-1, 3).min(axis=0), tris.reshape(-1, 3).max(axis=0)
    cam = scenes.make_camera(((lo[0] + hi[0]) / 2, hi[1] + 3.0, (lo[2] + hi[2]) / 2), 0.0, 1.5, 20.0)
    f1, c1 = gpu_trace(sah, cam, 128, 96, 0, root=0, count=1)
    o = ora.build_sah(tris)
    e1, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 128, 96, render_type=0)
    assert (f1 == e1).all() and int(oc[0]) == int(c1[0]) and int(oc[1]) == int(c1[1])
    assert (f1[..., 0] > 0).any(), "the triangles are visible"
    bu = gpu_build(tris)
    f0, _ = gpu_trace(bu, cam, 128, 96, 0)
    assert (f0 == f1).all()


def test_sah_10m_bit_exact(rt, scenes, ora):
    """10,008,338 triangles (BASELINE config 4's scene) through the SAH builder: bit-exact against the oracle."""
    from helpers import assert_nodes_equal
    tris = scenes.grid_mesh(2237, 1)
    g = _gpu_sah(rt, tris, False)
    o = ora.build_sah(tris)
    assert g["L"] == o["L"] == tris.shape[0]
    assert (g["cells"] == o["cell_counts"]).all()
    assert_nodes_equal(g["nodes"], o["nodes"], "10M sah")
    assert g["leaves"].tobytes() == o["leaves"].tobytes()
    L = g["L"]
    assert ora.count_nodes(g["nodes"], 0, 1) == (2 * L - 1, L, L - 1)
    assert ora.verify_hierarchy(g["nodes"], 0, 1) == 0


@pytest.mark.parametrize("pairs,splits", [(False, False), (True, False), (False, True)])
def test_sah_build_stress_back_to_back(pairs, splits, rt, scenes, ora):
    """150 SAH rebuilds of two alternating scenes into the same buffers, two out of three with trace launches of another
    scene in flight on four side streams; node and leaf bytes compared on the GPU after every build (the pattern that
    exposed a store hazard in the LBVH hand-off, tests/test_gpu_parity.py::test_lbvh_build_stress_back_to_back)."""
    import torch
    n = 150000
    load = rt.BuildInput.allocate(scenes.grid_mesh(200, 3))
    rt.RunBottomUpBuild(load)
    cam_d = rt.to_device(scenes.camera_a(200))
    frames = [torch.zeros(1280 * 720 * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    side = [torch.cuda.Stream() for _ in range(4)]
    sets = [scenes.soup(n, 41, dup_fraction=0.2), scenes.grid_mesh(275, 9)[:n]]
    args = rt.Arguments(build_type=rt.kSAH, enable_pairs=pairs, enable_splits=splits)
    oracles = [ora.build_sah(t, pairs, splits) for t in sets]
    dsets = [rt.to_device(s) for s in sets]
    exp_nodes = [torch.from_numpy(o["nodes"][: 128 + 2 * o["L"]].view(np.uint8).reshape(-1).copy()).cuda() for o in oracles]
    exp_leaves = [torch.from_numpy(o["leaves"][: o["R"]].view(np.uint8).reshape(-1).copy()).cuda() for o in oracles]
    inp = rt.BuildInput.allocate(sets[0], sah=True)
    main = torch.cuda.current_stream()
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for it in range(150):
        k = it & 1
        inp.triangles_in.copy_(dsets[k])
        if it % 3 != 2:
            for s_, fr in zip(side, frames):
                s_.wait_stream(main)
                with torch.cuda.stream(s_):
                    for _ in range(3):
                        rt.Trace(load.triangles_out, load.nodes_out, fr, (1280, 720), cam_d, 0, 2)
        rt.RunSahBuild(inp, args)
        got_n = inp.nodes_out.view(torch.uint8).reshape(-1)[: exp_nodes[k].numel()]
        got_l = inp.triangles_out.view(torch.uint8).reshape(-1)[: exp_leaves[k].numel()]
        bad += (got_n != exp_nodes[k]).any().to(torch.int64) + (got_l != exp_leaves[k]).any().to(torch.int64)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} of 150 builds differ from the oracle"
