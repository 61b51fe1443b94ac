"""GPU parity on the traversal paths no ordinary scene reaches (Tracer.cu:187-200, 308-374 vs csrc/trace_kernel.hip):

* the PRIVATE half of the traversal stack (entries 16..63; the first 16 live in an LDS column) -- the fixed test scenes
  peak at 13 entries, the 1M bench frame at 10; scenes.fractal_corner reaches 28 (LBVH) / 48 (SAH), asserted here;
* a FULL stack: pushes onto 64 entries are dropped, the nearest child's push included (oracle/rt_oracle.c PUSH);
* the second traversal of kTextureLitShadows re-using the stack after a deep primary traversal;
* nodes wider than a pair (advance()'s count > 2 branch, odd counts -> the lone-slot step);
* -0.0 / +0.0 coordinates, zero-area and repeated-vertex triangles through every builder;
* rays with a direction component of exactly 0 starting exactly on node-box planes (0 * inf = NaN in the slab test).

Every comparison is GPU == oracle: frame bytes and sum(box tests) / sum(triangle tests).  (The oracle itself is checked on
these scenes by tests/test_oracle_edges.py.)"""
import numpy as np
import pytest

import edge_scenes

pytestmark = pytest.mark.gpu


def _upload_tree(rt, nodes, leaves):
    """A hand-made Node[] / TrianglePair[] as device buffers (what rt_trace takes: any tree in the reference layout)."""
    class _Inp:
        pass
    inp = _Inp()
    inp.nodes_out = rt.to_device(nodes)
    inp.triangles_out = rt.to_device(leaves)
    return dict(inp=inp)


def _gpu_sah(rt, tris, pairs=False):
    import torch
    inp = rt.BuildInput.allocate(tris, sah=True)
    rt.RunSahBuild(inp, rt.Arguments(build_type=rt.kSAH, enable_pairs=pairs))
    torch.cuda.synchronize()
    status = rt.to_host(inp.scratch, np.uint32, 8, rt.sah_scratch_layout(tris.shape[0]).status)
    assert status[0] == 0
    return dict(inp=inp, L=int(status[1]), R=int(status[2]))


def _gpu_bu(rt, tris, hybrid=False, pairs=False):
    import torch
    inp = rt.BuildInput.allocate(tris)
    inp.nodes_out.fill_(0)
    rt.RunBottomUpBuild(inp, rt.Arguments(build_type=rt.kHybrid if hybrid else rt.kBottomUp, enable_pairs=pairs), hybrid=hybrid)
    torch.cuda.synchronize()
    lay = rt.scratch_layout(tris.shape[0])
    status = rt.to_host(inp.scratch, np.uint32, 8, lay.status)
    assert status[0] == 0
    return dict(inp=inp, L=int(status[1]))


@pytest.fixture(scope="module")
def fractal(rt, scenes, ora):
    from helpers import assert_nodes_equal
    tris = scenes.fractal_corner(4000, 3)
    n = tris.shape[0]
    out = dict(tris=tris, cam=scenes.diagonal_camera(2.0 ** -10, 2.0 ** 45))
    o = ora.build_bvh(tris)
    g = _gpu_bu(rt, tris)
    assert_nodes_equal(rt.to_host(g["inp"].nodes_out, rt.NODE, 2 * (n - 1)), o["nodes"], "fractal LBVH")
    out["lbvh"] = (g, o, 0, 2)
    o = ora.build_hybrid(tris)
    g = _gpu_bu(rt, tris, hybrid=True)
    assert_nodes_equal(rt.to_host(g["inp"].nodes_out, rt.NODE, o["nodes"].shape[0]), o["nodes"], "fractal hybrid")
    out["hybrid"] = (g, o, o["root"], 2)
    o = ora.build_sah(tris)
    g = _gpu_sah(rt, tris)
    assert_nodes_equal(rt.to_host(g["inp"].nodes_out, rt.NODE, 128 + 2 * g["L"]), o["nodes"], "fractal SAH")
    out["sah"] = (g, o, 0, 1)
    return out


@pytest.mark.parametrize("tree,min_depth", [("lbvh", 24), ("hybrid", 24), ("sah", 40)])
@pytest.mark.parametrize("render_type", [0, 1, 2])
def test_deep_stack_private_spill_path(fractal, ora, tree, min_depth, render_type):
    from helpers import gpu_trace
    g, o, root, count = fractal[tree]
    for (w, h) in ((33, 25), (96, 64)):
        oi, oc = ora.trace(o["leaves"], o["nodes"], root, count, fractal["cam"], w, h, render_type=render_type)
        assert oc[2] >= min_depth and oc[3] == 0, f"oracle max_stack {oc[2]}: the scene must overflow the 16 LDS entries"
        gi, gc = gpu_trace(g, fractal["cam"], w, h, render_type, root=root, count=count)
        assert (gc == oc[:2]).all(), f"{tree}: counters {gc} vs {oc}"
        assert (gi == oi).all(), f"{tree}: {(gi != oi).any(axis=2).sum()} pixels differ"


@pytest.mark.parametrize("tree", ["lbvh", "sah"])
def test_deep_stack_shaded_and_shadow_rays(fractal, scenes, ora, tree):
    """kDiffuse and kTextureLitShadows (a second wave-wide trace_ray over the lanes that hit, re-using the stack column and
    the private array) on the deep scene."""
    from helpers import gpu_trace
    g, o, root, count = fractal[tree]
    tris = fractal["tris"]
    n = tris.shape[0]
    mats = scenes.default_materials(3)
    at = scenes.flat_attributes(tris, np.arange(n, dtype=np.int32) % 3)
    light = (-2.0 ** 44, 2.0 ** 45, -2.0 ** 43)     # outside the scene (nearly every hit is shadowed by the nested geometry)
    for render_type in (5, 8):
        kw = dict(attributes=at, materials=mats, light=light)
        oi, oc = ora.trace(o["leaves"], o["nodes"], root, count, fractal["cam"], 96, 64, render_type=render_type, **kw)
        assert oc[2] >= 24
        gi, gc = gpu_trace(g, fractal["cam"], 96, 64, render_type, root=root, count=count, **kw)
        assert (gc == oc[:2]).all()
        assert (gi == oi).all(), f"{tree} render {render_type}: {(gi != oi).any(axis=2).sum()} pixels differ"
        assert (oi[..., :3].max(axis=2) > 0).mean() > 0.2, "the frame is shaded"


def test_full_stack_drops_pushes(rt, scenes, ora):
    """140 octaves under the binned SAH: the centre ray fills all 64 entries and later pushes are dropped -- identically on
    both sides (the reference writes past its array there: undefined, Tracer.cu:353-369)."""
    from helpers import gpu_trace, assert_nodes_equal
    tris = scenes.fractal_corner(8000, 3, octaves=140, top_exp=42)
    o = ora.build_sah(tris)
    g = _gpu_sah(rt, tris)
    assert_nodes_equal(rt.to_host(g["inp"].nodes_out, rt.NODE, 128 + 2 * g["L"]), o["nodes"], "140 octaves SAH")
    cam = scenes.diagonal_camera(2.0 ** -10, 2.0 ** 45)
    for render_type in (0, 1, 2):
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 33, 25, render_type=render_type)
        assert oc[2] == 64 and oc[3] > 0, oc
        gi, gc = gpu_trace(g, cam, 33, 25, render_type, root=0, count=1)
        assert (gc == oc[:2]).all(), f"counters {gc} vs {oc}"
        assert (gi == oi).all()


@pytest.mark.parametrize("width", [3, 4, 5, 7])
def test_nodes_wider_than_a_pair(rt, scenes, ora, width):
    """count > 2: the tracer walks such a node two slots at a time carrying the nearest child across steps; an odd count
    ends in a lone-slot step.  Hand-packed trees through rt_trace (no builder of the reference or of this library
    emits them, TraceRay accepts them: Tracer.cu:323)."""
    from helpers import gpu_trace
    tris = scenes.soup(1500, 5, dup_fraction=0.2, size=0.2)
    b = ora.build_bvh(tris)
    lo, hi = ora.ordered_to_float(b["aabb"][:3]), ora.ordered_to_float(b["aabb"][3:])
    nodes, root, count = edge_scenes.collapse_wide(b["nodes"], 0, 2, width, rt.NODE)
    g = _upload_tree(rt, nodes, b["leaves"])
    for cam in (scenes.camera_for_box(lo, hi), scenes.camera_for_box(lo, hi, yaw=-2.1, pitch=0.9, back=0.8)):
        for render_type in (0, 1, 2):
            oi, oc = ora.trace(b["leaves"], nodes, root, count, cam, 160, 100, render_type=render_type)
            gi, gc = gpu_trace(g, cam, 160, 100, render_type, root=root, count=count)
            assert (gc == oc[:2]).all(), f"width {width}: counters {gc} vs {oc}"
            assert (gi == oi).all()


@pytest.mark.parametrize("variant", ["bottom-up", "pairs", "hybrid", "sah", "sah+pairs"])
def test_signed_zero_and_degenerate_triangles(rt, scenes, ora, variant):
    from helpers import assert_nodes_equal, gpu_trace
    tris = edge_scenes.signed_zero_mesh(scenes)
    n = tris.shape[0]
    pairs = "pairs" in variant
    if variant.startswith("sah"):
        o = ora.build_sah(tris, pairs=pairs)
        g = _gpu_sah(rt, tris, pairs)
        assert g["L"] == o["L"] and g["R"] == o["R"]
        slots, nleaves, root, count = 128 + 2 * g["L"], g["R"], 0, 1
    else:
        hybrid = variant == "hybrid"
        o = ora.build_pairs(tris) if pairs else (ora.build_hybrid(tris) if hybrid else ora.build_bvh(tris))
        g = _gpu_bu(rt, tris, hybrid=hybrid, pairs=pairs)
        slots, nleaves = o["nodes"].shape[0], o["leaves"].shape[0]
        assert g["L"] == nleaves
        root, count = (o["root"], 2) if hybrid else (0, 2)
    assert_nodes_equal(rt.to_host(g["inp"].nodes_out, rt.NODE, slots), o["nodes"], variant)
    assert rt.to_host(g["inp"].triangles_out, rt.TRIANGLE_PAIR, nleaves).tobytes() == o["leaves"].tobytes(), "leaf bytes (signs of zero included)"
    if not variant.startswith("sah"):
        lay = rt.scratch_layout(n)
        assert (rt.to_host(g["inp"].scratch, np.int32, 6, lay.p_aabb) == o["aabb"]).all(), "ordered-int scene box: -0.0 < +0.0"
    mats = scenes.default_materials(3)
    at = scenes.flat_attributes(tris, np.arange(n, dtype=np.int32) % 3)      # degenerate triangles: NaN normals, never hit
    for cam in (scenes.make_camera((0.0, 6.0, 0.0), 0.3, 1.2, 60.0),         # origin components exactly +0
                scenes.make_camera((-0.0, 3.0, -12.0), 0.0, 0.2, 60.0)):
        for render_type in (0, 1, 2, 5):
            kw = dict(attributes=at, materials=mats, light=(3.0, 9.0, -4.0)) if render_type == 5 else {}
            oi, oc = ora.trace(o["leaves"], o["nodes"], root, count, cam, 160, 96, render_type=render_type, **kw)
            gi, gc = gpu_trace(g, cam, 160, 96, render_type, root=root, count=count, **kw)
            assert (gc == oc[:2]).all(), f"{variant} render {render_type}: counters {gc} vs {oc}"
            assert (gi == oi).all(), f"{variant} render {render_type}: {(gi != oi).any(axis=2).sum()} pixels differ"


@pytest.mark.parametrize("tree", ["bottom-up", "sah"])
def test_axis_parallel_rays_on_box_planes(rt, scenes, ora, tree):
    """yaw = pitch = 0, odd width and height, camera exactly on integer x / z planes of the grid's node boxes and on a leaf
    box's y plane: the centre column / row run the slab test with 1 / direction = inf and (plane - origin) = 0, i.e. NaN.
    v_min_f32 / v_max_f32 (IEEE mode) and C's fminf / fmaxf both return the other operand."""
    from helpers import gpu_trace
    G = 16
    tris = scenes.grid_mesh(G, 2)
    t3 = tris.reshape(-1, 3, 3)
    ys_all = np.sort(t3[:, :, 1].reshape(-1))
    y_plane = float(ys_all[ys_all.size // 2])
    if tree == "sah":
        o, g, root, count = ora.build_sah(tris), _gpu_sah(rt, tris), 0, 1
    else:
        o, g, root, count = ora.build_bvh(tris), _gpu_bu(rt, tris), 0, 2
    for pos in ((G // 2, y_plane, -3.0), (G // 2, y_plane, 4.0), (3.0, 1.0, G // 2)):
        cam = edge_scenes.axis_camera(scenes, pos, 64.0)
        for (w, h) in ((65, 49), (129, 97)):
            for render_type in (0, 1, 2):
                oi, oc = ora.trace(o["leaves"], o["nodes"], root, count, cam, w, h, render_type=render_type)
                gi, gc = gpu_trace(g, cam, w, h, render_type, root=root, count=count)
                assert (gc == oc[:2]).all(), f"{tree} {pos} {w}x{h} render {render_type}: counters {gc} vs {oc}"
                assert (gi == oi).all(), f"{tree} {pos}: {(gi != oi).any(axis=2).sum()} pixels differ"


BIG = 10_000_000   # a scene-size hint (DeviceScene::num_attributes) that selects the pair-prefetch instantiations


@pytest.mark.parametrize("tree", ["lbvh", "sah"])
def test_prefetch_instantiation_deep_stack_all_render_types(fractal, scenes, ora, tree):
    """trace_kernel<RENDER, PF = true> (picked for scenes of >= 8M primitives: the next pair's loads are issued right after
    advance() and carried in registers across the wave's vote) must walk exactly like the default instantiation: the deep
    scene, every untextured render type, frames and counters against the oracle."""
    from helpers import gpu_trace
    g, o, root, count = fractal[tree]
    tris = fractal["tris"]
    n = tris.shape[0]
    mats = scenes.default_materials(3)
    at = scenes.flat_attributes(tris, np.arange(n, dtype=np.int32) % 3)
    kw = dict(attributes=at, materials=mats, light=(-2.0 ** 44, 2.0 ** 45, -2.0 ** 43))
    for render_type in (0, 1, 2, 3, 5, 8):
        oi, oc = ora.trace(o["leaves"], o["nodes"], root, count, fractal["cam"], 96, 64, render_type=render_type, **kw)
        gi, gc = gpu_trace(g, fractal["cam"], 96, 64, render_type, root=root, count=count, num_primitives=BIG, **kw)
        assert (gc == oc[:2]).all(), f"{tree} render {render_type}: counters {gc} vs {oc}"
        assert (gi == oi).all(), f"{tree} render {render_type}: {(gi != oi).any(axis=2).sum()} pixels differ"


def test_prefetch_instantiation_full_stack_wide_nodes_and_bands(rt, scenes, ora):
    """The same instantiation on the full-stack scene (dropped pushes), on nodes wider than a pair (the prefetch must take
    the lone-slot form of a node's last step) and on row bands / 16 spp (ragged tiles, inactive lanes)."""
    from helpers import gpu_trace
    tris = scenes.fractal_corner(8000, 3, octaves=140, top_exp=42)
    o = ora.build_sah(tris)
    g = _gpu_sah(rt, tris)
    cam = scenes.diagonal_camera(2.0 ** -10, 2.0 ** 45)
    oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 33, 25, render_type=1)
    assert oc[3] > 0
    gi, gc = gpu_trace(g, cam, 33, 25, 1, root=0, count=1, num_primitives=BIG)
    assert (gc == oc[:2]).all() and (gi == oi).all()
    tris = scenes.soup(1500, 5, dup_fraction=0.2, size=0.2)
    b = ora.build_bvh(tris)
    lo, hi = ora.ordered_to_float(b["aabb"][:3]), ora.ordered_to_float(b["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    for width in (3, 4, 7):
        nodes, root, count = edge_scenes.collapse_wide(b["nodes"], 0, 2, width, rt.NODE)
        gw = _upload_tree(rt, nodes, b["leaves"])
        oi, oc = ora.trace(b["leaves"], nodes, root, count, cam, 160, 100, render_type=1)
        gi, gc = gpu_trace(gw, cam, 160, 100, 1, root=root, count=count, num_primitives=BIG)
        assert (gc == oc[:2]).all(), f"width {width}"
        assert (gi == oi).all()
    gb = _gpu_bu(rt, tris)
    # interleaved strips (rt_trace_strips: the multi-GPU partition's compact output) through the same instantiation
    import torch
    full, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, 130, 101, render_type=0)
    for first, stride in ((0, 3), (2, 3)):
        nstr = len(range(first, (101 + 7) // 8, stride))
        compact = torch.zeros(nstr * 8 * 130 * 4, dtype=torch.uint8, device="cuda")
        rt.Trace(gb["inp"].triangles_out, gb["inp"].nodes_out, compact, (130, 101), rt.to_device(cam), 0, 2, strips=(8, first, stride),
                 num_primitives=BIG)
        torch.cuda.synchronize()
        got = compact.cpu().numpy().reshape(nstr * 8, 130, 4)
        for j, st in enumerate(range(first, (101 + 7) // 8, stride)):
            rows_in = min(8, 101 - st * 8)
            assert (got[j * 8: j * 8 + rows_in] == full[st * 8: st * 8 + rows_in]).all(), f"strip {st}"
    for rows, spp in (((13, 50), 1), ((0, 101), 16)):
        oi, oc = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, 130, 101, render_type=0, rows=rows, spp=spp)
        gi, gc = gpu_trace(gb, cam, 130, 101, 0, rows=rows, spp=spp, num_primitives=BIG)
        assert (gc == oc[:2]).all() and (gi[rows[0]:rows[1]] == oi[rows[0]:rows[1]]).all()
