"""GPU parity tests proper: the HIP path (through the C ABI of include/rt_abi.h) against the CPU oracle on the
same seeded inputs.  Integer / index work is bit-exact; boxes are exact float equality; kDepth / kBoxtests /
kTriangleTests frames are byte-exact, and so is kDiffuse (its double pow() is csrc/rt_math.h on both sides)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes(scenes):
    return {
        "grid24": scenes.grid_mesh(24, 1),          # 1152 tris, the SURVEY appendix-A count case
        "grid37": scenes.grid_mesh(37, 5),          # 2738 tris: 3 leaf workgroups, ragged tail
        "soup65536": scenes.soup(65536, 7),         # 25 % duplicate centroids -> cpl index tie-break
        "flat20": scenes.flat_mesh(20, 3),          # flat axis -> NaN clamp path
        "soup5000": scenes.soup(5000, 11, dup_fraction=0.9),  # long runs of equal codes
    }


@pytest.fixture(scope="module")
def built(scenes, ora):
    from helpers import gpu_build
    out = {}
    for name, tris in _scenes(scenes).items():
        out[name] = (tris, gpu_build(tris), ora.build_bvh(tris))
    return out


@pytest.mark.parametrize("name", ["grid24", "grid37", "soup65536", "flat20", "soup5000"])
def test_scene_aabb_and_morton(name, built):
    tris, g, o = built[name]
    assert (g["aabb"] == o["aabb"]).all(), "ordered-int scene box"
    assert (g["codes"] == o["codes"]).all(), "sorted Morton codes"
    assert (g["indices"] == o["indices"]).all(), "sorted triangle indices (stable order)"


@pytest.mark.parametrize("name", ["grid24", "grid37", "soup65536", "flat20", "soup5000"])
def test_nodes_and_leaves_bit_exact(name, built, ora):
    from helpers import assert_nodes_equal
    tris, g, o = built[name]
    n = g["n"]
    assert_nodes_equal(g["nodes"], o["nodes"], name)
    assert g["leaves"].tobytes() == o["leaves"].tobytes(), "TrianglePair[] bytes"
    # the reference's own structural checks (Utilities.cpp), restated and -- when present -- the compiled original
    assert ora.count_nodes(g["nodes"], 0, 2) == (2 * n - 2, n, n - 2)
    assert ora.verify_hierarchy(g["nodes"], 0, 2) == 0
    if ora.ref_available():
        assert ora.ref_count_nodes(g["nodes"], 0, 2) == (2 * n - 2, n, n - 2)
        assert ora.ref_verify_hierarchy(g["nodes"], 0, 2) == ""


def test_standalone_stage_entry_points(rt, scenes, ora):
    """rt_calculate_scene_aabb / rt_generate_morton_codes / rt_radix_sort_u32_pairs called on their own."""
    import torch
    tris = scenes.soup(10007, 3)
    n = tris.shape[0]
    d_tri = rt.to_device(tris)
    d_aabb = torch.zeros(6, dtype=torch.int32, device="cuda")
    rt.CalculateSceneAabb(d_tri, n, d_aabb)
    exp_aabb = ora.scene_aabb(tris)
    assert (d_aabb.cpu().numpy() == exp_aabb).all()
    d_codes = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_vals = torch.zeros(n, dtype=torch.int32, device="cuda")
    rt.GenerateMortonCodes(d_codes, d_vals, d_tri, d_aabb, n)
    ec, ev = ora.morton_codes(tris, exp_aabb)
    assert (d_codes.cpu().numpy().view(np.uint32) == ec).all()
    assert (d_vals.cpu().numpy().view(np.uint32) == ev).all()
    t1, t2 = torch.zeros_like(d_codes), torch.zeros_like(d_vals)
    rt.RadixSort(d_codes, d_vals, t1, t2, n)
    sk, sv = ora.radix_sort(ec, ev)
    assert (d_codes.cpu().numpy().view(np.uint32) == sk).all()
    assert (d_vals.cpu().numpy().view(np.uint32) == sv).all()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 100003, 1 << 20])
@pytest.mark.parametrize("bits", [32, 6])
def test_radix_sort_stable(rt, ora, n, bits):
    """Full 32-bit keys and heavy-duplicate keys; result must equal a stable sort (values carry input order)."""
    import torch
    rng = np.random.default_rng(n * 31 + bits)
    keys = rng.integers(0, 2 ** bits, size=n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = rt.to_device(keys).view(torch.int32), rt.to_device(vals).view(torch.int32)
    t1, t2 = torch.zeros_like(dk), torch.zeros_like(dv)
    rt.RadixSort(dk, dv, t1, t2, n)
    order = np.argsort(keys, kind="stable")
    assert (dk.cpu().numpy().view(np.uint32) == keys[order]).all()
    assert (dv.cpu().numpy().view(np.uint32) == vals[order]).all()
    sk, sv = ora.radix_sort(keys, vals)
    assert (sk == keys[order]).all() and (sv == vals[order]).all()


@pytest.mark.parametrize("n", [1, 4097, 300001, 2097152 + 7])
@pytest.mark.parametrize("key_bits", [30, 17, 32])
@pytest.mark.parametrize("in_tmp", [False, True])
def test_radix_sort_bits_entry_point(rt, n, key_bits, in_tmp):
    """rt_radix_sort_u32_pairs_bits: keys of `key_bits` significant bits, input on either side (the library copies it across
    only when the pass count wants it elsewhere); result in keys / values, equal to a stable sort."""
    import torch
    rng = np.random.default_rng(n + key_bits)
    keys = rng.integers(0, 2 ** key_bits, size=n, dtype=np.uint64).astype(np.uint32)
    keys[::3] = keys[0]                                  # heavy duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = rt.to_device(keys).view(torch.int32), rt.to_device(vals).view(torch.int32)
    junk = torch.full_like(dk, -1)
    k, v, tk, tv = (junk.clone(), junk.clone(), dk.clone(), dv.clone()) if in_tmp else (dk.clone(), dv.clone(), junk.clone(), junk.clone())
    assert rt.lib().rt_radix_sort_input_in_tmp(n, key_bits) in (0, 1)
    rt.RadixSortBits(k, v, tk, tv, n, key_bits, input_in_tmp=in_tmp)
    order = np.argsort(keys, kind="stable")
    assert (k.cpu().numpy().view(np.uint32) == keys[order]).all()
    assert (v.cpu().numpy().view(np.uint32) == vals[order]).all()


@pytest.mark.parametrize("pattern", ["equal", "sorted", "reversed", "two", "blocks", "misaligned"])
def test_radix_sort_adversarial_inputs(rt, pattern):
    """Key distributions that stress the histogram kernels' clustered-digit path (lanes queueing on one LDS word), the
    ballot ranking with 64 equal digits, and the scalar (not 16-byte aligned) key path: all equal, already sorted, reversed,
    two values alternating, long constant runs, and key / value pointers offset by one element."""
    import torch
    n = 1_000_003
    rng = np.random.default_rng(7)
    if pattern == "equal":
        keys = np.full(n, 0xDEADBEEF, np.uint32)
    elif pattern == "sorted":
        keys = np.sort(rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32))
    elif pattern == "reversed":
        keys = np.sort(rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    elif pattern == "two":
        keys = np.where(np.arange(n) % 2 == 0, 0x01010101, 0xFEFEFEFE).astype(np.uint32)
    elif pattern == "blocks":
        keys = np.repeat(rng.integers(0, 2 ** 32, size=n // 5000 + 1, dtype=np.uint64).astype(np.uint32), 5000)[:n]
    else:
        keys = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    shift = 1 if pattern == "misaligned" else 0          # element offset into over-allocated buffers: 4-byte alignment only
    bufs = [torch.zeros(n + 4, dtype=torch.int32, device="cuda") for _ in range(4)]
    dk, dv, t1, t2 = (b[shift:shift + n] for b in bufs)
    dk.copy_(rt.to_device(keys).view(torch.int32))
    dv.copy_(rt.to_device(vals).view(torch.int32))
    rt.RadixSort(dk, dv, t1, t2, n)
    order = np.argsort(keys, kind="stable")
    assert (dk.cpu().numpy().view(np.uint32) == keys[order]).all()
    assert (dv.cpu().numpy().view(np.uint32) == vals[order]).all()
    # nothing outside the n elements was touched (the scatter goes through range-checked buffer descriptors)
    for b in bufs:
        assert int(b[:shift].abs().sum()) == 0 and int(b[shift + n:].abs().sum()) == 0


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5])
def test_tiny_builds(rt, scenes, ora, n):
    """n < 2 is special-cased; 2..5 exercise a single tiny workgroup."""
    from helpers import gpu_build, assert_nodes_equal, gpu_trace
    tris = scenes.soup(max(n, 1), 2, dup_fraction=0.0, size=0.3)[:n]
    g, o = gpu_build(tris), ora.build_bvh(tris)
    assert_nodes_equal(g["nodes"], o["nodes"], f"n={n}")
    assert g["leaves"].tobytes() == o["leaves"].tobytes()
    cam = scenes.camera_for_box([0, 0, 0], [1, 1, 1])
    gi, gc = gpu_trace(g, cam, 64, 48)
    oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 64, 48)
    assert (gi == oi).all() and (gc == oc[:2]).all()


@pytest.mark.parametrize("name,w,h", [("grid24", 256, 256), ("soup65536", 320, 200), ("flat20", 100, 60), ("grid37", 97, 53)])
@pytest.mark.parametrize("render_type", [0, 1, 2])
def test_trace_exact_modes(name, w, h, render_type, built, scenes, ora):
    """kDepth / kBoxtests / kTriangleTests: byte-exact frames and identical test counters."""
    from helpers import gpu_trace
    tris, g, o = built[name]
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    for cam in (scenes.camera_for_box(lo, hi), scenes.camera_for_box(lo, hi, yaw=-2.1, pitch=0.9, back=0.8)):
        gi, gc = gpu_trace(g, cam, w, h, render_type)
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, w, h, render_type=render_type)
        assert (gc == oc[:2]).all(), f"counters {gc} vs {oc}"
        diff = np.nonzero((gi != oi).any(axis=2))
        assert diff[0].size == 0, f"{diff[0].size} pixels differ, first {diff[0][:4]},{diff[1][:4]}"


@pytest.mark.parametrize("name", ["grid24", "soup65536"])
def test_trace_diffuse_and_material_id(name, built, scenes, ora):
    from helpers import gpu_trace
    tris, g, o = built[name]
    n = g["n"]
    mats = scenes.default_materials(3)
    at = scenes.flat_attributes(tris, np.arange(n, dtype=np.int32) % 3)
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    light = tuple(float(x) for x in (hi + (hi - lo) * 0.5))
    for rtype, tol in ((5, 0), (3, 0)):     # kDiffuse too: the specular pow() is rt_math.h's on both sides
        gi, gc = gpu_trace(g, cam, 200, 150, rtype, attributes=at, materials=mats, light=light)
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 200, 150, render_type=rtype, attributes=at,
                           materials=mats, light=light)
        d = np.abs(gi.astype(np.int32) - oi.astype(np.int32))
        assert d.max() <= tol, f"render {rtype}: max channel diff {d.max()}"
        assert (d > 0).sum() <= 0.001 * d.size, f"render {rtype}: {(d > 0).sum()} channels differ"
        assert (gc == oc[:2]).all()
        assert (oi[..., :3].max() > 0), "frame is not empty"


def test_trace_row_bands_and_spp(built, scenes, ora):
    """rows=(y0,y1) bands tile the frame exactly (multi-GPU sharding unit) and the spp extension matches."""
    from helpers import gpu_trace
    tris, g, o = built["grid24"]
    lo, hi = ora.ordered_to_float(o["aabb"][:3]), ora.ordered_to_float(o["aabb"][3:])
    cam = scenes.camera_for_box(lo, hi)
    w, h = 128, 101
    full, fc = gpu_trace(g, cam, w, h, 0)
    acc = np.zeros_like(full)
    tot = np.zeros(2, np.uint64)
    for y0, y1 in ((0, 13), (13, 50), (50, 51), (51, 101)):
        part, pc = gpu_trace(g, cam, w, h, 0, rows=(y0, y1))
        assert (part[:y0] == 0).all() and (part[y1:] == 0).all(), "band wrote outside its rows"
        acc[y0:y1] = part[y0:y1]
        tot += pc
    assert (acc == full).all() and (tot == fc).all()
    for spp in (4, 16):
        gi, gc = gpu_trace(g, cam, 64, 64, 0, spp=spp)
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 64, 64, render_type=0, spp=spp)
        assert (gi == oi).all() and (gc == oc[:2]).all()


def test_bottom_up_ignores_splits_like_the_reference(rt, scenes):
    """--splits only exists in RunSahBuild's front end (BuildWrapper.cu:188-210); RunBottomUpBuild never looks at it."""
    import torch
    tris = scenes.grid_mesh(12, 1)
    a, b = rt.BuildInput.allocate(tris), rt.BuildInput.allocate(tris)
    rt.RunBottomUpBuild(a, rt.Arguments(build_type=rt.kBottomUp))
    rt.RunBottomUpBuild(b, rt.Arguments(build_type=rt.kBottomUp, enable_splits=True))
    torch.cuda.synchronize()
    k = 2 * (tris.shape[0] - 1) * 32
    assert torch.equal(a.nodes_out[:k], b.nodes_out[:k])


def test_invalid_arguments_fail_loudly(rt, scenes):
    import ctypes
    tris = scenes.grid_mesh(4, 1)
    inp = rt.BuildInput.allocate(tris)
    ci = rt._BuildInput(0, 0, tris.shape[0], 0, 0)          # null buffers
    assert rt.lib().rt_run_bottom_up_build(ctypes.byref(ci), None, 0, None) == -1
    assert rt.lib().rt_run_sah_build(ctypes.byref(ci), None, None) == -1
    big = rt._BuildInput(rt._ptr(inp.triangles_in), rt._ptr(inp.triangles_out), (1 << 28) + 1, rt._ptr(inp.nodes_out),
                         rt._ptr(inp.scratch))
    assert rt.lib().rt_run_bottom_up_build(ctypes.byref(big), None, 0, None) == -3   # RT_ERR_TOO_LARGE, nothing launched


def test_full_size_1m_properties(rt, scenes, ora):
    """BASELINE config 2 at full size (grid G=708, 1,002,528 triangles, 1920x1080): size-independent properties
    plus a full comparison against the oracle (the oracle builds 1M in about a second)."""
    from helpers import gpu_build, gpu_trace, assert_nodes_equal
    G = 708
    tris = scenes.grid_mesh(G, 1)
    n = tris.shape[0]
    assert n == 1002528
    g = gpu_build(tris)
    assert (np.diff(g["codes"].astype(np.int64)) >= 0).all(), "codes sorted"
    assert (np.sort(g["indices"]) == np.arange(n)).all(), "indices are a permutation"
    eq = g["codes"][1:] == g["codes"][:-1]
    assert (g["indices"][1:][eq] > g["indices"][:-1][eq]).all(), "stable within equal codes"
    assert ora.count_nodes(g["nodes"], 0, 2) == (2 * n - 2, n, n - 2)
    assert ora.verify_hierarchy(g["nodes"], 0, 2) == 0
    if ora.ref_available():
        assert ora.ref_verify_hierarchy(g["nodes"], 0, 2) == ""
        assert ora.ref_count_nodes(g["nodes"], 0, 2) == (2 * n - 2, n, n - 2)
    o = ora.build_bvh(tris)
    assert_nodes_equal(g["nodes"], o["nodes"], "1M grid")
    assert g["leaves"].tobytes() == o["leaves"].tobytes()
    # both cameras at the full 1920x1080: camera A ("top-down") is bench.py's headline frame (448 box tests per ray),
    # camera B ("oblique") the short-traversal one; whole frame byte-exact and sum(box) / sum(tri) identical
    for cam in (scenes.camera_a(G), scenes.camera_b(G)):
        gi, gc = gpu_trace(g, cam, 1920, 1080, 0)
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 1920, 1080, render_type=0)
        assert (gc == oc[:2]).all(), f"{gc} vs {oc}"
        assert (gi == oi).all()
    assert int(gc[0]) > 0


def test_full_size_config5_4k_16spp(rt, scenes, ora):
    """BASELINE config 5 at full size: the 1M-triangle mesh, 3840x2160, 16 spp (camera A).  The GPU renders the whole
    frame (132.7 M rays); the oracle renders two bands of 32 full rows (3.9 M rays); those rows must be byte-exact, the
    bands' sum(box) / sum(tri) identical to a GPU launch restricted to the same rows, and the band launches of a
    4-way split must add up to the full frame's counters (what a multi-GPU run relies on)."""
    import torch
    from helpers import gpu_build, gpu_trace
    G, W, H, SPP = 708, 3840, 2160, 16
    tris = scenes.grid_mesh(G, 1)
    g = gpu_build(tris)
    o = ora.build_bvh(tris)
    assert g["nodes"].tobytes() == o["nodes"].tobytes()
    cam = scenes.camera_a(G)
    full, fc = gpu_trace(g, cam, W, H, 0, spp=SPP)
    assert int((full[..., 0] > 0).sum()) > W * H // 2
    for r0 in (256, 1064):
        r1 = r0 + 32
        oi, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, W, H, render_type=0, spp=SPP, rows=(r0, r1))
        assert (full[r0:r1] == oi[r0:r1]).all(), f"rows [{r0}, {r1}) differ from the oracle"
        _, bc = gpu_trace(g, cam, W, H, 0, spp=SPP, rows=(r0, r1))
        assert (bc == oc[:2]).all(), f"band counters {bc} vs oracle {oc}"
    total = np.zeros(2, np.uint64)
    for k in range(4):
        _, bc = gpu_trace(g, cam, W, H, 0, spp=SPP, rows=(k * H // 4, (k + 1) * H // 4))
        total += bc
    assert (total == fc).all(), f"band sums {total} vs full frame {fc}"


def test_build_and_trace_capture_in_a_hip_graph(rt, scenes, ora):
    """The C ABI neither allocates nor synchronises, so a whole frame -- LBVH rebuild + trace -- can be captured in a
    hipGraph and replayed (changing triangles / camera in place between replays)."""
    import torch
    from helpers import assert_nodes_equal
    tris_a, tris_b = scenes.grid_mesh(30, 1), scenes.grid_mesh(30, 9)
    n, w, h = tris_a.shape[0], 160, 96
    inp = rt.BuildInput.allocate(tris_a)
    cam_d = rt.to_device(scenes.camera_b(30))
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside capture (lazy module load, func attributes)
        rt.RunBottomUpBuild(inp)
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (w, h), cam_d, 0, 2, counters=counters)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rt.RunBottomUpBuild(inp)
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (w, h), cam_d, 0, 2, counters=counters)
    for tris in (tris_b, tris_a):
        inp.triangles_in.copy_(rt.to_device(tris))
        counters.zero_()
        frame.zero_()
        g.replay()
        torch.cuda.synchronize()
        o = ora.build_bvh(tris)
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 2 * (n - 1)), o["nodes"], "graph replay")
        img, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, scenes.camera_b(30), w, h)
        assert (frame.cpu().numpy().reshape(h, w, 4) == img).all()
        assert (counters.cpu().numpy()[:2].astype(np.uint64) == oc[:2]).all()


def test_full_size_10m_build(rt, scenes, ora):
    """BASELINE config 4 (grid G=2237, 10,008,338 triangles, builder-bound): the LBVH bit for bit against the oracle,
    plus the size-independent properties (sorted, stable, the reference's compiled checker over all 20M slots)."""
    from helpers import gpu_build, assert_nodes_equal
    tris = scenes.grid_mesh(2237, 1)
    n = tris.shape[0]
    assert n == 10008338
    g = gpu_build(tris)
    assert (np.diff(g["codes"].astype(np.int64)) >= 0).all(), "codes sorted"
    eq = g["codes"][1:] == g["codes"][:-1]
    assert (g["indices"][1:][eq] > g["indices"][:-1][eq]).all(), "stable within equal codes"
    assert ora.count_nodes(g["nodes"], 0, 2) == (2 * n - 2, n, n - 2)
    assert ora.verify_hierarchy(g["nodes"], 0, 2) == 0
    if ora.ref_available():
        assert ora.ref_verify_hierarchy(g["nodes"], 0, 2) == ""
    o = ora.build_bvh(tris)
    assert (g["codes"] == o["codes"]).all() and (g["indices"] == o["indices"]).all()
    assert_nodes_equal(g["nodes"], o["nodes"], "10M")
    assert g["leaves"].tobytes() == o["leaves"].tobytes()


def test_lbvh_hierarchy_handoff_alternating_scenes(rt, scenes, ora):
    """The one-launch hierarchy hands open-root records from workgroup to workgroup (release -> ticket -> acquire).
    Rebuilding DIFFERENT scenes of the same size in the SAME buffers would expose a stale read of the previous build's
    records (same addresses, different bytes): 12 alternating builds, 196 leaf blocks -> 4 level-1 blocks -> root,
    every Node word against the oracle each time."""
    import torch
    from helpers import assert_nodes_equal
    sets = [scenes.soup(200000, 5), scenes.soup(200000, 6, dup_fraction=0.6), scenes.soup(200000, 7, dup_fraction=0.0)]
    n = sets[0].shape[0]
    oracles = [ora.build_bvh(t) for t in sets]
    inp = rt.BuildInput.allocate(sets[0])
    lay = rt.scratch_layout(n)
    for it in range(12):
        k = (it * 2 + it // 3) % 3
        inp.triangles_in.copy_(rt.to_device(sets[k]))
        rt.RunBottomUpBuild(inp)
        torch.cuda.synchronize()
        assert int(rt.to_host(inp.scratch, np.uint32, 8, lay.status)[0]) == 0
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 2 * (n - 1)), oracles[k]["nodes"], f"iteration {it} scene {k}")
        assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, n).tobytes() == oracles[k]["leaves"].tobytes()


@pytest.mark.parametrize("scene,n", [("soup", 70000), ("soup", 200000), ("grid", 0), ("dups", 150000)])
def test_lbvh_subpass_path(rt, scenes, ora, scene, n):
    """An upper-level block whose 64 source blocks hold more open roots than one LDS pass takes them kSubFan blocks at
    a time and merges the results.  Real scenes stay far below the threshold (about 10 open roots per 1024-leaf
    block against 16 needed), so the path is exercised through librt_amd_smallcap.so: the same sources with the
    threshold at 48 open roots (csrc/Makefile) -- every multi-block group then takes it.  Same tree, bit for bit."""
    import ctypes
    import torch
    from helpers import assert_nodes_equal
    path = os.path.join(os.path.dirname(rt.LIB_PATH), "librt_amd_smallcap.so")
    L = ctypes.CDLL(path)
    L.rt_run_bottom_up_build.restype = ctypes.c_int
    L.rt_run_bottom_up_build.argtypes = [ctypes.POINTER(rt._BuildInput), ctypes.POINTER(rt._Arguments), ctypes.c_int, ctypes.c_void_p]
    tris = {"soup": lambda: scenes.soup(n, 11), "grid": lambda: scenes.grid_mesh(330, 4),
            "dups": lambda: scenes.soup(n, 12, dup_fraction=0.9)}[scene]()
    m = tris.shape[0]
    inp = rt.BuildInput.allocate(tris)
    inp.nodes_out.fill_(0xCD)
    ci = rt._BuildInput(rt._ptr(inp.triangles_in), rt._ptr(inp.triangles_out), m, rt._ptr(inp.nodes_out), rt._ptr(inp.scratch))
    ca = rt._Arguments(rt.kBottomUp, 0, 0, 0)
    for _ in range(2):
        assert L.rt_run_bottom_up_build(ctypes.byref(ci), ctypes.byref(ca), 0, rt._stream_ptr(None)) == 0
    torch.cuda.synchronize()
    o = ora.build_bvh(tris)
    assert int(rt.to_host(inp.scratch, np.uint32, 8, rt.scratch_layout(m).status)[0]) == 0
    assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 2 * (m - 1)), o["nodes"], f"{scene} {m} (sub-pass path)")


@pytest.mark.parametrize("n", [511, 512, 513, 1025, 32767, 32768, 32769, 65537, 2097152, 2097153])
def test_lbvh_level_and_sort_boundaries(rt, scenes, ora, n):
    """Sizes on both sides of every geometry switch of the builder: one leaf block (512 leaves) / two; one upper block
    (64 leaf blocks = 32,768 leaves) / two + a third level; 512 sort tiles (2,097,152 keys: 3 x 10-bit passes, the Morton
    kernel writing into the temporaries) / 513 tiles (4 x 8-bit passes) with 4097 leaf blocks -> 65 -> 2 -> 1.  Node[],
    leaves, sorted codes and indices bit-exact."""
    import torch
    from helpers import gpu_build, assert_nodes_equal
    tris = scenes.soup(n, 21, dup_fraction=0.3 if n < 100000 else 0.05)
    g = gpu_build(tris)
    o = ora.build_bvh(tris)
    assert (g["codes"] == o["codes"]).all() and (g["indices"] == o["indices"]).all()
    assert_nodes_equal(g["nodes"], o["nodes"], f"n = {n}")
    assert g["leaves"].tobytes() == o["leaves"].tobytes()
    if n <= 65537:      # the same through the hybrid and the pairs builds (device-side leaf count L < n)
        inp = rt.BuildInput.allocate(tris)
        rt.RunBottomUpBuild(inp, rt.Arguments(build_type=rt.kBottomUp, enable_pairs=True))
        torch.cuda.synchronize()
        op = ora.build_pairs(tris)
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, op["nodes"].shape[0]), op["nodes"], f"pairs n = {n}")


def test_lbvh_handoff_under_concurrent_load(rt, scenes, ora):
    """The upper-level hand-off (write-through records, ticket, acquire by the last arriver) while the GPU is busy with
    something else: trace launches of another scene run on four other streams during every rebuild, so workgroups of the
    build are dispatched unevenly, share CUs (and their L1s) with foreign waves and find the caches full of other data.
    Every Node word and every leaf byte of 10 rebuilds (two alternating scenes, same buffers) against the oracle."""
    import torch
    from helpers import assert_nodes_equal
    load_tris = scenes.grid_mesh(200, 3)
    load = rt.BuildInput.allocate(load_tris)
    rt.RunBottomUpBuild(load)
    cam_d = rt.to_device(scenes.camera_a(200))
    frames = [torch.zeros(1280 * 720 * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    side = [torch.cuda.Stream() for _ in range(4)]
    sets = [scenes.soup(300000, 31, dup_fraction=0.2), scenes.grid_mesh(388, 8)[:300000]]
    n = 300000
    oracles = [ora.build_bvh(t) for t in sets]
    inp = rt.BuildInput.allocate(sets[0])
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    for it in range(10):
        k = it & 1
        inp.triangles_in.copy_(rt.to_device(sets[k]))
        for s_, fr in zip(side, frames):
            s_.wait_stream(main)
            with torch.cuda.stream(s_):
                for _ in range(3):
                    rt.Trace(load.triangles_out, load.nodes_out, fr, (1280, 720), cam_d, 0, 2)
        rt.RunBottomUpBuild(inp)                      # on the main stream, concurrently with the 12 trace launches
        torch.cuda.synchronize()
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 2 * (n - 1)), oracles[k]["nodes"], f"iteration {it}")
        assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, n).tobytes() == oracles[k]["leaves"].tobytes()


def test_lbvh_build_stress_back_to_back(rt, scenes, ora):
    """600 rebuilds of two alternating scenes into the same buffers without a host synchronisation in between, two out of
    three with trace launches of another scene in flight on four side streams, every Node byte compared on the GPU.
    This is the test that exposes a hand-off store whose data registers are reused too early (csrc/lbvh_levels.hip,
    store_sc1): with such stores about 7 % of these builds carried a wrong box word; the single-build tests passed."""
    import torch
    n = 300000
    load = rt.BuildInput.allocate(scenes.grid_mesh(200, 3))
    rt.RunBottomUpBuild(load)
    cam_d = rt.to_device(scenes.camera_a(200))
    frames = [torch.zeros(1280 * 720 * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    side = [torch.cuda.Stream() for _ in range(4)]
    sets = [scenes.soup(n, 31, dup_fraction=0.2), scenes.grid_mesh(388, 8)[:n]]
    oracles = [ora.build_bvh(t) for t in sets]
    dsets = [rt.to_device(s) for s in sets]
    exp_nodes = [torch.from_numpy(o["nodes"].view(np.uint8).reshape(-1).copy()).cuda() for o in oracles]
    exp_leaves = [torch.from_numpy(o["leaves"].view(np.uint8).reshape(-1).copy()).cuda() for o in oracles]
    inp = rt.BuildInput.allocate(sets[0])
    main = torch.cuda.current_stream()
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for it in range(600):
        k = it & 1
        inp.triangles_in.copy_(dsets[k])
        if it % 3 != 2:
            for s_, fr in zip(side, frames):
                s_.wait_stream(main)
                with torch.cuda.stream(s_):
                    for _ in range(3):
                        rt.Trace(load.triangles_out, load.nodes_out, fr, (1280, 720), cam_d, 0, 2)
        rt.RunBottomUpBuild(inp)
        got_n = inp.nodes_out.view(torch.uint8).reshape(-1)[: exp_nodes[k].numel()]
        got_l = inp.triangles_out.view(torch.uint8).reshape(-1)[: exp_leaves[k].numel()]
        bad += (got_n != exp_nodes[k]).any().to(torch.int64) + (got_l != exp_leaves[k]).any().to(torch.int64)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, f"{int(bad.item())} of 600 builds differ from the oracle"


@pytest.mark.parametrize("poison", [0xCD, 0xFF, 0x00])
@pytest.mark.parametrize("variant", ["bottom-up", "hybrid", "pairs"])
def test_build_is_independent_of_stale_scratch(rt, scenes, ora, poison, variant):
    """The caller's scratch is uninitialised memory (torch.empty here, cudaMalloc in main.cu:231-237): every word the build
    reads from it must have been written earlier in the SAME build.  The scratch (and the outputs) are filled with a
    poison byte before each build; the result must not change.  (Round 2's unexplained GPU fault, gpurun_out/r2l, was
    an uncommitted experiment -- DESIGN section 5 -- but this is the class of defect it would have been in shipped code;
    the kernels also range-check every address they form from a scratch word, status bits 1 and 2.)"""
    import torch
    from helpers import assert_nodes_equal
    pairs, hybrid = variant == "pairs", variant == "hybrid"
    args = rt.Arguments(build_type=rt.kHybrid if hybrid else rt.kBottomUp, enable_pairs=pairs)
    for tris in (scenes.grid_mesh(24, 1), scenes.soup(70001, 5), scenes.grid_mesh(120, 2)):
        n = tris.shape[0]
        inp = rt.BuildInput.allocate(tris)
        inp.scratch.fill_(poison)
        inp.triangles_out.fill_(poison)
        inp.nodes_out.fill_(0 if hybrid else poison)   # (slots the hybrid top tree never writes stay as they were)
        rt.RunBottomUpBuild(inp, args, hybrid=hybrid)
        torch.cuda.synchronize()
        lay = rt.scratch_layout(n)
        status = rt.to_host(inp.scratch, np.uint32, 8, lay.status)
        assert status[0] == 0, f"error flags {status[0]:#x}"
        L = int(status[1])
        o = ora.build_pairs(tris) if pairs else (ora.build_hybrid(tris) if hybrid else ora.build_bvh(tris))
        assert L == o["leaves"].shape[0]
        what = f"{variant} n={n} poison={poison:#x}"
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, o["nodes"].shape[0]), o["nodes"], what)
        assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, L).tobytes() == o["leaves"].tobytes(), what


def test_counters_of_concurrent_launches_do_not_mix(rt, scenes, ora):
    """The test counters are published through slots of module-static device memory (trace_kernel.hip g_ctr: a launch's
    workgroups add into 16 rows of its slot, the last arriver folds them into the caller's buffer and zeroes the slot).
    400 launches on 8 streams, each with its own counter buffer, several frame sizes and row bands (1 .. 8100 workgroups, i.e.
    fewer rows than 16 too), slots reused after 256 launches: every buffer must hold exactly its own frame's sums."""
    import torch
    from helpers import gpu_build
    tris = scenes.grid_mesh(60, 3)
    g, o = gpu_build(tris), ora.build_bvh(tris)
    inp = g["inp"]
    cams = [scenes.camera_a(60), scenes.camera_b(60)]
    cam_d = [rt.to_device(c) for c in cams]
    shapes = [(64, 8, None), (8, 8, None), (200, 120, None), (320, 200, (16, 120)), (1920, 1080, None), (96, 40, (8, 9))]
    expect = {}
    for ci, cam in enumerate(cams):
        for si, (w, h, rows) in enumerate(shapes):
            if w * h > 100000:
                continue                                   # (the big frame's sums come from a single GPU launch below)
            _, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, w, h, rows=rows)
            expect[(ci, si)] = (int(oc[0]), int(oc[1]))
    frames = {s: torch.zeros(s[0] * s[1] * 4, dtype=torch.uint8, device="cuda") for s in {(w, h) for w, h, _ in shapes}}
    for ci in range(2):                                    # reference sums of the 1080p frame: one launch alone
        c = torch.zeros(4, dtype=torch.int64, device="cuda")
        rt.Trace(inp.triangles_out, inp.nodes_out, frames[(1920, 1080)], (1920, 1080), cam_d[ci], 0, 2, counters=c)
        torch.cuda.synchronize()
        expect[(ci, 4)] = (int(c[0]), int(c[1]))
    streams = [torch.cuda.Stream() for _ in range(8)]
    bufs, keys = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in range(400)], []
    torch.cuda.synchronize()                               # (the buffers are zero before any side stream touches them)
    for k in range(400):
        ci, si = k % 2, (k * 7 + k // 5) % len(shapes)
        w, h, rows = shapes[si]
        c = bufs[k]
        with torch.cuda.stream(streams[k % 8]):
            rt.Trace(inp.triangles_out, inp.nodes_out, frames[(w, h)], (w, h), cam_d[ci], 0, 2, counters=c, rows=rows)
        keys.append((ci, si))
    torch.cuda.synchronize()
    got = torch.stack(bufs).cpu().numpy()
    bad = [(k, keys[k], tuple(got[k][:2]), expect[keys[k]]) for k in range(400) if (int(got[k][0]), int(got[k][1])) != expect[keys[k]]]
    assert not bad, bad[:5]
