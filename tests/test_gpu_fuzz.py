"""Randomised GPU-vs-oracle parity over scene shapes the fixed tests do not enumerate: random sizes (incl. just above /
below workgroup and tile boundaries), clustered and duplicated triangles, flat axes, huge and tiny triangles mixed,
every builder (bottom-up, pairs, hybrid, SAH with pairs / splits) -- Node[] and TrianglePair[] bit-exact, kDepth and
kBoxtests frames byte-exact, counters equal.  Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rng, scenes):
    kind = rng.integers(0, 6)
    n = int(rng.choice([3, 7, 63, 64, 65, 257, 511, 513, 1023, 1024, 1025, 2049, 4097, 5000, 12345, 32769, 40000, 70001, 262145]))
    if kind == 0:
        return scenes.soup(n, int(rng.integers(1, 1000)), dup_fraction=float(rng.choice([0.0, 0.25, 0.9])),
                           size=float(rng.choice([0.002, 0.02, 0.3])))
    if kind == 1:   # clustered: a few tight clusters far apart (empty grid cells, long Morton prefixes)
        t = scenes.soup(n, int(rng.integers(1, 1000)), dup_fraction=0.1, size=0.01).reshape(-1, 3, 3)
        centres = rng.uniform(-100, 100, (5, 3)).astype(np.float32)
        t = t * np.float32(0.5) + centres[rng.integers(0, 5, t.shape[0])][:, None, :]
        return t.reshape(-1, 9).astype(np.float32)
    if kind == 2:   # a flat axis (NaN clamp in the Morton code, NaN grid cell in the SAH front end)
        t = scenes.soup(n, int(rng.integers(1, 1000)), dup_fraction=0.0, size=0.05).reshape(-1, 3, 3)
        t[:, :, int(rng.integers(0, 3))] = np.float32(rng.uniform(-1, 1))
        return t.reshape(-1, 9)
    if kind == 3:   # one huge triangle stretches the bounds: everything else falls into one grid cell / Morton octant
        t = scenes.soup(n, int(rng.integers(1, 1000)), dup_fraction=0.0, size=0.01)
        t[0] = np.array([-500, -500, -500, 500, -500, 500, 0, 600, 0], np.float32)
        return t
    if kind == 4:   # shared-edge mesh (pairs merge) with a ragged tail
        g = int(rng.integers(2, 70))
        return scenes.grid_mesh(g, int(rng.integers(1, 100)))[:max(3, 2 * g * g - int(rng.integers(0, 5)))]
    # negative coordinates and mixed magnitudes (ordered-int encoding of negative floats)
    t = scenes.soup(n, int(rng.integers(1, 1000)), dup_fraction=0.2, size=0.1)
    return ((t - np.float32(0.5)) * np.float32(rng.choice([1e-3, 1.0, 1e4]))).astype(np.float32)


def _camera(scenes, tris):
    lo, hi = tris.reshape(-1, 3).min(axis=0), tris.reshape(-1, 3).max(axis=0)
    hi = np.maximum(hi, lo + np.float32(1e-3))
    return scenes.camera_for_box(lo, hi)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RT_FUZZ_SEEDS", "24"))))
def test_fuzz_builders_and_traces(seed, rt, scenes, ora):
    import torch
    from helpers import assert_nodes_equal, gpu_trace
    rng = np.random.default_rng(1000 + seed)
    tris = np.ascontiguousarray(_scene(rng, scenes), np.float32).reshape(-1, 9)
    n = tris.shape[0]
    cam = _camera(scenes, tris)
    mode = seed % 6
    if mode in (0, 1, 2):                       # bottom-up / pairs / hybrid
        pairs, hybrid = mode == 1, mode == 2
        inp = rt.BuildInput.allocate(tris)
        inp.nodes_out.fill_(0xCD)
        rt.RunBottomUpBuild(inp, rt.Arguments(build_type=rt.kHybrid if hybrid else rt.kBottomUp, enable_pairs=pairs), hybrid=hybrid)
        torch.cuda.synchronize()
        o = ora.build_pairs(tris) if pairs else (ora.build_hybrid(tris) if hybrid else ora.build_bvh(tris))
        L = o.get("L", n)
        slots = o["nodes"].shape[0]
        got = rt.to_host(inp.nodes_out, rt.NODE, slots)
        if hybrid:   # slots the hybrid build never writes keep the poison
            used = (o["nodes"]["w28"] >> 29) != 0
            assert_nodes_equal(got[used], o["nodes"][used], f"seed {seed}")
        else:
            assert_nodes_equal(got, o["nodes"], f"seed {seed}")
        assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, L).tobytes() == o["leaves"][:L].tobytes()
        root, count = o.get("root", 0), 2
        build = dict(inp=inp)
    else:                                       # SAH, SAH + pairs, SAH + splits (+ pairs on odd seeds)
        pairs, splits = mode == 4 or (mode == 5 and seed % 2 == 1), mode == 5
        inp = rt.BuildInput.allocate(tris, sah=True)
        rt.RunSahBuild(inp, rt.Arguments(build_type=rt.kSAH, enable_pairs=pairs, enable_splits=splits))
        torch.cuda.synchronize()
        lay = rt.sah_scratch_layout(n)
        status = rt.to_host(inp.scratch, np.uint32, 8, lay.status)
        assert status[0] == 0
        o = ora.build_sah(tris, pairs, splits)
        assert int(status[1]) == o["L"] and int(status[2]) == o["R"]
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 128 + 2 * o["L"]), o["nodes"], f"seed {seed}")
        assert rt.to_host(inp.triangles_out, rt.TRIANGLE_PAIR, o["R"]).tobytes() == o["leaves"].tobytes()
        root, count = 0, 1
        build = dict(inp=inp)
    for render in (0, 1):
        got, gc = gpu_trace(build, cam, 200, 144, render, root=root, count=count)
        exp, oc = ora.trace(o["leaves"], o["nodes"], root, count, cam, 200, 144, render_type=render)
        assert (got == exp).all(), (seed, render)
        assert int(gc[0]) == int(oc[0]) and int(gc[1]) == int(oc[1])
