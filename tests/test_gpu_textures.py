"""GPU parity of the textured render types (kLODs, kTexture, kTextureLit, kTextureLitShadows; Tracer.cu:58-469,
543-590) against the CPU oracle on the same scene, textures and cameras: BYTE-EXACT.

These modes are floating point end to end (log2f LOD selection, bilinear / trilinear blends, a normal map through
normalize(), double pow() specular, a shadow ray started on the surface), and two of their steps are discontinuous:
int(lod) picks the mip level of kLODs and of the diffuse texture in kTextureLit*, and the shadow ray either hits or
misses.  Every operation on the path is IEEE-exact on both sides (kernels and oracle are compiled -ffp-contract=off,
division and sqrt correctly rounded) except the three transcendental calls (log2f, powf(2, lod), double pow), and
those are csrc/rt_math.h on both sides -- plain double arithmetic in a fixed order, itself checked against libm in
tests/test_oracle_cpu.py.  (Round 1 used the device library's log2f / pow against libm's and needed a 2 LSB tolerance
plus an allowance for mip-level flips.)  What stays unpinned is the reference's own CUDA log2f / pow, specified to
1-2 ulp: no implementation can match those bit for bit without the CUDA library."""
import numpy as np
import pytest

import texture_scene

pytestmark = pytest.mark.gpu

W, H = 320, 200


@pytest.fixture(scope="module")
def world(scenes, ora):
    from helpers import gpu_build
    sc = texture_scene.make(scenes, ora)
    g = gpu_build(sc["tris"])
    o = ora.build_bvh(sc["tris"])
    assert g["nodes"].tobytes() == o["nodes"].tobytes()
    return sc, g, o


def _both(world, ora, cam_name, render_type):
    from helpers import gpu_trace
    sc, g, o = world
    cam = sc["cameras"][cam_name]
    kw = dict(attributes=sc["attributes"], materials=sc["materials"], light=sc["light"], textures=sc["textures"])
    got, gc = gpu_trace(g, cam, W, H, render_type=render_type, **kw)
    exp, oc = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, W, H, render_type=render_type, **kw)
    return got.astype(np.int32), exp.astype(np.int32), gc, oc


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_lods_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 4)
    assert gc[0] == oc[0] and gc[1] == oc[1], "box / triangle test counters"
    assert (got == exp).all(), np.abs(got - exp).max()      # grey = int(lod) * 20: the mip level of every pixel agrees
    assert len(np.unique(exp[..., 0])) >= 3


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 6)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    assert (got == exp).all(), np.abs(got - exp).max()      # incl. alpha
    assert len(np.unique(exp.reshape(-1, 4), axis=0)) > 500, "the frame is actually textured"


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_lit_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 7)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    assert (got == exp).all(), np.abs(got - exp).max()


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_lit_shadows_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 8)
    # only the primary rays are counted: the shadow traversal has its own stats (Tracer.cu:451)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    assert (got == exp).all(), np.abs(got - exp).max()
    lit, _, _, _ = _both(world, ora, cam, 7)
    assert (got[..., :3] <= lit[..., :3]).all(), "shadows only remove light"
    assert (got != lit).any(axis=-1).mean() > 0.01, "some pixels are shadowed"


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_discontinuities_agree(world, ora, cam):
    """The two discontinuous decisions of the lit modes -- BilinearSample(tex, uv, (int)lod) (Tracer.cu:432-445) and the
    shadow ray's hit / miss (:447-462) -- agree pixel for pixel between GPU and oracle: kLODs draws int(lod) of the same
    ComputeLOD call, and "kTextureLitShadows differs from kTextureLit" marks the shadowed pixels.  (With the device
    library's log2f against libm's, round 1 had up to 0.2 % of pixels on the other side of a mip boundary and every
    > 2 LSB outlier of the lit modes was one of those or a flipped shadow decision.)"""
    lod_g, lod_o, _, _ = _both(world, ora, cam, 4)
    assert (lod_g == lod_o).all()
    g7, o7, _, _ = _both(world, ora, cam, 7)
    g8, o8, _, _ = _both(world, ora, cam, 8)
    assert ((g8 != g7).any(axis=-1) == (o8 != o7).any(axis=-1)).all()
    assert (g7 == o7).all() and (g8 == o8).all()


def test_out_of_range_material_and_texture_indices(world, rt, ora):
    """material_id -1 (faces before the first usemtl, FileIO.cpp:191) and texture indices beyond the table are
    range-checked on the device (material 0 / untextured) instead of read out of bounds; same rule in the oracle."""
    from helpers import gpu_trace
    sc, g, o = world
    at = sc["attributes"].copy()
    at["material_id"][::7] = -1
    at["material_id"][3::11] = 1 << 20
    mats = sc["materials"].copy()
    mats["texture"][-1] = 99
    cam = sc["cameras"]["oblique"]
    for render_type in (3, 5, 6):
        kw = dict(attributes=at, materials=mats, light=sc["light"], textures=sc["textures"])
        got, _ = gpu_trace(g, cam, W, H, render_type=render_type, **kw)
        exp, _ = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, W, H, render_type=render_type, **kw)
        assert (got == exp).all(), (render_type, np.abs(got.astype(np.int32) - exp.astype(np.int32)).max())


def test_missing_texture_table_is_an_error(world, rt):
    """materials that index textures with no table -> RT_ERR_INVALID_ARGUMENT, not a fault"""
    import torch
    sc, g, _ = world
    inp = g["inp"]
    cam_d = rt.to_device(sc["cameras"]["top"])
    rgba = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(rt.RtError):
        rt.Trace(inp.triangles_out, inp.nodes_out, rgba, (W, H), cam_d, 0, 2, render_type=6,
                 attributes=None, materials=None, num_materials=0)
