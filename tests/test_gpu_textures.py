"""GPU parity of the textured render types (kLODs, kTexture, kTextureLit, kTextureLitShadows; Tracer.cu:58-469,
543-590) against the CPU oracle on the same scene, textures and cameras.

These modes are floating point end to end (log2f LOD selection, bilinear / trilinear blends, a normal map through
normalize(), double pow() specular, a shadow ray started on the surface), and two of their steps are discontinuous:
int(lod) picks the mip level of kLODs and of the diffuse texture in kTextureLit*, and the shadow ray either hits or
misses.  The tolerance, stated per mode below: every pixel whose mip level / shadow decision agrees is within 2 LSB
per 8-bit channel; the fraction of pixels that sit on a discontinuity (device log2f / sqrt vs libm differing in the
last bit) is bounded.  Hit / miss, depth and the test counters stay bit-exact (same traversal)."""
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
    d = np.abs(got - exp).max(axis=-1)
    # grey = int(lod) * 20: a differing pixel is a mip level flipped by the last bit of log2f -> exactly one level
    assert set(np.unique(d)) <= {0, 20}, np.unique(d)
    assert (d != 0).mean() < 2e-3, (d != 0).mean()
    assert len(np.unique(exp[..., 0])) >= 3


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 6)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    d = np.abs(got - exp).max(axis=-1)
    # trilinear is continuous across mip levels: everything within 2 LSB (incl. alpha)
    assert d.max() <= 2, (d.max(), (d > 2).sum())
    assert (d == 0).mean() > 0.97
    assert len(np.unique(exp.reshape(-1, 4), axis=0)) > 500, "the frame is actually textured"


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_lit_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 7)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    d = np.abs(got - exp).max(axis=-1)
    # the diffuse texture is sampled at int(lod): pixels on a level boundary may take the neighbouring mip
    assert (d > 2).mean() < 3e-3, ((d > 2).mean(), d.max())
    assert (d == 0).mean() > 0.9


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_texture_lit_shadows_mode(world, ora, cam):
    got, exp, gc, oc = _both(world, ora, cam, 8)
    # only the primary rays are counted: the shadow traversal has its own stats (Tracer.cu:451)
    assert gc[0] == oc[0] and gc[1] == oc[1]
    d = np.abs(got - exp).max(axis=-1)
    assert (d > 2).mean() < 5e-3, ((d > 2).mean(), d.max())
    lit, _, _, _ = _both(world, ora, cam, 7)
    assert (got[..., :3] <= lit[..., :3]).all(), "shadows only remove light"
    assert (got != lit).any(axis=-1).mean() > 0.01, "some pixels are shadowed"


@pytest.mark.parametrize("cam", ["oblique", "top", "grazing"])
def test_outliers_are_mip_level_or_shadow_flips(world, ora, cam):
    """Where do the > 2 LSB pixels of the lit modes come from?  Two steps of these shaders are discontinuous:
    BilinearSample(tex, uv, (int)lod) (Tracer.cu:432-445) and the shadow ray's hit / miss (:447-462).  kLODs draws
    int(lod) of the same ComputeLOD call, so a pixel whose kLODs value differs between GPU and oracle is a pixel whose
    mip level was flipped by the last bit of log2f (device vs libm; CUDA's own log2f is a third 1-ulp variant); a
    pixel whose "kTextureLitShadows differs from kTextureLit" predicate differs is a flipped shadow decision.  EVERY
    outlier must be one of the two; all other pixels are within 2 LSB (stacked u8 truncations of the filters)."""
    lod_g, lod_o, _, _ = _both(world, ora, cam, 4)
    flip_lod = (lod_g != lod_o).any(axis=-1)
    g7, o7, _, _ = _both(world, ora, cam, 7)
    g8, o8, _, _ = _both(world, ora, cam, 8)
    out7 = np.abs(g7 - o7).max(axis=-1) > 2
    assert not (out7 & ~flip_lod).any(), f"{int((out7 & ~flip_lod).sum())} kTextureLit outliers are not mip-level flips"
    flip_shadow = (g8 != g7).any(axis=-1) != (o8 != o7).any(axis=-1)
    out8 = np.abs(g8 - o8).max(axis=-1) > 2
    unexplained = out8 & ~(flip_lod | flip_shadow)
    assert not unexplained.any(), f"{int(unexplained.sum())} kTextureLitShadows outliers are neither mip nor shadow flips"
    # and the flips themselves are rare: a mip level changes where log2f lands within an ulp of an integer
    assert flip_lod.mean() < 2e-3 and flip_shadow.mean() < 5e-3


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
        d = np.abs(got.astype(np.int32) - exp.astype(np.int32)).max()
        assert d <= (0 if render_type == 3 else 2), (render_type, d)


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
